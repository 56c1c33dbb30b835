"""MI355X-native (gfx950) implementation of the camouflage-multimodal fusion hot path.

Public surface = the reference's operator API for this path:
``build_multimodal_model``, ``MultimodalCamouflageDetector`` (fusion_model.py),
``AggressiveFocalLoss``, ``calculate_f1_score``, ``train_epoch_fixed``, ``validate_fixed``
(train_multimodal.py), plus the native packed-batch trainer.  All arithmetic runs in
``libcamo_fusion.so`` (hand-written HIP, C ABI in include/camo_fusion.h); there is no CPU
fallback.
"""
from .embedding_matcher import DeviceResidentDataset, EmbeddingMatcher  # noqa: F401
from .fusion_model import (CrossAttentionFusion, LateFusion, MultimodalCamouflageDetector,  # noqa: F401
                           build_multimodal_model)
from .losses import AggressiveFocalLoss, multitask_loss  # noqa: F401
from .optim import FusedClipAdamW, cosine_warm_restarts_lr  # noqa: F401
from .test_multimodal import (build_ordered_kg_tensor, load_multimodal_model, predict_embedding_directory,  # noqa: F401
                              predict_from_embeddings, predict_from_region_graph)
from .train_multimodal import (NativeTrainer, SmartMultimodalDataset, calculate_f1_score, collate_fn,  # noqa: F401
                               extract_label_from_mask, fit, pack_samples, train_epoch_fixed, train_multimodal_fixed,
                               validate_fixed)

from .region_graph import (RegionGraphData, RegionGraphGNN, build_target_csr, create_region_graph,  # noqa: F401,E402
                           create_region_graph_from_segments)

__all__ = ["RegionGraphGNN", "RegionGraphData", "create_region_graph", "create_region_graph_from_segments", "build_target_csr", "build_multimodal_model", "MultimodalCamouflageDetector", "CrossAttentionFusion", "LateFusion",
           "AggressiveFocalLoss", "multitask_loss", "FusedClipAdamW", "cosine_warm_restarts_lr", "NativeTrainer",
           "calculate_f1_score", "collate_fn", "fit", "pack_samples", "train_epoch_fixed", "validate_fixed",
           "EmbeddingMatcher", "DeviceResidentDataset", "SmartMultimodalDataset", "extract_label_from_mask", "train_multimodal_fixed", "load_multimodal_model", "build_ordered_kg_tensor",
           "predict_from_embeddings", "predict_from_region_graph", "predict_embedding_directory"]
