"""MI355X-native (gfx950) implementation of the camouflage-multimodal fusion hot path.

Public surface = the reference's operator API for this path:
``build_multimodal_model``, ``MultimodalCamouflageDetector`` (fusion_model.py),
``AggressiveFocalLoss``, ``calculate_f1_score``, ``train_epoch_fixed``, ``validate_fixed``
(train_multimodal.py), plus the native packed-batch trainer.  All arithmetic runs in
``libcamo_fusion.so`` (hand-written HIP, C ABI in include/camo_fusion.h); there is no CPU
fallback.
"""
from .fusion_model import (CrossAttentionFusion, LateFusion, MultimodalCamouflageDetector,  # noqa: F401
                           build_multimodal_model)
from .losses import AggressiveFocalLoss, multitask_loss  # noqa: F401
from .optim import FusedClipAdamW, cosine_warm_restarts_lr  # noqa: F401
from .train_multimodal import (NativeTrainer, calculate_f1_score, collate_fn, fit, pack_samples,  # noqa: F401
                               train_epoch_fixed, validate_fixed)

__all__ = ["build_multimodal_model", "MultimodalCamouflageDetector", "CrossAttentionFusion", "LateFusion",
           "AggressiveFocalLoss", "multitask_loss", "FusedClipAdamW", "cosine_warm_restarts_lr", "NativeTrainer",
           "calculate_f1_score", "collate_fn", "fit", "pack_samples", "train_epoch_fixed", "validate_fixed"]
