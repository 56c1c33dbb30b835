"""Inference side of the fusion path (the reference's models/multimodal/test_multimodal.py minus
the SLIC / feature extraction and the matplotlib figures, which are outside the hot path --
SURVEY 8f; the Region-Graph GNN forward that sits between them is region_graph.py).  The model call is the HIP forward with ``return_attention=True``.

Kept from the reference (file:line):
  * ``load_multimodal_model`` reads the training checkpoint dict, rebuilds the model from
    ``checkpoint['config']['model']`` and loads ``model_state_dict`` [:30-55]
  * ``build_ordered_kg_tensor`` orders KG categories by sorted key [:58-80] (the logits do not depend
    on the order; the columns of the returned rg2kg map do)
  * the post-processing of ``predict_single_image``: softmax over mask / instance logits, sigmoid of
    the edge logit, arg-max class [:105-150]
  * ``batch_results.json`` entries [:350-357] and file [:371-375]
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch

from .fusion_model import build_multimodal_model


def load_multimodal_model(checkpoint_path, device):
    """Returns (model in eval mode on ``device``, config).  Accepts the checkpoints the reference
    trainer and this package's trainer write (same dict, train_multimodal.py:464-474)."""
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    config = checkpoint["config"]
    model = build_multimodal_model(config["model"])
    model.load_state_dict(checkpoint["model_state_dict"])
    return model.to(device).eval(), config


def build_ordered_kg_tensor(kg_embeddings):
    """dict {category: [1,128] or [128]} (or a tensor) -> (kg_tensor stacked in sorted-key order,
    OrderedDict category -> embedding)."""
    if isinstance(kg_embeddings, dict) or hasattr(kg_embeddings, "items"):
        ordered = OrderedDict((k, torch.as_tensor(kg_embeddings[k])) for k in sorted(kg_embeddings.keys()))
        return torch.stack(list(ordered.values())), ordered
    kg_tensor = torch.as_tensor(kg_embeddings)
    return kg_tensor, OrderedDict((f"cat_{i}", kg_tensor[i]) for i in range(kg_tensor.shape[0]))


@torch.no_grad()
def predict_from_embeddings(multimodal_model, rg_node_emb, kg_embeddings_dict, device):
    """The fusion part of the reference's ``predict_single_image`` [:97-152]: ``rg_node_emb`` [Nr,128]
    are the image's Region-Graph node embeddings.  Returns (predictions, attention maps, ordered KG dict)."""
    kg_tensor, kg_ordered = build_ordered_kg_tensor(kg_embeddings_dict)
    kg_emb = kg_tensor.unsqueeze(0).to(device)            # [1, Nk, 1, 128] for [1,128] rows: collapsed by the model
    rg = rg_node_emb.unsqueeze(0).to(device)
    mask_out, inst_out, edge_out, score_out, attn = multimodal_model(rg, kg_emb, return_attention=True)
    mask_prob = torch.softmax(mask_out, dim=1)
    inst_prob = torch.softmax(inst_out, dim=1)
    predictions = {
        "mask_logits": mask_out.cpu(),
        "mask_prob": mask_prob.cpu(),
        "mask_pred": int(mask_out.argmax(dim=1).item()),
        "instance_prob": inst_prob.cpu(),
        "instance_pred": int(inst_out.argmax(dim=1).item()),
        "edge_prob": float(torch.sigmoid(edge_out).item()),
        "score": float(score_out.item()),
    }
    return predictions, attn, kg_ordered


def predict_from_region_graph(multimodal_model, rg_model, graph_data, kg_embeddings_dict, device):
    """The reference's ``predict_single_image`` [:83-152] from the region graph on: the Region-Graph GNN turns the
    image's graph (``graph_data.x`` [Nr, 15], ``edge_index``, ``edge_attr`` as ``create_region_graph`` builds them,
    extract_rg_embeddings.py:213-246) into node embeddings on the device [:93 -> extract_rg_embeddings.py:276-279] and
    the fusion model consumes them without a host round trip.  ``rg_model``: a ``RegionGraphGNN`` of this package."""
    class _OnDevice:
        pass
    d = _OnDevice()
    d.x = graph_data.x.to(device); d.edge_index = graph_data.edge_index.to(device)
    ea = getattr(graph_data, "edge_attr", None)
    d.edge_attr = None if ea is None else ea.to(device)
    node_emb = rg_model.extract_node_embeddings(d)
    return predict_from_embeddings(multimodal_model, node_emb, kg_embeddings_dict, device)


def batch_result_entry(image_name, predictions):
    """One element of the reference's ``batch_results.json`` [:350-357]."""
    probs = predictions["mask_prob"]
    return {
        "image": image_name,
        "prediction": "Camouflaged" if predictions["mask_pred"] == 1 else "Not Camouflaged",
        "pred_label": predictions["mask_pred"],
        "camo_prob": float(probs[0, 1]),
        "not_camo_prob": float(probs[0, 0]),
        "score": predictions["score"],
    }


def predict_embedding_directory(multimodal_model, rg_embeddings, kg_embeddings_dict, output_dir, device, max_images=None):
    """Batch mode over precomputed RG embeddings {image: {'node_embeddings': ...}} -> batch_results.json."""
    os.makedirs(output_dir, exist_ok=True)
    results = []
    for i, (name, rg) in enumerate(rg_embeddings.items()):
        if max_images is not None and i >= max_images:
            break
        pred, _, _ = predict_from_embeddings(multimodal_model, rg["node_embeddings"], kg_embeddings_dict, device)
        results.append(batch_result_entry(name, pred))
    with open(os.path.join(output_dir, "batch_results.json"), "w") as f:
        json.dump(results, f, indent=2)
    return results
