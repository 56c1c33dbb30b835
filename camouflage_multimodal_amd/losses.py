"""Losses of the reference trainer (models/multimodal/train_multimodal.py).

``multitask_loss`` is the measured path: one HIP kernel computes, per sample, the four
weighted terms of train_multimodal.py:256-268, their gradient w.r.t. the model outputs and
the arg-max prediction -- no host synchronisation.

``AggressiveFocalLoss`` keeps the reference class's name and call signature
(train_multimodal.py:29-57) for code that imports it; it is a few elementwise torch ops on a
[B, C] tensor and is not on the native training path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .engine import _on, _ptr, _stream_ptr, check_labels

LOSS_WEIGHTS = dict(mask=3.0, instance=1.0, edge=0.5, score=0.3)   # train_multimodal.py:257,260,263,266


def multitask_loss(outs, mask_label, edge_label, score_label, num_classes=2, want_pred=True, pre_activation=False):
    """outs [B, 2C+2] (as FusionEngine.forward_raw returns them); labels: int64 [B], float [B], float [B].
    Returns (loss_terms [B,4] weighted, d_outs [B,2C+2], pred int32 [B]).  With ``pre_activation`` the
    score column of d_outs is taken w.r.t. the score head's pre-sigmoid value (what
    ``FusionEngine.backward_raw(..., pre_activation=True)`` expects)."""
    _lib.require_device(outs, "outs")
    B = outs.shape[0]
    dev = outs.device
    check_labels(mask_label, num_classes)
    y = mask_label.to(device=dev, dtype=torch.int64).contiguous()
    e = edge_label.to(device=dev, dtype=torch.float32).contiguous()
    s = score_label.to(device=dev, dtype=torch.float32).contiguous()
    terms = torch.empty(B, 4, dtype=torch.float32, device=dev)
    d_outs = torch.empty_like(outs)
    pred = torch.empty(B, dtype=torch.int32, device=dev) if want_pred else None
    with _on(dev):
        rc = _lib.lib().camo_loss(_ptr(outs), _ptr(y), _ptr(e), _ptr(s), B, num_classes, _ptr(terms),
                                  _ptr(None if pre_activation else d_outs), _ptr(d_outs if pre_activation else None),
                                  _ptr(pred), _stream_ptr(dev))
    _lib.check(rc, "camo_loss")
    return terms, d_outs, pred


class AggressiveFocalLoss(nn.Module):
    """alpha_t * (1 - p_t)^gamma * CE, mean over the batch; alpha_t = alpha for class 1."""

    def __init__(self, alpha=0.75, gamma=3.0):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    def forward(self, inputs, targets):
        logp = F.log_softmax(inputs, dim=1)
        logpt = logp.gather(1, targets.unsqueeze(1)).squeeze(1)
        pt = logpt.exp()
        alpha_t = torch.where(targets == 1, self.alpha, 1 - self.alpha)
        return (alpha_t * (1 - pt) ** self.gamma * (-logpt)).mean()
