"""Fused global-norm clip + AdamW over the model's flat buffers (two kernel launches per
step, no host synchronisation).

Stands behind ``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)`` followed
by ``torch.optim.AdamW(lr, weight_decay).step()`` as the reference trainer calls them
(models/multimodal/train_multimodal.py:278-279, :403-407) and behind its
``CosineAnnealingWarmRestarts(T_0=10, T_mult=2)`` schedule stepped once per epoch (:409-411, :439).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .engine import _on, _ptr, _stream_ptr


def cosine_warm_restarts_lr(base_lr, epoch, T_0=10, T_mult=2, eta_min=0.0):
    """Learning rate in effect during 0-based ``epoch``."""
    t, Ti = epoch, T_0
    while t >= Ti:
        t -= Ti
        Ti *= T_mult
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / Ti)) / 2


class FusedClipAdamW:
    def __init__(self, model, lr=5e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
        self.engine = model._engine
        self.base_lr = self.lr = float(lr)
        self.weight_decay, self.betas, self.eps, self.max_norm = float(weight_decay), tuple(betas), float(eps), float(max_norm)
        self.step_count = 0
        self._m = self._v = self._sumsq = None

    def _state(self):
        p = self.engine.flat_params
        if self._m is None or self._m.data_ptr() == 0 or self._m.device != p.device or self._m.numel() != p.numel():
            self._m = torch.zeros_like(p)
            self._v = torch.zeros_like(p)
            self._sumsq = torch.zeros(_lib.SUMSQ_FLOATS, dtype=torch.float32, device=p.device)
        return self._m, self._v, self._sumsq

    def zero_grad(self):
        self.engine.ensure_flat_grads().zero_()

    def set_epoch(self, epoch, T_0=10, T_mult=2):
        self.lr = cosine_warm_restarts_lr(self.base_lr, epoch, T_0, T_mult)
        return self.lr

    def step(self, allreduce=None, zero_grads=False, shadows=False):
        """``allreduce``: optional callable applied to the flat gradient buffer before the norm
        (data-parallel SUM, see ddp.py).  ``zero_grads``: clear the gradient buffer in the same pass
        (the next minibatch's ``zero_grad()`` fused in) instead of leaving the clipped gradients in
        it.  ``shadows``: also leave the fused schedule's bf16 weight shadows of the updated parameters in the engine's
        persistent buffer (camo_clip_adamw_shadows), so that the next ``train_raw(use_shadows=True)`` needs no shadow
        launch.  Returns nothing; ``grad_norm()`` reads the norm lazily."""
        eng = self.engine
        _lib.require_device(eng.flat_params, "model parameters")
        # param.grad views are attached once; walking named_parameters() every step cost ~130 us of host time
        g = eng.ensure_flat_grads(attach=not getattr(eng, "_grads_attached", False))
        if allreduce is not None:
            allreduce(g)
        m, v, ss = self._state()
        self.step_count += 1
        L = _lib.lib()
        with _on(eng.device):
            st = _stream_ptr(eng.device)
            _lib.check(L.camo_grad_sumsq(_ptr(g), g.numel(), _ptr(ss), st), "camo_grad_sumsq")
            sh = eng.shadow_buffer() if shadows else None
            if sh is not None:
                _lib.check(L.camo_clip_adamw_shadows(C.byref(eng.dims), eng._ptab, _ptr(eng.flat_params), _ptr(g), _ptr(m), _ptr(v), g.numel(),
                                                     _ptr(ss), self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps,
                                                     self.weight_decay, self.step_count, int(bool(zero_grads)), _ptr(sh), st),
                           "camo_clip_adamw_shadows")
                eng._shadows_version = eng.param_version(); eng._shadows_full = True; eng._shadows_fold = False
                return
            eng._shadows_version = None
            _lib.check(L.camo_clip_adamw(_ptr(eng.flat_params), _ptr(g), _ptr(m), _ptr(v), g.numel(), _ptr(ss),
                                         self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps,
                                         self.weight_decay, self.step_count, int(bool(zero_grads)), st), "camo_clip_adamw")

    def grad_norm(self):
        """Pre-clip global gradient norm of the last step (device tensor)."""
        return self._state()[2][:1].sqrt()

    # ---- checkpoint interchange with torch.optim.AdamW (train_multimodal.py:467) -------------------
    def state_dict(self):
        m, v, _ = self._state()
        state = {}
        for idx, (_, name, o, n, shape) in enumerate(self.engine._layout):
            state[idx] = {"step": torch.tensor(float(self.step_count)),
                          "exp_avg": m[o:o + n].view(shape).clone(), "exp_avg_sq": v[o:o + n].view(shape).clone()}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=True, initial_lr=self.base_lr, params=list(range(len(self.engine._layout))))
        return {"state": state if self.step_count else {}, "param_groups": [group]}

    def load_state_dict(self, sd):
        m, v, _ = self._state()
        grp = sd["param_groups"][0]
        self.lr = float(grp["lr"]); self.base_lr = float(grp.get("initial_lr", grp["lr"]))
        self.betas = tuple(grp["betas"]); self.eps = float(grp["eps"]); self.weight_decay = float(grp["weight_decay"])
        steps = 0
        for idx, (_, name, o, n, shape) in enumerate(self.engine._layout):
            st = sd["state"].get(idx)
            if st is None:
                continue
            m[o:o + n].copy_(st["exp_avg"].reshape(-1)); v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps = int(float(st["step"]))
        self.step_count = steps
