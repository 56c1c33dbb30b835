"""Host-side engine: owns the flat parameter / gradient buffers, the workspace and the
ctypes calls into ``libcamo_fusion.so``.  PyTorch is used for device memory, streams and
autograd plumbing only; no arithmetic of the path happens in torch ops.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib

# parameter slots in the order of include/camo_fusion.h (= the reference state_dict order)
_HEADS = [f"{h}.{i}.{k}" for h in ("mask_head", "instance_head", "edge_head", "score_head")
          for i, k in ((0, "weight"), (0, "bias"), (3, "weight"), (3, "bias"))]
CROSS_SLOTS = (
    ["fusion.rg_proj.weight", "fusion.rg_proj.bias", "fusion.kg_proj.weight", "fusion.kg_proj.bias"]
    + [f"fusion.{a}.{k}" for a in ("cross_attn_rg2kg", "cross_attn_kg2rg")
       for k in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")]
    + ["fusion.ln_rg.weight", "fusion.ln_rg.bias", "fusion.ln_kg.weight", "fusion.ln_kg.bias"]
    + [f"fusion.{f}.{i}.{k}" for f in ("ffn_rg", "ffn_kg", "fusion_layer") for i in (0, 3) for k in ("weight", "bias")]
    + _HEADS)
LATE_SLOTS = [f"fusion.fusion.{i}.{k}" for i in (0, 3, 6) for k in ("weight", "bias")] + _HEADS
assert len(CROSS_SLOTS) == _lib.NPARAMS_CROSS and len(LATE_SLOTS) == _lib.NPARAMS_LATE

_PREC = {"f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device=None):
    """The current HIP stream of ``device`` (default: the current device) as a void*."""
    if _raw_stream is not None:                              # (a plain C call: ~0.3 us against ~8 us for torch.cuda.current_stream())
        idx = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
        return C.c_void_p(_raw_stream(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _NullCtx:
    def __enter__(self): return None
    def __exit__(self, *a): return False


_NULL = _NullCtx()


def _on(device):
    """Context that makes ``device`` the current HIP device for the launches of a C-ABI call (they go to the stream
    handed over, but the library's hipFuncSetAttribute / hipGetDevice calls and torch's allocations must see the device
    that owns the pointers).  Free when it already is current."""
    if device.type != "cuda" or torch.cuda.current_device() == device.index:
        return _NULL
    return torch.cuda.device(device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def check_labels(mask_label, num_classes):
    """torch's cross_entropy raises on a target outside [0, C); the loss kernel would index with it.  Labels that are
    still on the host (the usual case: they come out of the data loader) are checked here for free; labels already
    on the device are checked by the kernel, which poisons that sample's loss and gradients with NaN."""
    if not mask_label.is_cuda and mask_label.numel():
        lo, hi = int(mask_label.min()), int(mask_label.max())
        if lo < 0 or hi >= num_classes:
            raise IndexError(f"Target {lo if lo < 0 else hi} is out of bounds for {num_classes} classes")


class Batch:
    """A packed minibatch on the device, ready for the C ABI."""
    __slots__ = ("rg", "kg", "offsets", "desc", "nrs", "B", "T", "Nk", "max_nr")

    def __init__(self, rg, kg, offsets, desc, nrs):
        self.rg, self.kg, self.offsets, self.desc, self.nrs = rg, kg, offsets, desc, nrs
        self.B, self.T, self.Nk, self.max_nr = len(nrs), rg.shape[0], kg.shape[1], max(nrs)


# Schedule options are per ENGINE (camo_options_t, owned here, reached by the library through dims.options).  Tests and developer
# tools that want "the whole process" on one schedule go through set_option_all: every live engine + the defaults of later ones.
_LIVE_ENGINES = weakref.WeakSet()
_OPTION_DEFAULTS = {}


def set_option_all(name, value):
    """Set one schedule option (include/camo_fusion.h, camo_options_t) on every live engine and for engines created from now on.
    Returns 0, or raises CamoError for an unknown name."""
    if name not in _lib.OPTION_NAMES:
        raise _lib.CamoError(f"unknown option {name}")
    if name == "tail_skip_arrival":
        raise _lib.CamoError("tail_skip_arrival is a one-shot hook of ONE engine's next call: use engine.set_option")
    _OPTION_DEFAULTS[name] = int(value)
    for eng in list(_LIVE_ENGINES):
        eng.set_option(name, value)
    return 0


class FusionEngine:
    def __init__(self, module):
        self._mod = weakref.ref(module)
        c = module.config
        self.cross = c["fusion_type"] == "cross_attention"
        self.slots = CROSS_SLOTS if self.cross else LATE_SLOTS
        self.options = _lib.default_options()                  # this engine's schedule options: nothing of the kind lives in the library
        self.dims = _lib.CamoDims(c["rg_dim"], c["kg_dim"], c["hidden_dim"], c["num_heads"], c["num_classes"],
                                  _lib.FUSION_CROSS_ATTENTION if self.cross else _lib.FUSION_LATE, c["dropout"], C.pointer(self.options))
        for k, v in _OPTION_DEFAULTS.items():
            setattr(self.options, k, v)
        _LIVE_ENGINES.add(self)
        self.out_width = 2 * c["num_classes"] + 2
        self.flat_params = None
        self.flat_grads = None
        self._layout = None           # [(slot index, name, offset, numel, shape)]
        self._ptab = None             # ctypes array of parameter pointers
        self._gtab = None
        self._offsets_cache = {}
        self._desc_bytes = {}
        self._ws_bytes = {}
        self._ws = None
        self._shadows = None          # persistent bf16 weight shadows of the fused schedule (camo_shadow_bytes)
        self._shadows_version = None  # param_version() right after the call that left them current (optimizer step or forward)
        self._shadows_full = False    # they include the transposed set the backward needs (an inference call builds the forward set only)
        self._shadows_fold = False    # they include the RG rows' folded in-projection (built by inference calls only; the optimizer's set lacks it)
        self._plist = None
        self._seed_base = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._calls = 0
        self.reflatten()

    # ------------------------------------------------------------------ flat buffers
    def module(self):
        m = self._mod()
        if m is None:
            raise RuntimeError("model was garbage-collected")
        return m

    def reflatten(self):
        """(Re)build the flat fp32 parameter buffer and point every nn.Parameter at its slice.
        Called at construction and after every nn.Module._apply (.to(), .cuda(), ...)."""
        mod = self.module()
        named = dict(mod.named_parameters())
        extra = set(named) - set(self.slots)
        if extra:
            raise RuntimeError(f"unexpected parameters {sorted(extra)}")
        layout, off = [], 0
        for i, name in enumerate(self.slots):
            p = named.get(name)
            if p is None:
                continue                    # nn.Identity projection: slot stays NULL
            layout.append((i, name, off, p.numel(), tuple(p.shape)))
            off += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        dev = next(iter(named.values())).device
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for i, name, o, n, shape in layout:
            p = named[name]
            flat[o:o + n].copy_(p.data.reshape(-1).to(torch.float32))
            p.data = flat[o:o + n].view(shape)
            p.grad = None
        self.flat_params, self._layout = flat, layout
        self.flat_grads = None
        self._ptab = self._gtab = None
        self._ws = None
        self._shadows = None
        self._shadows_version = None
        self._plist = None
        self._offsets_cache = {}
        self._desc_bytes = {}
        self._ws_bytes = {}
        if dev.type == "cuda":
            n = len(self.slots)
            tab = (C.c_void_p * n)()
            for i, name, o, cnt, shape in layout:
                tab[i] = flat.data_ptr() + 4 * o
            self._ptab = tab

    def ensure_flat_grads(self, attach=True):
        """Persistent flat gradient buffer (native training mode); ``param.grad`` become views of it."""
        if self.flat_grads is None:
            self.flat_grads = torch.zeros_like(self.flat_params)
            self._gtab = self._grad_table(self.flat_grads)
            self._grads_attached = False
        if attach:
            self._grads_attached = True
            named = dict(self.module().named_parameters())
            for i, name, o, n, shape in self._layout:
                p = named[name]
                if p.grad is None or p.grad.data_ptr() != self.flat_grads.data_ptr() + 4 * o:
                    p.grad = self.flat_grads[o:o + n].view(shape)
        return self.flat_grads

    def tail_grad_offset(self):
        """Offset (in floats) of the first gradient of the per-sample tail's contiguous run in the flat buffer --
        ``fusion.ffn_kg.3.weight`` (CAMO_P_F2_W3) .. end -- or None when the model has no such run (late fusion)."""
        for i, name, o, n, shape in self._layout:
            if name == "fusion.ffn_kg.3.weight":
                return o
        return None

    def _grad_table(self, gflat):
        tab = (C.c_void_p * len(self.slots))()
        for i, name, o, n, shape in self._layout:
            tab[i] = gflat.data_ptr() + 4 * o
        return tab

    # ------------------------------------------------------------------ batches
    def _require_ready(self):
        _lib.lib()
        _lib.require_device(self.flat_params, "model parameters")
        if self._ptab is None:
            self.reflatten()

    @property
    def device(self):
        return self.flat_params.device

    def _same_device(self, t, name):
        """Raw device pointers cross the ABI: a tensor on another GPU than the parameters would make the kernels
        dereference a foreign address (a memory fault, not a Python error), so refuse it here."""
        _lib.require_device(t, name)
        if t.device != self.flat_params.device:
            raise _lib.CamoError(f"{name} is on {t.device} but the model parameters are on {self.flat_params.device}: "
                                 "every tensor of a call must live on the model's device")
        return t

    def fold_rank(self, rank):
        """Data parallelism: give every rank its own dropout mask stream (the mask index of an element is its LOCAL
        position in the rank's packed batch, so identically seeded ranks would draw identical masks)."""
        self._seed_base = (int(torch.initial_seed()) ^ ((int(rank) + 1) * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF

    def set_option(self, name, value):
        """One schedule option of THIS engine (developer / test switch; product code leaves the defaults)."""
        _lib.check(_lib.lib().camo_options_set(C.byref(self.options), name.encode(), int(value)), "camo_options_set")

    def make_batch(self, rg_packed, nrs, kg):
        """rg_packed [T, rg_dim], nrs: host ints, kg [B, Nk, kg_dim] -> Batch (device, fp32, contiguous)."""
        self._require_ready()
        self._same_device(rg_packed, "rg_embeddings")
        self._same_device(kg, "kg_embeddings")
        dev_offs = getattr(nrs, "offsets_dev", None)            # DeviceResidentDataset.batch: the offsets already exist on the device
        nrs_obj = nrs if dev_offs is not None else None         # (its descriptor is kept ON the object: a prebuilt validation batch is reused every epoch)
        nrs = [int(n) for n in nrs]
        B = len(nrs)
        if B < 1 or min(nrs) < 1:
            raise ValueError("every sample needs at least one RG node")
        if rg_packed.dim() != 2 or rg_packed.shape[0] != sum(nrs) or rg_packed.shape[1] != self.dims.rg_dim:
            raise RuntimeError(f"rg embeddings of shape {tuple(rg_packed.shape)} do not match {sum(nrs)} rows x rg_dim {self.dims.rg_dim}")
        if kg.dim() != 3 or kg.shape[0] != B or kg.shape[2] != self.dims.kg_dim:
            raise RuntimeError(f"kg embeddings of shape {tuple(kg.shape)} do not match batch {B} x kg_dim {self.dims.kg_dim}")
        if rg_packed.dtype is not torch.float32 or rg_packed.requires_grad or not rg_packed.is_contiguous():
            rg_packed = rg_packed.detach().to(torch.float32).contiguous()
        if kg.dtype is not torch.float32 or kg.requires_grad or not kg.is_contiguous():
            kg = kg.detach().to(torch.float32).contiguous()
        key = None if dev_offs is not None else tuple(nrs)
        desc = self._offsets_cache.get(key) if key is not None else getattr(nrs_obj, "_desc", None)
        if desc is not None and (desc[0].device != rg_packed.device or (nrs_obj is not None and desc[0] is not dev_offs)):
            desc = None
        if desc is None:
            # batch descriptor (row offsets + the library's opaque row/tile maps): ONE launch.  Repeated shape tuples (a fixed
            # validation set, a benchmark's minibatches) are cached; a training epoch's minibatches each have their own tuple
            # and come with device-built offsets, so nothing is copied from the host and nothing is kept.
            T = sum(nrs)
            if dev_offs is not None:
                offs = dev_offs
                if offs.dtype != torch.int32 or offs.numel() != B + 1 or offs.device != rg_packed.device:
                    raise _lib.CamoError("offsets_dev must be an int32 tensor of B + 1 elements on the batch's device")
            else:
                host = torch.zeros(B + 1, dtype=torch.int32)
                host[1:] = torch.tensor(nrs, dtype=torch.int32).cumsum(0)
                offs = host.to(rg_packed.device, non_blocking=False)
            nbytes = self._desc_bytes.get((B, T))
            if nbytes is None:
                nbytes = _lib.lib().camo_batch_desc_bytes(B, T)
                if nbytes == 0:
                    _lib.check(-1, "camo_batch_desc_bytes")
                if len(self._desc_bytes) > 4096:
                    self._desc_bytes.clear()
                self._desc_bytes[(B, T)] = nbytes
            buf = torch.empty(nbytes, dtype=torch.uint8, device=rg_packed.device)
            with _on(self.device):
                _lib.check(_lib.lib().camo_prepare_batch(_ptr(offs), B, T, max(nrs), _ptr(buf), nbytes,
                                                         _stream_ptr(self.device)), "camo_prepare_batch")
            desc = (offs, buf)
            if key is not None:
                if len(self._offsets_cache) > 1024:
                    self._offsets_cache.clear()
                self._offsets_cache[key] = desc
            elif nrs_obj is not None:
                try:
                    nrs_obj._desc = desc
                except AttributeError:
                    pass
        return Batch(rg_packed, kg, desc[0], desc[1], nrs)

    def workspace(self, batch, private=False):
        need = self._ws_bytes.get((batch.B, batch.T, batch.Nk))
        if need is None:
            need = _lib.lib().camo_workspace_bytes(C.byref(self.dims), batch.B, batch.T, batch.Nk)
            if need == 0:
                _lib.check(-1, "camo_workspace_bytes")
            if len(self._ws_bytes) > 4096:
                self._ws_bytes.clear()
            self._ws_bytes[(batch.B, batch.T, batch.Nk)] = need
        if private:
            return torch.empty(need, dtype=torch.uint8, device=batch.rg.device)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=batch.rg.device)
        return self._ws

    def shadow_buffer(self):
        """Caller-owned weight shadows for camo_forward_loss_backward / camo_clip_adamw_shadows, or None when the model's
        configuration has no fused schedule (or the model is not in bf16 mode)."""
        if self.module().precision != "bf16" or self.flat_params.device.type != "cuda":
            return None
        if self._shadows is None:
            need = _lib.lib().camo_shadow_bytes(C.byref(self.dims))
            if need == 0:
                return None
            self._shadows = torch.empty(need, dtype=torch.uint8, device=self.flat_params.device)
            self._shadows_version = None
        return self._shadows

    def param_version(self):
        """Torch's in-place version counters of the flat buffer and of every parameter (a parameter written through
        ``copy_`` / ``load_state_dict`` / any in-place op bumps its own counter, not the flat buffer's; the library's kernels
        bump none).  Writes through ``param.data`` are invisible to torch and therefore to this."""
        if self._plist is None:                               # (walking the module tree costs ~30 us of host time per call)
            self._plist = list(self.module().parameters())
        return (self.flat_params._version, sum(p._version for p in self._plist))

    def invalidate_shadows(self):
        """Forget that the persistent weight shadows match the parameters.  For writers torch's version counters do not see:
        ``param.data`` / ``flat_params.data`` writes, raw-pointer writers, collectives that fill the flat buffer in place
        (``ddp.broadcast_parameters`` calls this)."""
        self._shadows_version = None

    def shadows_current(self):
        """True when the shadows were left by the last optimizer call AND nothing on the torch side has written the parameters
        since."""
        return self._shadows is not None and self._shadows_version is not None and self._shadows_version == self.param_version()

    def next_seed(self):
        self._calls += 1
        return (self._seed_base + 0x9E3779B97F4A7C15 * self._calls) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ raw calls
    def forward_raw(self, batch, ws, training, seed, want_attention=False, outs=None, inference=False, cache_shadows=True):
        """``inference``: no backward_raw will follow on this workspace (lets the library skip what it would save).
        ``cache_shadows=False``: an inference call that keeps its weight shadows in ``ws`` like any other call (tests read them there)."""
        mod = self.module()
        if outs is None:
            outs = torch.empty(batch.B, self.out_width, dtype=torch.float32, device=batch.rg.device)
        a1 = a2 = None
        if want_attention and self.cross:
            a1 = torch.empty(batch.T, batch.Nk, dtype=torch.float32, device=batch.rg.device)
            a2 = torch.empty(batch.T, batch.Nk, dtype=torch.float32, device=batch.rg.device)
        # inference calls keep the fused schedule's weight shadows in the engine's persistent buffer: a validation / prediction
        # loop builds them once per parameter change instead of once per call (the library says what it left there)
        sh = self.shadow_buffer() if (inference and cache_shadows and a1 is None) else None
        with _on(self.device):
            if sh is not None:
                # (1 = current including the folded in-projection, which only inference calls build; 2 = current but left by the
                # optimizer call: this call adds the fold and leaves the rest -- the transposed set included -- alone)
                valid = (1 if self._shadows_fold else 2) if self.shadows_current() else 0
                state = C.c_int32(0)
                rc = _lib.lib().camo_forward_cached(C.byref(self.dims), self._ptab, _ptr(batch.rg), _ptr(batch.offsets),
                                                    _ptr(batch.desc), _ptr(batch.kg), batch.B, batch.T, batch.Nk, batch.max_nr, _ptr(ws), ws.numel(),
                                                    _ptr(outs), None, None, int(bool(training)), seed, _PREC[mod.precision],
                                                    _lib.FWD_INFERENCE, _ptr(sh), valid, C.byref(state), _stream_ptr(self.device))
                if rc == 0 and state.value:
                    self._shadows_full = state.value == 2 or bool(valid and self._shadows_full)
                    self._shadows_version = self.param_version()
                    self._shadows_fold = True
            else:
                rc = _lib.lib().camo_forward(C.byref(self.dims), self._ptab, _ptr(batch.rg), _ptr(batch.offsets),
                                             _ptr(batch.desc), _ptr(batch.kg), batch.B, batch.T, batch.Nk, batch.max_nr, _ptr(ws), ws.numel(), _ptr(outs),
                                             _ptr(a1), _ptr(a2), int(bool(training)), seed, _PREC[mod.precision],
                                             _lib.FWD_INFERENCE if inference else 0, _stream_ptr(self.device))
        _lib.check(rc, "camo_forward")
        return outs, ((a1, a2) if a1 is not None else None)

    def backward_raw(self, batch, ws, outs, d_outs, training, seed, gtab, pre_activation=False, had_attention=False):
        """``had_attention``: the forward_raw call that filled ``ws`` was asked for attention maps."""
        mod = self.module()
        self._same_device(d_outs, "d_outs")
        with _on(self.device):
            rc = _lib.lib().camo_backward(C.byref(self.dims), self._ptab, gtab, _ptr(batch.rg), _ptr(batch.offsets),
                                          _ptr(batch.desc), _ptr(batch.kg), batch.B, batch.T,
                                          batch.Nk, batch.max_nr, _ptr(ws), ws.numel(), _ptr(outs), _ptr(d_outs),
                                          int(bool(pre_activation)), int(bool(training)), seed, _PREC[mod.precision],
                                          _lib.FLAG_ATTN_MAPS if had_attention else 0, _stream_ptr(self.device))
        _lib.check(rc, "camo_backward")

    def train_raw(self, batch, ws, mask_label, edge_label, score_label, training, seed, gtab, tail_event=None, use_shadows=False):
        """The native training call (camo_forward_loss_backward): forward, the reference's 4-term loss and backward into
        the flat gradient buffer in one library call.  Returns (outs [B, 2C+2], loss_terms [B, 4], pred int32 [B]).
        ``tail_event``: raw hipEvent_t handle (int) recorded when the per-sample tail's gradients are final
        (``tail_grad_offset()`` .. end of the flat buffer): the hook for overlapping the data-parallel all-reduce.
        ``use_shadows``: keep the fused schedule's weight shadows in the engine's persistent buffer and skip rebuilding them
        when the optimizer left them current (``FusedClipAdamW.step(shadows=True)``)."""
        mod = self.module()
        dev = batch.rg.device
        check_labels(mask_label, self.dims.num_classes)
        ok = lambda t, dt: t.dtype is dt and t.device == dev and t.is_contiguous()        # (the data loader's labels usually are)
        y = mask_label if ok(mask_label, torch.int64) else mask_label.to(device=dev, dtype=torch.int64).contiguous()
        e = edge_label if ok(edge_label, torch.float32) else edge_label.to(device=dev, dtype=torch.float32).contiguous()
        s = score_label if ok(score_label, torch.float32) else score_label.to(device=dev, dtype=torch.float32).contiguous()
        outs = torch.empty(batch.B, self.out_width, dtype=torch.float32, device=dev)
        terms = torch.empty(batch.B, 4, dtype=torch.float32, device=dev)
        pred = torch.empty(batch.B, dtype=torch.int32, device=dev)
        sh = self.shadow_buffer() if use_shadows else None
        valid = int(sh is not None and self.shadows_current() and self._shadows_full)
        self._shadows_version = None                         # (whatever happens next, they are consumed: the optimizer renews them)
        with _on(self.device):
            rc = _lib.lib().camo_forward_loss_backward(
                C.byref(self.dims), self._ptab, gtab, _ptr(batch.rg), _ptr(batch.offsets), _ptr(batch.desc),
                _ptr(batch.kg), batch.B, batch.T, batch.Nk, batch.max_nr, _ptr(ws), ws.numel(), _ptr(y), _ptr(e), _ptr(s),
                _ptr(outs), _ptr(terms), _ptr(pred), int(bool(training)), seed, _PREC[mod.precision],
                C.c_void_p(tail_event) if tail_event else None, _ptr(sh) if sh is not None else None, valid, _stream_ptr(self.device))
        _lib.check(rc, "camo_forward_loss_backward")
        return outs, terms, pred

    # ------------------------------------------------------------------ autograd (drop-in) mode
    def forward_autograd(self, rg_packed, nrs, kg, want_attention=False):
        mod = self.module()
        batch = self.make_batch(rg_packed, nrs, kg)
        named = dict(mod.named_parameters())
        params = [named[name] for _, name, _, _, _ in self._layout]
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        seed = self.next_seed()
        if not needs_grad:
            return self.forward_raw(batch, self.workspace(batch), mod.training, seed, want_attention, inference=True)
        ws = self.workspace(batch, private=True)   # saved activations live until this call's backward
        res = _FusionFn.apply(self, batch, ws, mod.training, seed, want_attention and self.cross, *params)
        if want_attention and self.cross:
            return res[0], (res[1], res[2])
        return res, None


class _FusionFn(torch.autograd.Function):
    """Lets ``loss.backward()`` of an unmodified reference training loop drive camo_backward."""

    @staticmethod
    def forward(ctx, eng, batch, ws, training, seed, want_attention, *params):
        outs, attn = eng.forward_raw(batch, ws, training, seed, want_attention)
        ctx.eng, ctx.batch, ctx.ws, ctx.training, ctx.seed, ctx.had_attention = eng, batch, ws, training, seed, bool(want_attention)
        ctx.save_for_backward(outs)
        ctx.n_params = len(params)
        if attn is not None:
            ctx.mark_non_differentiable(*attn)
            return outs, attn[0], attn[1]
        return outs

    @staticmethod
    def backward(ctx, d_outs, *_):
        eng = ctx.eng
        (outs,) = ctx.saved_tensors
        g = torch.zeros_like(eng.flat_params)
        eng.backward_raw(ctx.batch, ctx.ws, outs, d_outs.contiguous().to(torch.float32), ctx.training, ctx.seed,
                         eng._grad_table(g), had_attention=ctx.had_attention)
        ctx.ws = None
        grads = [g[o:o + n].view(shape) for _, _, o, n, shape in eng._layout]
        return (None, None, None, None, None, None, *grads)
