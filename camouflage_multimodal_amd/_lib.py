"""ctypes binding of include/camo_fusion.h.

There is no CPU fallback: if the shared library is missing or the tensors are
not on a HIP device the callers raise.  Build with
``python -m camouflage_multimodal_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcamo_fusion.so")

ABI_VERSION = 10
FWD_INFERENCE = 1
FLAG_ATTN_MAPS = 2
SUMSQ_FLOATS = 257
FUSION_CROSS_ATTENTION, FUSION_LATE = 0, 1
PREC_F32, PREC_BF16 = 0, 1
NPARAMS_CROSS, NPARAMS_LATE = 44, 22

# every symbol include/camo_fusion.h declares
SYMBOLS = ("camo_abi_version", "camo_last_error", "camo_workspace_bytes", "camo_batch_desc_bytes", "camo_prepare_batch", "camo_gather_batch", "camo_forward", "camo_forward_cached", "camo_backward", "camo_forward_loss_backward",
           "camo_loss", "camo_grad_sumsq", "camo_clip_adamw", "camo_shadow_bytes", "camo_clip_adamw_shadows", "camo_debug_gemm", "camo_debug_gemm16", "camo_debug_ws_offset",
           "camo_options_init", "camo_options_set", "camo_debug_set_stamps", "camo_prof_begin", "camo_prof_end", "camo_prof_kind", "camo_tail_timeouts", "camo_tail_poison_to_grads")


# every symbol include/camo_rg_gnn.h declares
RG_SYMBOLS = ("camo_rg_workspace_bytes", "camo_rg_node_embeddings", "camo_rg_build_csr")
# every symbol include/camo_rg_features.h declares
RGF_SYMBOLS = ("camo_rg_graph_workspace_bytes", "camo_rg_region_graph")
RG_MAX_LABELS = 4096
RG_NPARAMS = 28


class CamoRgDims(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32)]


OPTION_NAMES = ("sched16", "fused", "tail17", "fused_rt", "wide2", "fused_one", "wide_front_rt", "tailw", "tailw_bwd", "param_space", "tn_big",
                "fused_variant", "back_lead", "tn_balance", "tn_kcap", "tn_exp", "exp", "fused_save", "tail_skip_arrival", "wide2_bwd")


class CamoOptions(C.Structure):
    """camo_options_t (include/camo_fusion.h): the schedule options of ONE engine -- caller-owned, reached through CamoDims.options."""
    _fields_ = [(n, C.c_int32) for n in OPTION_NAMES]


# camo_options_init's values (a CPU test holds the two to each other): an engine can be constructed before the library is loadable
OPTION_DEFAULTS = dict(sched16=-1, fused=-1, tail17=-1, fused_rt=-1, wide2=-1, fused_one=1, wide_front_rt=0, tailw=-1, tailw_bwd=-1, param_space=-1, tn_big=-1,
                       fused_variant=1, back_lead=1, tn_balance=1, tn_kcap=0, tn_exp=0, exp=0, fused_save=0, tail_skip_arrival=0, wide2_bwd=-1)


def default_options():
    o = CamoOptions()
    for k, v in OPTION_DEFAULTS.items():
        setattr(o, k, v)
    return o


class CamoDims(C.Structure):
    _fields_ = [("rg_dim", C.c_int32), ("kg_dim", C.c_int32), ("hidden_dim", C.c_int32), ("num_heads", C.c_int32),
                ("num_classes", C.c_int32), ("fusion_type", C.c_int32), ("dropout", C.c_float), ("options", C.POINTER(CamoOptions))]


class CamoError(RuntimeError):
    pass


_lib = None


def _set_option_everywhere(name, value):
    from . import engine
    return engine.set_option_all(name.decode() if isinstance(name, bytes) else name, int(value))


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CamoError(f"{LIB_PATH} is missing: the HIP extension has not been built and there is no CPU "
                        "fallback. Run `python -m camouflage_multimodal_amd.build` (needs hipcc).")
    L = C.CDLL(LIB_PATH)
    vp, i32, u64, f32, sz = C.c_void_p, C.c_int32, C.c_uint64, C.c_float, C.c_size_t
    L.camo_abi_version.restype = C.c_int
    L.camo_abi_version.argtypes = []
    L.camo_last_error.restype = C.c_char_p
    L.camo_last_error.argtypes = []
    L.camo_workspace_bytes.restype = sz
    L.camo_workspace_bytes.argtypes = [C.POINTER(CamoDims), i32, i32, i32]
    L.camo_batch_desc_bytes.restype = sz
    L.camo_batch_desc_bytes.argtypes = [i32, i32]
    L.camo_prepare_batch.restype = C.c_int
    L.camo_prepare_batch.argtypes = [vp, i32, i32, i32, vp, sz, vp]
    L.camo_gather_batch.restype = C.c_int
    L.camo_gather_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, f32, u64, vp]
    L.camo_forward.restype = C.c_int
    L.camo_forward.argtypes = [C.POINTER(CamoDims), vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, vp, vp, i32, u64, i32, i32, vp]
    L.camo_backward.restype = C.c_int
    L.camo_backward.argtypes = [C.POINTER(CamoDims), vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, vp, i32, i32, u64, i32, i32, vp]
    L.camo_loss.restype = C.c_int
    L.camo_loss.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.camo_grad_sumsq.restype = C.c_int
    L.camo_grad_sumsq.argtypes = [vp, sz, vp, vp]
    L.camo_clip_adamw.restype = C.c_int
    L.camo_clip_adamw.argtypes = [vp, vp, vp, vp, sz, vp, f32, f32, f32, f32, f32, f32, i32, i32, vp]
    L.camo_forward_loss_backward.restype = C.c_int
    L.camo_forward_loss_backward.argtypes = [C.POINTER(CamoDims), vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz,
                                             vp, vp, vp, vp, vp, vp, i32, C.c_uint64, i32, vp, vp, i32, vp]
    L.camo_forward_cached.restype = C.c_int
    L.camo_forward_cached.argtypes = [C.POINTER(CamoDims), vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, vp, vp, i32, u64, i32, i32, vp, i32,
                                      C.POINTER(C.c_int32), vp]
    L.camo_shadow_bytes.restype = sz
    L.camo_shadow_bytes.argtypes = [C.POINTER(CamoDims)]
    L.camo_clip_adamw_shadows.restype = C.c_int
    L.camo_clip_adamw_shadows.argtypes = [C.POINTER(CamoDims), vp, vp, vp, vp, vp, sz, vp, f32, f32, f32, f32, f32, f32, i32, i32, vp, vp]
    L.camo_rg_workspace_bytes.restype = sz
    L.camo_rg_workspace_bytes.argtypes = [C.POINTER(CamoRgDims), i32]
    L.camo_rg_build_csr.restype = C.c_int
    L.camo_rg_build_csr.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.camo_rg_node_embeddings.restype = C.c_int
    L.camo_rg_node_embeddings.argtypes = [C.POINTER(CamoRgDims), vp, vp, vp, vp, vp, i32, i32, vp, sz, vp, vp]
    L.camo_tail_timeouts.restype = C.c_int
    L.camo_tail_timeouts.argtypes = [vp]
    L.camo_tail_poison_to_grads.restype = C.c_int
    L.camo_tail_poison_to_grads.argtypes = [vp, vp]
    L.camo_rg_graph_workspace_bytes.restype = sz
    L.camo_rg_graph_workspace_bytes.argtypes = [i32]
    L.camo_rg_region_graph.restype = C.c_int
    L.camo_rg_region_graph.argtypes = [vp, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp, vp, i32, vp, vp]
    L.camo_debug_gemm.restype = C.c_int
    L.camo_debug_gemm.argtypes = [vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]
    L.camo_debug_gemm16.restype = C.c_int
    L.camo_debug_gemm16.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp]
    L.camo_debug_ws_offset.restype = C.c_int64
    L.camo_debug_ws_offset.argtypes = [C.POINTER(CamoDims), i32, i32, i32, C.c_char_p]
    L.camo_options_init.restype = C.c_int
    L.camo_options_init.argtypes = [C.POINTER(CamoOptions)]
    L.camo_options_set.restype = C.c_int
    L.camo_options_set.argtypes = [C.POINTER(CamoOptions), C.c_char_p, i32]
    # tests and developer tools switch schedules "for the process": a Python-side convenience that sets the option on every live
    # engine and on the defaults of engines created later (engine.set_option_all) -- the library itself keeps no option state
    L.camo_debug_set_option = _set_option_everywhere
    L.camo_debug_set_stamps.restype = C.c_int
    L.camo_debug_set_stamps.argtypes = [vp, i32]
    L.camo_prof_begin.restype = C.c_int
    L.camo_prof_begin.argtypes = [i32]
    L.camo_prof_end.restype = C.c_int
    L.camo_prof_end.argtypes = [C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(C.c_double)]
    L.camo_prof_kind.restype = C.c_int
    L.camo_prof_kind.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(i32), C.POINTER(C.c_double)]
    v = L.camo_abi_version()
    if v != ABI_VERSION:
        raise CamoError(f"libcamo_fusion.so has ABI version {v}, this package expects {ABI_VERSION}: rebuild it")
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().camo_last_error().decode("utf-8", "replace")
        raise CamoError(f"{what or 'camo call'} failed (code {rc}): {msg}")


def require_device(t, name):
    """The product path runs on a HIP device only."""
    if not t.is_cuda:
        raise CamoError(f"{name} is on {t.device}: the fusion path runs as HIP kernels on an MI355X and has no CPU "
                        "fallback (move the model and its inputs to 'cuda').")


def tail_timeouts(device=None):
    """Number of arrival waits of the one-launch tail kernel that gave up on ``device`` (default: the current HIP device) since the
    library was loaded (synchronous; see camo_tail_timeouts in include/camo_fusion.h).  A step that hit one is not applied: its
    loss terms are NaN and the optimizer kernels skip an update whose gradient norm is not finite."""
    n = C.c_uint32(0)
    if device is not None:
        import torch
        with torch.cuda.device(device):
            check(lib().camo_tail_timeouts(C.byref(n)), "camo_tail_timeouts")
    else:
        check(lib().camo_tail_timeouts(C.byref(n)), "camo_tail_timeouts")
    return int(n.value)
