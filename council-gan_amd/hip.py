"""ctypes binding of include/council_gan_hip.h (the C-ABI boundary).

There is NO fallback: if the shared library is missing, or a kernel is called without a GPU,
this raises.  The product path never routes through PyTorch compute ops or the oracle."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int8, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CG_LIB_PATH") or os.path.join(_HERE, "lib", "libcouncilgan_hip.so")      # CG_LIB_PATH: measurement builds (tools/)

ACT = {"none": 0, None: 0, "relu": 1, "lrelu": 2, "tanh": 3}
MAX_TAPS = 64
X3_LO_ELEMS = 32             # CG_X3_LO_ELEMS
X3_WSCALE = 1024.0      # CG_X3_WSCALE: power-of-two pre-scale of split-precision weights
SPLIT_STATE_FLOATS = 1026  # CG_SPLIT_STATE_FLOATS


class ConvGeom(ctypes.Structure):
    _fields_ = [("N", c_int32), ("H", c_int32), ("W", c_int32), ("C1", c_int32), ("C2", c_int32),
                ("up", c_int32), ("Ho", c_int32), ("Wo", c_int32), ("HoF", c_int32), ("WoF", c_int32),
                ("osy", c_int32), ("osx", c_int32), ("ooy", c_int32), ("oox", c_int32),
                ("stride", c_int32), ("T", c_int32), ("Cout", c_int32), ("act", c_int32),
                ("dy", c_int8 * MAX_TAPS), ("dx", c_int8 * MAX_TAPS)]


class Group(ctypes.Structure):
    """cg_group: n council members whose parameters sit `stride` fp32 elements apart in one pool (optim.ParamPool)."""
    _fields_ = [("n", c_int32), ("reserved", c_int32), ("stride", c_int64)]


class X3Epilogue(ctypes.Structure):
    """cg_x3_epilogue: bounded-split / fused activation-backward extras of the split-precision forward and data-gradient calls."""
    _fields_ = [("l1_ctl", c_void_p), ("in_state", c_void_p), ("in_nslots", c_int32), ("act_type", c_int32),
                ("act_src", c_void_p), ("out_state", c_void_p), ("addend", c_void_p)]


class Tuning(ctypes.Structure):
    """cg_tuning: the library's kernel-selection table (its only process-wide state)."""
    _fields_ = [(n, c_int32) for n in ("fwd_thin", "wgrad_thin", "wgrad_x3_bm256", "wgrad_x3_wide", "wgrad_x3_perm",
                                       "wgrad_legacy", "x3_wide", "x3_thin_out", "x3_korder", "tile_rows_scale",
                                       "no_amax_atomic", "wgrad_x3_multitap", "x3_cls_minor", "x3_generic_epilogue", "wgrad_xcd_group", "fp32_chunked_sum")]


class HipLibraryMissing(RuntimeError):
    pass


_lib = None

_P = c_void_p
_SIGS = {
    "cg_last_error": (c_char_p, []),
    "cg_version": (c_int, []),
    "cg_conv2d_fwd": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P]),
    "cg_conv2d_fwd_stats": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P, c_size_t, POINTER(c_int), _P]),
    "cg_instnorm_stats_from_partials": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "cg_split_f16": (c_int, [_P, _P, c_size_t, c_size_t, c_float, _P]),
    "cg_conv2d_fwd_x3": (c_int, [POINTER(ConvGeom), _P, c_size_t, _P, c_size_t, c_float, _P, _P, _P, _P, c_size_t, _P, c_size_t,
                                 POINTER(c_int), c_int, _P, POINTER(c_int), _P]),
    "cg_conv2d_fwd_x3_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_size_t, _P, c_size_t, c_float, _P, _P, _P, _P, _P,
                                   c_size_t, _P, c_size_t, POINTER(c_int), c_int, _P, POINTER(c_int), _P]),
    "cg_conv2d_fwd_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, _P, _P, _P, _P, _P, c_size_t, POINTER(c_int), _P,
                                POINTER(c_int), _P]),
    "cg_conv2d_fwd_thin_x3_ok": (c_int, [POINTER(ConvGeom)]),
    "cg_conv2d_fwd_thin_x3_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, _P, _P, _P, _P, _P, POINTER(c_int), _P]),
    "cg_split_f16_dynamic_capped": (c_int, [_P, _P, c_size_t, c_size_t, _P, c_int, c_float, _P]),
    "cg_conv2d_dgrad_x3_wt_elems": (c_size_t, [POINTER(ConvGeom), c_int]),
    "cg_conv2d_dgrad_x3_prep": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_int, c_int, c_float, _P, _P, c_size_t, _P]),
    "cg_conv2d_dgrad_x3_run": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_size_t, _P, _P, c_float, _P, c_int, c_int, _P, _P,
                                       POINTER(c_int), _P]),
    "cg_weight_l1_workspace": (c_size_t, [c_int, c_int]),
    "cg_weight_l1_bound": (c_int, [POINTER(Group), _P, c_int, c_int, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "cg_conv2d_fwd_x3_e": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_size_t, _P, c_size_t, c_float, _P, _P, _P, _P, _P,
                                   c_size_t, POINTER(X3Epilogue), c_int, POINTER(c_int), _P]),
    "cg_conv2d_dgrad_x3_run_e": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_size_t, _P, _P, c_float, _P, c_int, c_int, _P, _P,
                                         c_size_t, POINTER(X3Epilogue), _P, POINTER(c_int), _P]),
    "cg_compose1x1_fwd": (c_int, [POINTER(Group), _P, _P, _P, _P, c_int, _P, c_int, _P]),
    "cg_compose1x1_bwd": (c_int, [POINTER(Group), _P, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "cg_unsplit_f16": (c_int, [_P, c_size_t, _P, _P, c_size_t, _P]),
    "cg_upconv_wt_elems": (c_size_t, [c_int, c_int]),
    "cg_upconv_prep_x3": (c_int, [POINTER(Group), _P, c_int, c_int, c_float, _P, _P, _P, _P]),
    "cg_upconv2d_fwd_x3": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_size_t, _P, _P, c_float, _P, _P, _P, _P, c_size_t,
                                   POINTER(c_int), _P]),
    "cg_upconv_fold_dw": (c_int, [POINTER(Group), _P, _P, c_int, c_int, c_int, _P]),
    "cg_colsum_split_workspace": (c_size_t, [c_int, c_int]),
    "cg_colsum_split": (c_int, [POINTER(Group), _P, c_size_t, _P, ctypes.c_long, c_int, _P, c_int, _P, c_size_t, _P]),
    "cg_conv2d_wgrad_x3_ok_g": (c_int, [POINTER(ConvGeom), POINTER(Group)]),
    "cg_conv2d_wgrad_x3_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, c_size_t, _P, _P, c_size_t, _P, _P, _P, c_int, _P,
                                     c_size_t, _P]),
    "cg_conv2d_wgrad_workspace_g": (c_size_t, [POINTER(ConvGeom), POINTER(Group)]),
    "cg_conv2d_wgrad_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cg_conv2d_wgrad_act_ok": (c_int, [POINTER(ConvGeom)]),
    "cg_conv2d_wgrad_act_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, _P, _P, _P, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "cg_conv2d_dgrad_workspace_g": (c_size_t, [POINTER(ConvGeom), POINTER(Group), c_int]),
    "cg_conv2d_dgrad_g": (c_int, [POINTER(ConvGeom), POINTER(Group), _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "cg_lsgan_fwd_g": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "cg_lsgan_bwd_g": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "cg_focus_sums_g": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, c_size_t, _P]),
    "cg_focus_total_g": (c_int, [_P, c_size_t, c_int, c_float, c_float, c_float, c_int, c_int, _P, _P]),
    "cg_focus_bwd_g": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                               c_int, c_int, _P, _P]),
    "cg_adam_step_g": (c_int, [_P, _P, _P, _P, c_size_t, c_int, ctypes.c_longlong, c_float, c_float, c_float, c_float, c_float,
                               c_int, _P]),
    "cg_adam_hyper": (c_int, [c_float, c_float, c_float, c_int, POINTER(c_float)]),
    "cg_adam_step_dev": (c_int, [_P, _P, _P, _P, c_size_t, c_int, ctypes.c_longlong, c_float, c_float, c_float, c_float, _P, _P]),
    "cg_ring_push_dev": (c_int, [_P, c_int, _P, _P, c_int, _P]),
    "cg_loss_match_dev": (c_int, [_P, _P, c_int, _P, _P, _P, c_int, _P]),
    "cg_ring_push_g": (c_int, [_P, c_int, c_int, _P, c_int, _P]),
    "cg_loss_match_g": (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, _P]),
    "cg_gather_rows2": (c_int, [_P, _P, _P, _P, c_int, c_size_t, _P]),
    "cg_gen_total": (c_int, [_P, _P, _P, _P, c_float, c_float, _P, _P, _P, c_int, _P]),
    "cg_conv2d_fwd_amax": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, _P, POINTER(c_int), _P]),
    "cg_conv2d_wgrad_x3_ok": (c_int, [POINTER(ConvGeom)]),
    "cg_conv2d_wgrad_x3": (c_int, [POINTER(ConvGeom), _P, c_size_t, _P, _P, c_size_t, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cg_split_f16_dynamic": (c_int, [_P, _P, c_size_t, c_size_t, _P, c_int, _P]),
    "cg_act_bwd_split": (c_int, [_P, _P, c_size_t, c_int, _P, c_size_t, _P, c_int, _P, _P]),
    "cg_conv2d_dgrad_x3": (c_int, [POINTER(ConvGeom), _P, c_size_t, _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "cg_instnorm_apply_split": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, _P]),
    "cg_conv2d_fwd_tile": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, c_int, _P]),
    "cg_conv2d_wgrad_workspace": (c_size_t, [POINTER(ConvGeom)]),
    "cg_conv2d_wgrad": (c_int, [POINTER(ConvGeom), _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cg_tuning_get": (c_int, [POINTER(Tuning)]),
    "cg_tuning_set": (c_int, [POINTER(Tuning)]),
    "cg_conv2d_wgrad_legacy": (c_int, [c_int]),
    "cg_conv2d_wgrad_x3_bm256": (c_int, [c_int]),
    "cg_conv2d_wgrad_x3_wide": (c_int, [c_int]),
    "cg_conv2d_fwd_thin": (c_int, [c_int]),
    "cg_conv2d_wgrad_thin": (c_int, [c_int]),
    "cg_conv2d_dgrad_workspace": (c_size_t, [POINTER(ConvGeom), c_int]),
    "cg_conv2d_dgrad": (c_int, [POINTER(ConvGeom), _P, _P, c_int, c_int, _P, _P, c_size_t, _P]),
    "cg_weight_transpose": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_int32), c_int, _P]),
    "cg_act_fwd": (c_int, [_P, _P, c_size_t, c_int, _P]),
    "cg_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_int, _P]),
    "cg_instnorm_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cg_instnorm_stats": (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "cg_instnorm_apply": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cg_instnorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P, c_size_t, _P,
                                POINTER(c_int), _P]),
    "cg_decoder_head_fwd_x3": (c_int, [_P, c_size_t, _P, _P, _P, c_size_t, c_float, _P, _P, _P, _P, POINTER(Group), _P, _P, _P,
                                       ctypes.c_longlong, c_int, c_int, c_int, _P]),
    "cg_instnorm_bwd_split_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cg_instnorm_bwd_split": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                      _P, c_size_t, _P]),
    "cg_layernorm_workspace": (c_size_t, [c_int, c_int, c_int]),
    "cg_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "cg_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    "cg_avgpool3s2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cg_avgpool3s2_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cg_upsample2x_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cg_upsample2x_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cg_global_avgpool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "cg_mask_blend_fwd": (c_int, [_P, _P, _P, _P, c_size_t, c_int, c_int, _P]),
    "cg_mask_blend_bwd": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, _P]),
    "cg_lsgan_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, _P]),
    "cg_lsgan_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "cg_focus_workspace": (c_size_t, []),
    "cg_focus_sums": (c_int, [_P, c_int, c_int, c_int, c_int, c_float, c_float, _P, _P, c_size_t, _P]),
    "cg_focus_total": (c_int, [_P, c_size_t, c_float, c_float, c_float, c_int, c_int, _P, _P]),
    "cg_focus_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                             c_int, c_int, _P, _P]),
    "cg_l1_mean_fwd": (c_int, [_P, _P, c_size_t, _P, _P]),
    "cg_l1_mean_bwd": (c_int, [_P, _P, _P, c_size_t, _P, _P]),
    "cg_adam_step": (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "cg_u8_to_f32_nhwc": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, c_float, c_float, _P, _P]),
    "cg_x3_interleaved": (c_int, []),
    "cg_fill": (c_int, [_P, c_size_t, c_float, _P]),
    "cg_add": (c_int, [_P, _P, _P, c_size_t, _P]),
    "cg_axpby": (c_int, [c_float, _P, c_float, _P, c_size_t, _P]),
    "cg_gather_rows": (c_int, [_P, _P, _P, c_int, c_size_t, _P]),
    "cg_loss_match": (c_int, [_P, _P, c_int, c_int, _P, _P, _P]),
    "cg_ring_push": (c_int, [_P, c_int, c_int, _P, _P]),
    "cg_debug_fetch": (c_int, [POINTER(c_int64), c_int]),
    "cg_prof_enable": (c_int, [c_int]),
    "cg_prof_collect": (c_int, [POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    "cg_prof_slot_name": (c_char_p, [c_int]),
    "cg_prof_report": (c_char_p, []),
    "cg_comm_unique_id": (c_int, [_P]),
    "cg_comm_create": (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    "cg_comm_destroy": (c_int, [_P]),
    "cg_comm_info": (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    "cg_allgather_images": (c_int, [_P, _P, _P, c_size_t, _P]),
    "cg_allreduce_sum": (c_int, [_P, _P, c_size_t, _P]),
}
EXPORTS = tuple(_SIGS)


def load():
    """Load the library (no GPU needed for this step) and declare every signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found -- build it with `python council-gan_amd/build_hip.py` "
            "(there is no CPU/PyTorch fallback for the hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise HipError("%s failed (%d): %s" % (what, rc, load().cg_last_error().decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle():
    # the raw getter is ~20x cheaper than torch.cuda.current_stream() (no Stream object, no device lookups); it honours
    # the thread-local current stream, including the one the autograd engine sets for backward nodes
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def stream():
    return c_void_p(_stream_handle())


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL).  Only fp32 CUDA tensors cross the ABI."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("tensor on %s passed to a HIP kernel: the hot path has no CPU fallback" % t.device)
    if t.dtype not in _PTR_DTYPES:
        raise HipError("unsupported dtype %s" % t.dtype)
    return c_void_p(t.data_ptr())


_PTR_DTYPES = frozenset((torch.float32, torch.int32, torch.uint8, torch.float16))


_ws_cache = {}
# The workspace table of the hipGraph capture in progress (capture_workspaces), or None.  Process-wide ON PURPOSE: the backward of a
# captured segment runs on autograd's worker threads, which must see the capturing thread's table (a thread-local made every
# backward workspace() call of a capture raise and the trainer fall back to eager execution); captures on two host threads at once
# are not supported (the trainer is single-threaded, like the reference).
_ws_scope = [None]


class capture_workspaces:
    """`with hip.capture_workspaces(table):` around a hipGraph capture -- every workspace() call inside takes its scratch from
    `table` (a dict the graph's owner keeps for as long as the graph lives) instead of the process-wide per-stream cache.

    Why: a captured kernel bakes the address of its scratch buffer.  The process-wide cache is keyed by the raw stream handle,
    and torch hands out the same 32 handles round-robin -- so the capture stream of a new trainer can find a buffer that an
    EARLIER trainer allocated eagerly on that handle.  Baked into the graph, that buffer is then dropped from the cache the
    next time some launch on the handle needs a larger one, returns to the caching allocator, and the next capture's
    `torch.cuda.empty_cache()` (torch.cuda.graph.__enter__ calls it) unmaps it: the graph's next replay writes to freed memory --
    round 4's `Memory access fault ... Write access to a read-only page` in the driver's run (it needed the earlier tests of the
    same process to seed the cache, which is why the test passed on its own).  Buffers allocated inside the capture come from
    the graph's private pool; buffers a capture outgrows stay in the table (earlier kernels of the graph still use them)."""

    def __init__(self, table):
        self.table = table

    def __enter__(self):
        self.prev, _ws_scope[0] = _ws_scope[0], self.table
        return self.table

    def __exit__(self, *exc):
        _ws_scope[0] = self.prev
        return False


def workspace(nbytes, slot=0):
    """Per-device scratch owned by PyTorch's caching allocator; kernels on one stream are ordered,
    so one buffer per stream (and slot) is shared by every op.  Slot 1 carries the instance-norm partials
    from a convolution epilogue to the norm that follows it.  Inside a hipGraph capture the buffers belong to the graph
    (capture_workspaces)."""
    scope = _ws_scope[0]
    if scope is not None:
        key = (_stream_handle(), slot)
        buf = scope.get(key)
        if buf is None or buf.numel() < nbytes:
            n = (max(int(nbytes), 1 << 20) + (1 << 20) - 1) & ~((1 << 20) - 1)
            if buf is not None:
                scope.setdefault('retired', []).append(buf)      # kernels captured so far keep writing to it
            buf = scope[key] = torch.empty(n, dtype=torch.uint8, device="cuda")
        return buf
    dev = (_stream_handle(), slot)   # one scratch per stream (stream handles are unique across devices)
    if torch.cuda.is_current_stream_capturing():
        # a hit is as wrong as a miss: the process-wide buffer would be baked into the graph and may be replaced (and freed) later
        raise HipError("workspace(): called inside a hipGraph capture outside hip.capture_workspaces(...): the process-wide "
                       "per-stream buffer would be baked into the graph and later freed by the cache")
    buf = _ws_cache.get(dev)
    if buf is None or buf.numel() < nbytes:
        n = max(int(nbytes), 1 << 20)
        n = (n + (1 << 20) - 1) & ~((1 << 20) - 1)
        buf = torch.empty(n, dtype=torch.uint8, device="cuda")
        _ws_cache[dev] = buf
    return buf


_const_stream = {}

# Value-keyed device constants (gather index vectors, LSGAN target vectors: ops._idx_cache, networks._LossVectors) are cached for
# the life of the process ONLY while a captured hipGraph may hold their addresses.  Without a capture (eager training with
# varying colleague picks or a ramped loss weight) a cache clears itself at CONST_CACHE_MAX entries: a leak bound, not an LRU.
# Kernels in flight on other streams may still read an evicted tensor (consumers do not record_stream on it), so the device is
# drained first -- once per CONST_CACHE_MAX misses.
CONST_CACHE_MAX = 4096
_const_pinned = [False]


def pin_const_caches():
    """Called before the first hipGraph capture of the process: from here on the value-keyed constant caches never evict."""
    _const_pinned[0] = True


def const_cache_put(cache, key, value):
    if not _const_pinned[0] and len(cache) >= CONST_CACHE_MAX:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        cache.clear()
    cache[key] = value
    return value


def upload_const(host):
    """Device copy of a small host tensor that is about to enter a BY-VALUE cache read from several HIP streams (gather
    index vectors, LSGAN target / weight vectors).  A `non_blocking` copy on the stream that happens to miss first is not
    enough there: a second stream hitting the cache a moment later has no ordering against that copy and its kernel reads
    whatever the fresh allocation held -- garbage row indices, i.e. an out-of-bounds gather (the intermittent memory fault
    of round 4's driver run).  So the copy goes to a dedicated stream and the HOST waits for it (a few microseconds: that
    stream carries nothing else), after which the tensor is valid for every stream.  Not possible while the current
    stream is being captured into a hipGraph: the warm-up pass has filled the caches by then, and a miss there raises
    (Council_Trainer._run then falls back to eager execution)."""
    if torch.cuda.is_current_stream_capturing():
        raise HipError("upload_const: cache miss inside a hipGraph capture (the eager warm-up pass did not see this value)")
    dev = torch.cuda.current_device()
    st = _const_stream.get(dev)
    if st is None:
        st = _const_stream[dev] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        out = host.pin_memory().to("cuda:%d" % dev, non_blocking=True)
    st.synchronize()
    return out


def tuning():
    """Current kernel-selection table (cg_tuning_get)."""
    t = Tuning()
    check(load().cg_tuning_get(ctypes.byref(t)), "cg_tuning_get")
    return t


class tuned:
    """`with hip.tuned(x3_korder=1, tile_rows_scale=4): ...` -- cg_tuning fields for a scope (tests / A-B tools)."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.prev = tuning()
        t = tuning()
        for k, v in self.fields.items():
            if k not in dict(Tuning._fields_):
                raise AttributeError("cg_tuning has no field %r" % k)
            setattr(t, k, int(v))
        check(load().cg_tuning_set(ctypes.byref(t)), "cg_tuning_set")
        return t

    def __exit__(self, *exc):
        check(load().cg_tuning_set(ctypes.byref(self.prev)), "cg_tuning_set")
        return False


PROF_SLOTS = 200


def prof_enable(on):
    check(load().cg_prof_enable(int(bool(on))), "cg_prof_enable")


def prof_collect():
    """{kernel name: (launches, total ms, total algorithmic FLOPs)} since profiling was enabled."""
    lib = load()
    counts = (c_int64 * PROF_SLOTS)()
    ms = (c_double * PROF_SLOTS)()
    flops = (c_double * PROF_SLOTS)()
    check(lib.cg_prof_collect(counts, ms, flops), "cg_prof_collect")
    out = {}
    for i in range(PROF_SLOTS):
        if counts[i]:
            out[lib.cg_prof_slot_name(i).decode()] = (int(counts[i]), float(ms[i]), float(flops[i]))
    return out


def prof_report():
    """Per-layer-shape text table of the last prof_collect()."""
    return load().cg_prof_report().decode()


COMM_ID_BYTES = 128


def comm_unique_id():
    """The 128-byte id the first rank of a group draws and hands to the others (cg_comm_unique_id)."""
    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    check(load().cg_comm_unique_id(buf), "cg_comm_unique_id")
    return bytes(buf)


class Comm:
    """One RCCL communicator of the C-ABI (include/council_gan_hip.h: cg_comm_*): this rank's membership in one process
    group.  Creation is collective over the group; the calls enqueue on the current stream."""

    def __init__(self, uid, rank, nranks):
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("communicator id must be %d bytes" % COMM_ID_BYTES)
        self._h = c_void_p()
        self.rank, self.nranks = rank, nranks
        buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
        check(load().cg_comm_create(buf, rank, nranks, ctypes.byref(self._h)), "cg_comm_create")

    def all_gather(self, recv, send):
        """recv (flat, nranks x send.numel() fp32) <- every rank's send (contiguous fp32), on the current stream."""
        if recv.numel() != self.nranks * send.numel() or not (send.is_contiguous() and recv.is_contiguous()):
            raise HipError("all_gather: recv must be a contiguous nranks x send buffer")
        if send.dtype != torch.float32 or recv.dtype != torch.float32:
            raise HipError("all_gather: fp32 only")
        check(load().cg_allgather_images(self._h, ptr(send), ptr(recv), send.numel(), stream()), "cg_allgather_images")

    def all_reduce_sum_(self, t):
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise HipError("all_reduce_sum_: contiguous fp32 only")
        check(load().cg_allreduce_sum(self._h, ptr(t), t.numel(), stream()), "cg_allreduce_sum")
        return t

    def close(self):
        if self._h:
            h, self._h = self._h, c_void_p()
            check(load().cg_comm_destroy(h), "cg_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass
