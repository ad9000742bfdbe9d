"""Operators of the Council-GAN hot path: HIP forward + HIP backward behind torch.autograd.

PyTorch's role here is plumbing only: device memory (caching allocator), the stream, and the
autograd tape that sequences the backward kernels.  All arithmetic runs in the gfx950 library
through the C-ABI (`hip.py`).  Activations are logical NCHW tensors with `channels_last`
strides (= physical NHWC); conv weights are logical OIHW with `channels_last` strides
(= physical [O][KH][KW][I]), so state_dict shapes stay those of the reference.

Parameter gradients: when a parameter carries a `_cg_grad` buffer (a view into the owning
optimizer's flat gradient buffer, see optim.py) the backward kernels accumulate straight into it
and autograd receives None for that input; otherwise the gradient is returned normally.
"""
import contextlib
import ctypes
import os
from ctypes import byref, c_void_p

import torch

from . import hip
from .hip import ACT, ConvGeom, check, ptr, stream, workspace

CL = torch.channels_last


def nhwc(t):
    return t if t is None else t.contiguous(memory_format=CL)


def empty_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), dtype=torch.float32, device=like.device, memory_format=CL)


def _lib():
    return hip.load()


def _off(t, elems):
    return c_void_p(t.data_ptr() + 4 * elems)


# ------------------------------------------------------------------------------------------
# member-batched ("grouped") execution
# ------------------------------------------------------------------------------------------
# Inside `with members(n):` every operator treats the batch dimension of its activations as n consecutive blocks of
# samples, one per council member (the members' SAME layer runs as one launch): convolutions / linears read member z's
# parameters z * pool.stride elements after the lead member's (optim.ParamPool; hip.Group = cg_group), the loss
# reductions return one value per member; every other operator is per sample and does not care.
class _Scope:
    n = 1


_G = _Scope()
_group_cache = {}


@contextlib.contextmanager
def members(n):
    prev, _G.n = _G.n, int(n)
    try:
        yield
    finally:
        _G.n = prev


def group_n():
    return _G.n


def _grp(weight):
    """byref(cg_group) for a parameter under the current scope, or None for an ordinary single-member call."""
    n = _G.n
    if n <= 1:
        return None
    pool = getattr(weight, '_cg_pool', None)
    if pool is None:
        raise hip.HipError("member-batched launch on a parameter that does not live in an optim.ParamPool")
    key = (n, pool.stride)
    g = _group_cache.get(key)
    if g is None:
        g = _group_cache[key] = hip.Group(n, 0, pool.stride)
    return byref(g)


# ------------------------------------------------------------------------------------------
# convolution geometry
# ------------------------------------------------------------------------------------------
_geom_cache = {}


def fwd_geom(N, H, W, C1, C2, up, KH, KW, stride, pad, Cout, act):
    """Forward pass of ZeroPad2d(pad) -> Conv2d(KHxKW, stride) (networks.py:515-516) on an input that
    is optionally read through a nearest 2x upsample (networks.py:385).  Geometry structs are immutable once built
    and cached by signature (a training step asks for the same few dozen thousands of times)."""
    key = (N, H, W, C1, C2, up, KH, KW, stride, pad, Cout, act)
    g = _geom_cache.get(key)
    if g is not None:
        return g
    g = _build_geom(*key)
    if len(_geom_cache) > 4096:
        _geom_cache.clear()
    _geom_cache[key] = g
    return g


def _build_geom(N, H, W, C1, C2, up, KH, KW, stride, pad, Cout, act):
    g = ConvGeom()
    Hl, Wl = H << up, W << up
    Ho = (Hl + 2 * pad - KH) // stride + 1
    Wo = (Wl + 2 * pad - KW) // stride + 1
    if KH * KW > hip.MAX_TAPS:
        raise ValueError("kernel %dx%d has more than %d taps" % (KH, KW, hip.MAX_TAPS))
    g.N, g.H, g.W, g.C1, g.C2, g.up = N, H, W, C1, C2, up
    g.Ho, g.Wo, g.HoF, g.WoF = Ho, Wo, Ho, Wo
    g.osy = g.osx = 1
    g.ooy = g.oox = 0
    g.stride, g.T, g.Cout, g.act = stride, KH * KW, Cout, act
    for kh in range(KH):
        for kw in range(KW):
            g.dy[kh * KW + kw] = kh - pad
            g.dx[kh * KW + kw] = kw - pad
    return g


def dgrad_classes(Hl, Wl, KH, KW, stride, pad):
    """Data-gradient of a strided conv = one stride-1 pass per output-parity class (ph, pw):
    dx[s*oy+ph] = sum over taps kh with (ph+pad-kh) % s == 0 of dz[oy + (ph+pad-kh)/s] * W[kh]."""
    out = []
    for ph in range(stride):
        for pw in range(stride):
            taps = []
            for kh in range(KH):
                if (ph + pad - kh) % stride:
                    continue
                for kw in range(KW):
                    if (pw + pad - kw) % stride:
                        continue
                    taps.append((kh * KW + kw, (ph + pad - kh) // stride, (pw + pad - kw) // stride))
            Hc = (Hl - ph + stride - 1) // stride
            Wc = (Wl - pw + stride - 1) // stride
            if Hc > 0 and Wc > 0:
                out.append((ph, pw, Hc, Wc, taps))
    return out


def conv_dgrad(g, dz, w, ci0, nci, grp=None):
    """dx (w.r.t. channels [ci0, ci0+nci) of the conv input) from dz [N,Cout,Ho,Wo].  `g` is the FORWARD geometry of
    the layer; the library derives the output-parity classes of a strided conv (`dgrad_classes` above is the host
    restatement the CPU tests check) and runs them as one launch of the forward implicit-GEMM kernel on dz."""
    lib = _lib()
    N, H, W, up = g.N, g.H, g.W, g.up
    dxl = empty_nhwc(N, nci, H << up, W << up, dz)
    ws = workspace(lib.cg_conv2d_dgrad_workspace_g(byref(g), grp, nci))
    check(lib.cg_conv2d_dgrad_g(byref(g), grp, ptr(dz), ptr(w), ci0, nci, ptr(dxl), ptr(ws), ws.numel(), stream()),
          "cg_conv2d_dgrad")
    if not up:
        return dxl
    dx = empty_nhwc(N, nci, H, W, dz)
    check(lib.cg_upsample2x_bwd(ptr(dxl), ptr(dx), N, H, W, nci, stream()), "cg_upsample2x_bwd")
    return dx


class _Conv2d(torch.autograd.Function):
    """act(conv2d(zero_pad(x (++ x2)), weight) + bias); networks.py:515-521 without the norm.  Under ops.members(n) the
    batch holds n members' samples and `weight` / `bias` / the gradient buffers are the LEAD member's (the others sit
    pool.stride elements further on)."""

    @staticmethod
    def forward(ctx, x, x2, weight, bias, wgrad_buf, bgrad_buf, stride, pad, act, up, stats, xsplit=None, wsplit=None,
                out_split=None, amax_out=None, wmgr=None, wref=None, dz_split_ok=None, x_no_f32=False, bounded=None, x_bwd=None,
                bwd_want=None, want_f32=True, skip=None):
        # wref = (cg_group or None, the Parameter object): resolved by the caller, where the tensor still carries its
        # Python attributes
        lib = _lib()
        x, x2, w = nhwc(x), nhwc(x2), nhwc(weight)
        N, C1, H, W = x.shape
        C2 = x2.shape[1] if x2 is not None else 0
        Cout, Ct, KH, KW = w.shape
        if Ct != C1 + C2:
            raise ValueError("conv weight expects %d input channels, got %d" % (Ct, C1 + C2))
        if x2 is not None and tuple(x2.shape[0:1] + x2.shape[2:]) != (N, H, W):
            raise ValueError("concat sources must agree in N, H, W")
        grp, wparam = wref if wref is not None else (None, None)
        if grp is not None and w.data_ptr() != weight.data_ptr():
            raise hip.HipError("member-batched launch: the weight is not stored channels_last in its pool")
        g = fwd_geom(N, H, W, C1, C2, int(up), KH, KW, stride, pad, Cout, act)
        y = empty_nhwc(N, Cout, g.Ho, g.Wo, x)
        want_stats = stats is not None and act == 0
        rows = ctypes.c_int(0)
        sws, sbytes, rp = None, 0, None
        if want_stats:           # an instance norm follows: the conv epilogue emits its partial sums when it can
            sws = workspace(((N * g.Ho * g.Wo + 63) // 64) * Cout * 16, slot=1)
            sbytes, rp = sws.numel(), byref(rows)
        state, nslots = None, None
        if xsplit is not None and wsplit is not None and x3_eligible(C1, C2):
            # split-precision forward (fp16 x 3 MFMA, 22 significand bits)
            ysp = None
            if bounded is not None:
                # bounded split (include/council_gan_hip.h, cg_x3_epilogue): the epilogue writes the {hi, lo} planes of this
                # un-normalised output on an a-priori scale -- the next convolution reads them, nobody measures + splits y.
                # With a sign-only activation (relu / lrelu) nothing reads y in fp32 any more (the backward takes the sign from
                # the hi plane): the fp32 copy is not written, `y` stays an uninitialised carrier for the autograd graph
                skip_f32 = BOUNDED_NO_F32 and not want_f32 and act in (ACT["relu"], ACT["lrelu"])
                ostate = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=y.device)
                buf = torch.empty(2 * y.numel(), dtype=torch.float16, device=y.device)
                epi = hip.X3Epilogue(bounded.data_ptr(), xsplit.state.data_ptr(), int(xsplit.nslots), 0, None, ostate.data_ptr())
                ns = ctypes.c_int(0)
                check(lib.cg_conv2d_fwd_x3_e(byref(g), grp, xsplit.hi_ptr(), xsplit.lo, wsplit.hi_ptr(), wsplit.lo,
                                             float(wsplit.scale), wsplit.scale_ptr(), xsplit.scale_ptr(), ptr(bias),
                                             None if skip_f32 else ptr(y), ptr(buf), x3_lo(y.numel()), byref(epi), -1, byref(ns),
                                             stream()), "cg_conv2d_fwd_x3_e")
                ysp = SplitTensor(buf, y.shape, state=ostate, nslots=ns.value)
                out_split.append(ysp)
                if skip_f32:
                    out_split.append(True)
                if ns.value == 0:
                    raise hip.HipError("bounded split: the kernel reported no output maxima")
            else:
                if out_split is not None:
                    ysp = SplitTensor(torch.empty(2 * y.numel(), dtype=torch.float16, device=y.device), y.shape)
                    out_split.append(ysp)
                if amax_out is not None and ysp is None:     # the consumer will split y dynamically: hand it the block maxima
                    state, nslots = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=y.device), ctypes.c_int(0)
                check(lib.cg_conv2d_fwd_x3_g(byref(g), grp, xsplit.hi_ptr(), xsplit.lo, wsplit.hi_ptr(), wsplit.lo,
                                             float(wsplit.scale), wsplit.scale_ptr(), xsplit.scale_ptr(), ptr(bias), ptr(y),
                                             ysp.hi_ptr() if ysp else None, ysp.lo if ysp else 0, ptr(sws), sbytes, rp, -1,
                                             ptr(state), byref(nslots) if nslots is not None else None, stream()),
                      "cg_conv2d_fwd_x3")
        else:
            if amax_out is not None and not want_stats:
                state, nslots = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=y.device), ctypes.c_int(0)
            if X3_FORWARD and THIN_X3 and lib.cg_conv2d_fwd_thin_x3_ok(byref(g)):
                # thin-input first layers (3 / 6 / 12 -> 64 channels) of the split-precision datapath: fp32 tensors in and out, the
                # products on the fp16 x 3 MFMA (round 6; no epilogue statistics on this kernel: rows stays 0, the norm measures itself)
                check(lib.cg_conv2d_fwd_thin_x3_g(byref(g), grp, ptr(x), ptr(x2), ptr(w), ptr(bias), ptr(y), ptr(state),
                                                  byref(nslots) if nslots is not None else None, stream()), "cg_conv2d_fwd_thin_x3")
            else:
                check(lib.cg_conv2d_fwd_g(byref(g), grp, ptr(x), ptr(x2), ptr(w), ptr(bias), ptr(y), ptr(sws), sbytes, rp,
                                          ptr(state), byref(nslots) if nslots is not None else None, stream()), "cg_conv2d_fwd")
        if nslots is not None and nslots.value:
            amax_out.append((state, nslots.value))
        if want_stats and rows.value:
            stats.append((sws, rows.value))
        y_no_f32 = bool(out_split is not None and len(out_split) > 1)
        ctx.save_for_backward(x, x2, w, y if (act and not y_no_f32) else None)
        ctx.ysplit = out_split[0] if y_no_f32 else None       # the activation backward takes the sign from its hi plane
        ctx.xsplit = xsplit if (xsplit is not None and wsplit is not None) else None     # reused by the x3 weight gradient
        ctx.x_no_f32 = bool(x_no_f32)
        ctx.x_bwd = x_bwd        # what the producer of x wants from this layer's data gradient (see backward)
        ctx.bwd_cell = None
        ctx.skip = skip          # SkipLink: the gradient of the ResBlock's skip connection joins this layer's data gradient
        ctx.g, ctx.meta = g, (KH, KW, stride, pad, act, int(up), bias is not None)
        ctx.wgrad_buf, ctx.bgrad_buf = wgrad_buf, bgrad_buf
        ctx.grp, ctx.weight, ctx.wmgr, ctx.nm = grp, wparam, wmgr, _G.n      # backward may run outside the members() scope
        if bwd_want is not None and act in (ACT["relu"], ACT["lrelu"]) and X3_BACKWARD and FUSED_ACT_BWD:
            # This layer's activation backward can be folded into the data-gradient epilogue of the convolution that reads y
            # (cg_x3_epilogue.act_src): tell it which forms of dz = dy * act'(y) this layer's own gradients will read
            need_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
            need_dw = ctx.needs_input_grad[2] or (bias is not None and ctx.needs_input_grad[3])
            dg = need_dx and g.Cout % 32 == 0 and g.stride <= 2 and (grp is None or (wmgr is not None and wparam is not None))
            wg = need_dw and ctx.xsplit is not None and bool(lib.cg_conv2d_wgrad_x3_ok_g(byref(g), grp))
            if need_dx or need_dw:
                ctx.bwd_cell = [0]       # counts the consumers that folded this layer's activation backward into their data gradient
                bwd_want.append((act, bool(dg or wg), bool((need_dx and not dg) or (need_dw and not wg)), ctx.bwd_cell))
        if dz_split_ok is not None and act == 0 and X3_BACKWARD:
            # Will BOTH gradients of this layer run on the split-precision kernels?  Then the norm behind it may hand its
            # dx over in split form only (cg_instnorm_bwd_split) -- the same conditions backward() evaluates.
            need_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
            need_dw = ctx.needs_input_grad[2] or (bias is not None and ctx.needs_input_grad[3])
            dg = need_dx and g.Cout % 32 == 0 and g.stride <= 2 and (grp is None or (wmgr is not None and wparam is not None))
            wg = need_dw and ctx.xsplit is not None and bool(lib.cg_conv2d_wgrad_x3_ok_g(byref(g), grp))
            if (need_dx or need_dw) and (dg or not need_dx) and (wg or not need_dw):
                dz_split_ok.append(True)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, x2, w, y = ctx.saved_tensors
        KH, KW, stride, pad, act, up, has_bias = ctx.meta
        g, grp = ctx.g, ctx.grp
        # maxima / split form a producer attached to THIS gradient tensor: valid only while nobody wrote to it since (the
        # autograd engine may accumulate another contribution into the same tensor in place: that bumps its version)
        amax = getattr(dy, "_cg_amax", None)
        if amax is not None and len(amax) == 3:
            amax = amax[:2] if dy._version == amax[2] else None
        pre = getattr(dy, "_cg_dz_split", None)      # the norm behind this layer delivered dz in split form ONLY
        if pre is not None and dy._version != getattr(dy, "_cg_dz_version", dy._version):
            raise hip.HipError("a gradient delivered in split form only was modified in place before its consumer ran")
        dy = nhwc(dy)
        dx = dw = db = None
        # split-precision backward: dz gets a device-side power-of-two scale once, for both gradients
        need_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
        need_dw = ctx.needs_input_grad[2] or (has_bias and ctx.needs_input_grad[3])
        # (a member-batched launch reads the prepared data-gradient weights of a pool: plain tensors / linears stay fp32)
        x3_dgrad = X3_BACKWARD and need_dx and g.Cout % 32 == 0 and g.stride <= 2 and \
            (grp is None or (ctx.wmgr is not None and ctx.weight is not None))
        x3_wgrad = X3_BACKWARD and need_dw and ctx.xsplit is not None and bool(lib.cg_conv2d_wgrad_x3_ok_g(byref(g), grp))
        fp32_needed = (need_dx and not x3_dgrad) or (need_dw and not x3_wgrad)
        if ctx.x_no_f32 and need_dw and not x3_wgrad:
            # this width takes the fp32 weight gradient, but the producer wrote the {hi, lo} planes only: rebuild the values
            # (22 significand bits -- what the split-precision kernel would have multiplied)
            xs = ctx.xsplit
            x = torch.empty_like(x)
            check(lib.cg_unsplit_f16(xs.hi_ptr(), xs.lo, xs.scale_ptr(), ptr(x), x.numel(), stream()), "cg_unsplit_f16")
        dzs = None
        wgrad_act = None
        applied = act and getattr(dy, "_cg_act_applied", None) == dy._version       # the consumer's data gradient already holds
        if applied:                                                                 # dy * act'(y) in the forms this layer asked for
            dz = dy
            dzs = pre
            if (dzs is None and (x3_dgrad or x3_wgrad)) or (fp32_needed and not getattr(dy, "_cg_has_f32", True)):
                raise hip.HipError("a gradient with the activation backward folded in arrived without the form this layer reads")
        elif act:
            if ctx.bwd_cell is not None and ctx.bwd_cell[0]:
                # a consumer multiplied its data gradient by act'(y) already, but what arrived here is not that tensor (several
                # consumers' gradients were accumulated): applying act' again would square it
                raise hip.HipError("fused activation backward: the output of this layer has more than one differentiated consumer "
                                   "(set CG_FUSED_ACT_BWD=0)")
            if y is None:       # y exists as {hi, lo} planes only (bounded split): rebuild it for the un-fused activation backward
                ys = ctx.ysplit
                y = torch.empty_like(dy)
                check(lib.cg_unsplit_f16(ys.hi_ptr(), ys.lo, ys.scale_ptr(), ptr(y), y.numel(), stream()), "cg_unsplit_f16")
            if x3_dgrad or x3_wgrad:
                dz, dzs = act_bwd_split(dy, y, act, fp32_needed, amax)  # no fp32 round trip of dz when nobody reads it
            elif WGRAD_ACT and need_dw and not need_dx and bool(lib.cg_conv2d_wgrad_act_ok(byref(g))):
                # only the weight gradient reads dz, and this layer's kernel applies act'(y) while it loads dy (the
                # discriminators' first layers in their own updates): dz is never written
                dz, wgrad_act = dy, (y, act)
            else:
                dz = torch.empty_like(dy)
                check(lib.cg_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), act, stream()), "cg_act_bwd")
        else:
            dz = dy
            if pre is not None:
                if act or fp32_needed:
                    raise hip.HipError("a gradient delivered in split form only reached a kernel that needs it in fp32")
                dzs = pre
            elif x3_dgrad or x3_wgrad:
                dzs = split_f16_dynamic(dz, amax)
        if pre is not None and act and not applied:
            raise hip.HipError("a gradient delivered in split form only reached a layer with a fused activation")
        # data gradients: split-precision kernel when the layer qualifies (dz gets a device-side power-of-two scale)
        addend = ctx.skip.take() if ctx.skip is not None else None
        if addend is not None and (x2 is not None or not ctx.needs_input_grad[0]):
            raise hip.HipError("a skip-connection gradient was handed to a layer that has no single input gradient to add it to")

        def dgrad(ci0, nci):
            if x3_dgrad:
                fuse = ctx.x_bwd if (ci0 == 0 and nci == x.shape[1] and x2 is None and not up and addend is None) else None
                return conv_dgrad_x3(g, dzs, w, ci0, nci, grp=grp, weight=ctx.weight, wmgr=ctx.wmgr, nm=ctx.nm,
                                     fuse=fuse, xsplit=ctx.xsplit, addend=addend)
            d = conv_dgrad(g, dz, w, ci0, nci, grp=grp)
            return d if addend is None else add(d, addend)

        def run_dgrads():
            d1 = dgrad(0, x.shape[1]) if ctx.needs_input_grad[0] else None
            d2 = dgrad(x.shape[1], x2.shape[1]) if (x2 is not None and ctx.needs_input_grad[1]) else None
            return d1, d2

        dx2 = None
        side = _companion() if (need_dw and ctx.wgrad_buf is not None) else None
        if side is not None and WGRAD_AFTER_DGRAD:
            # the weight gradient starts when this layer's DATA gradient has finished: it then shares the GPU with the
            # HBM-bound passes that follow on the compute stream (norm / activation backward of the layer below), not with
            # the equally MFMA-bound data gradient
            dx, dx2 = run_dgrads()
        if need_dw:
            if ctx.wgrad_buf is not None:
                # accumulate straight into the optimizer's flat gradient buffer (optim.py); grouped: every member's slice
                dw_t, acc = ctx.wgrad_buf, 1
                db_t = ctx.bgrad_buf if has_bias else None
                dw_t._cg_touched = True
                if db_t is not None:
                    db_t._cg_touched = True
            else:
                if grp is not None:
                    raise hip.HipError("member-batched weight gradients need pool-backed gradient buffers")
                dw_t, acc = torch.empty_like(w), 0
                db_t = torch.empty(w.shape[0], dtype=torch.float32, device=w.device) if has_bias else None
                dw, db = dw_t, db_t
            if side is not None:
                ev = torch.cuda.Event()
                ev.record()                           # dz / its split form and the zeroed gradient buffers are ready
                side.wait_event(ev)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                ws = workspace(lib.cg_conv2d_wgrad_workspace_g(byref(g), grp))
                if x3_wgrad:
                    xs = ctx.xsplit
                    check(lib.cg_conv2d_wgrad_x3_g(byref(g), grp, xs.hi_ptr(), xs.lo, xs.scale_ptr(), dzs.hi_ptr(), dzs.lo,
                                                   dzs.scale_ptr(), ptr(dw_t), ptr(db_t), acc, ptr(ws), ws.numel(), stream()),
                          "cg_conv2d_wgrad_x3")
                    used = (xs.buf, xs.state, dzs.buf, dzs.state)
                elif wgrad_act is not None:
                    check(lib.cg_conv2d_wgrad_act_g(byref(g), grp, ptr(x), ptr(x2), ptr(dz), ptr(wgrad_act[0]), wgrad_act[1],
                                                    ptr(dw_t), ptr(db_t), acc, ptr(ws), ws.numel(), stream()), "cg_conv2d_wgrad_act")
                    used = (x, x2, dz, wgrad_act[0])
                else:
                    check(lib.cg_conv2d_wgrad_g(byref(g), grp, ptr(x), ptr(x2), ptr(dz), ptr(dw_t), ptr(db_t), acc, ptr(ws),
                                                ws.numel(), stream()), "cg_conv2d_wgrad")
                    used = (x, x2, dz)
            if side is not None:                      # the allocator must not hand these blocks out before the companion
                for t in used:                        # stream has read them
                    if t is not None:
                        t.record_stream(side)
        if not (side is not None and WGRAD_AFTER_DGRAD):
            dx, dx2 = run_dgrads()
        return (dx, dx2, dw, db) + (None,) * 20


# Weight gradients leave the chain of dependent backward kernels: nothing downstream of a layer's backward needs its dW
# before the optimizer step.  They are therefore launched on a COMPANION stream of the stream the backward runs on (one
# per compute stream), behind an event that marks "dz is ready", and the compute stream goes straight on with the data
# gradient: the MFMA-bound weight-gradient kernels then share the GPU with the HBM-bound passes of the chain (norm /
# activation backward, splits) instead of queueing between them.  wgrad_join() -- called by the trainer before the
# optimizer step -- makes the compute stream wait for its companion.  Measured on the member-batched step: 73.5 ms with it
# against 71.7 ms without (the weight-gradient kernels are MFMA-bound like the data-gradient kernels they then compete with,
# and the batched elementwise passes are too short to hide them) -- so it is OFF unless CG_WGRAD_STREAM=1.
WGRAD_STREAM = os.environ.get("CG_WGRAD_STREAM", "0") == "1"
# CG_WGRAD_ACT=0: the thin-input layers run their activation backward as a pass of its own before the weight gradient
WGRAD_ACT = os.environ.get("CG_WGRAD_ACT", "1") != "0"
# CG_WGRAD_AFTER_DGRAD=1 (with CG_WGRAD_STREAM=1): the companion stream waits for the layer's data gradient, not just for dz
WGRAD_AFTER_DGRAD = os.environ.get("CG_WGRAD_AFTER_DGRAD", "1") == "1"
_companions = {}


def _companion():
    """Companion stream of the current stream (created on first use), or None when the feature is off."""
    if not WGRAD_STREAM:
        return None
    key = hip._stream_handle()
    st = _companions.get(key)
    if st is None:
        st = _companions[key] = torch.cuda.Stream()
    return st


def wgrad_join():
    """The current stream continues after every weight gradient launched from it so far (no host wait)."""
    st = _companions.get(hip._stream_handle())
    if st is not None:
        torch.cuda.current_stream().wait_stream(st)


# ------------------------------------------------------------------------------------------
# nearest-2x upsample + 3x3 convolution as a 4x4 stride-2 transposed convolution (include/council_gan_hip.h, cg_upconv_*)
# ------------------------------------------------------------------------------------------
UPC_WSCALE = 0.25      # the summed-tap weights are up to 4x a weight: a quarter of the pool's scale keeps their hi halves finite
# CG_UPCONV=0: the upsampling layers run as 3x3 convolutions that gather through the upsample (A/B switch)
UPCONV = os.environ.get("CG_UPCONV", "1") != "0"
# CG_THIN_X3=0: the thin-input first layers stay on the exact-fp32 MFMA kernel under the split-precision datapath too (A/B switch)
THIN_X3 = os.environ.get("CG_THIN_X3", "1") != "0"
_upc_groups = {}


def _upc_group(n, elems):
    """cg_group over the packed per-member summed-tap weights / their gradient (None for a single member)."""
    if n <= 1:
        return None
    key = (n, elems)
    g = _upc_groups.get(key)
    if g is None:
        g = _upc_groups[key] = hip.Group(n, 0, elems)
    return byref(g)


def upconv_fwd_x3(xs, weight, bias, wmgr, grp, n, stats=None):
    """y = conv3x3(upsample2x(x)) + bias on the summed-tap weights (no autograd): xs SplitTensor [N, Cin, H, W]."""
    lib = _lib()
    N, C1, H, W = xs.shape
    Cout = weight.shape[0]
    wt = wmgr.upconv_weights(weight, grp, n)
    if wt is None:
        raise hip.HipError("upsample-convolution: the weight is not managed by a split-weight table")
    g = fwd_geom(N, H, W, C1, 0, 1, 3, 3, 1, 1, Cout, 0)
    y = torch.empty((N, Cout, 2 * H, 2 * W), dtype=torch.float32, device=xs.buf.device, memory_format=CL)
    rows = ctypes.c_int(0)
    sws, sbytes, rp = None, 0, None
    if stats is not None:
        sws = workspace(((N * 4 * H * W + 63) // 64) * Cout * 16, slot=1)
        sbytes, rp = sws.numel(), byref(rows)
    check(lib.cg_upconv2d_fwd_x3(byref(g), grp, xs.hi_ptr(), xs.lo, xs.scale_ptr(), ptr(wt[0]), UPC_WSCALE, wmgr.scale_ptr(),
                                 ptr(bias), ptr(y), ptr(sws), sbytes, rp, stream()), "cg_upconv2d_fwd_x3")
    if stats is not None and rows.value:
        stats.append((sws, rows.value))
    return y


class _UpConv2d(torch.autograd.Function):
    """conv2d(zero_pad(upsample2x(x)), weight) + bias (networks.py:385-386 + 513-516) on the summed-tap form: forward = the
    four output-parity classes of a 4x4 stride-2 transposed convolution, data gradient = that 4x4 stride-2 convolution over
    dz, weight gradient = its weight gradient folded back onto the nine taps.  Split-precision operands only."""

    @staticmethod
    def forward(ctx, x, weight, bias, wgrad_buf, bgrad_buf, stats, xsplit, wmgr, wref):
        grp, wparam = wref
        ctx.n = _G.n
        y = upconv_fwd_x3(xsplit, wparam, bias, wmgr, grp, _G.n, stats)
        ctx.xsplit, ctx.wmgr, ctx.grp, ctx.weight = xsplit, wmgr, grp, wparam
        ctx.bufs = (wgrad_buf, bgrad_buf)
        ctx.has_bias = bias is not None
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        N, Cin, H, W = ctx.shape
        weight, wmgr, grp, n = ctx.weight, ctx.wmgr, ctx.grp, ctx.n
        Cout = weight.shape[0]
        pre = getattr(dy, "_cg_dz_split", None)
        if pre is not None and dy._version != getattr(dy, "_cg_dz_version", dy._version):
            raise hip.HipError("a gradient delivered in split form only was modified in place before its consumer ran")
        if pre is not None:
            dzs = pre
        else:
            amax = getattr(dy, "_cg_amax", None)
            if amax is not None and len(amax) == 3:
                amax = amax[:2] if dy._version == amax[2] else None
            dzs = split_f16_dynamic(nhwc(dy), amax)
        wt = wmgr.upconv_weights(weight, grp, n)
        elems = lib.cg_upconv_wt_elems(Cout, Cin)
        gF = fwd_geom(N, 2 * H, 2 * W, Cout, 0, 0, 4, 4, 2, 1, Cin, 0)       # the 4x4 stride-2 convolution over dz
        grpF = _upc_group(n, elems)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, Cin, H, W), dtype=torch.float32, device=dzs.buf.device, memory_format=CL)
            state, nslots = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=dx.device), ctypes.c_int(0)
            check(lib.cg_conv2d_fwd_x3_g(byref(gF), grpF, dzs.hi_ptr(), dzs.lo, ptr(wt[1]), x3_lo(1), UPC_WSCALE, wmgr.scale_ptr(),
                                         dzs.scale_ptr(), None, ptr(dx), None, 0, None, 0, None, -1, ptr(state), byref(nslots),
                                         stream()), "cg_conv2d_fwd_x3 (upsample-convolution data gradient)")
            if nslots.value:
                dx._cg_amax = (state, nslots.value, dx._version)
        if ctx.needs_input_grad[1]:
            wbuf, bbuf = ctx.bufs
            if wbuf is None:
                raise hip.HipError("upsample-convolution weight gradients need pool-backed gradient buffers")
            xs = ctx.xsplit
            if not lib.cg_conv2d_wgrad_x3_ok_g(byref(gF), grpF):
                raise hip.HipError("upsample-convolution: the 4x4 stride-2 weight gradient does not take this shape")
            dwf = torch.empty(n * elems, dtype=torch.float32, device=dzs.buf.device)
            ws = workspace(lib.cg_conv2d_wgrad_workspace_g(byref(gF), grpF))
            check(lib.cg_conv2d_wgrad_x3_g(byref(gF), grpF, dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(), xs.hi_ptr(), xs.lo,
                                           xs.scale_ptr(), ptr(dwf), None, 0, ptr(ws), ws.numel(), stream()),
                  "cg_conv2d_wgrad_x3 (upsample-convolution)")
            check(lib.cg_upconv_fold_dw(grp, ptr(dwf), ptr(wbuf), Cout, Cin, 1, stream()), "cg_upconv_fold_dw")
            wbuf._cg_touched = True
            if ctx.has_bias and bbuf is not None:
                ws2 = workspace(lib.cg_colsum_split_workspace(Cout, n), slot=2)
                check(lib.cg_colsum_split(grp, dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(), N * 4 * H * W, Cout, ptr(bbuf), 1, ptr(ws2),
                                          ws2.numel(), stream()), "cg_colsum_split")
                bbuf._cg_touched = True
        return (dx, dw, db) + (None,) * 6


_upc_ok_cache = {}


def upconv_weight_ok(weight, wmgr):
    """The shape-independent half of _upconv_ok: could a layer with this weight take the summed-tap path at all?  Also the
    predicate of Decoder.prepare_split_weights, so that preparation and use cannot diverge."""
    return bool(UPCONV and X3_FORWARD and X3_BACKWARD and wmgr is not None and weight.dim() == 4
                and tuple(weight.shape[2:]) == (3, 3) and weight.shape[0] % 32 == 0 and weight.shape[1] % 32 == 0
                and 256 % weight.shape[0] == 0 and getattr(weight, "_cg_grad", None) is not None and x3_interleaved())


def _upconv_ok(shape, weight, stride, pad, act, x2, wmgr):
    """Does this upsample + convolution layer take the summed-tap path?  `shape` = (N, Cin, H, W) of the source."""
    if not (upconv_weight_ok(weight, wmgr) and x2 is None and stride == 1 and pad == 1 and ACT[act] == 0):
        return False
    N, Cin, H, W = shape
    Cout, n = weight.shape[0], _G.n
    key = (N, Cin, H, W, Cout, n)
    ok = _upc_ok_cache.get(key)
    if ok is None:       # the weight gradient of the 4x4 stride-2 convolution over dz must run on the split-precision kernels
        gF = fwd_geom(N, 2 * H, 2 * W, Cout, 0, 0, 4, 4, 2, 1, Cin, 0)
        ok = _upc_ok_cache[key] = bool(_lib().cg_conv2d_wgrad_x3_ok_g(byref(gF), _upc_group(n, 16 * Cout * Cin)))
    return ok


# Bounded split / fused activation backward (include/council_gan_hip.h, cg_x3_epilogue) -- built, parity-tested
# (tests/test_gpu_ops.py::test_bounded_split_chain_and_fused_activation_backward) and OFF by default: on the benchmark they remove
# 5.7 ms of bandwidth passes per step (28 -> 10-12 bytes per element at every discriminator layer boundary), but the
# convolution epilogues that take the work over are slower by as much, and the passes had been hiding under the sibling
# update's convolutions (two side streams): 60.7 vs 60.5 ms per step on one GPU, 21.8 vs 21.4 ms on a one-member rank
# (profiles/r04_bounded_split.txt).
#   CG_BOUNDED_SPLIT=1: un-normalised conv outputs leave their kernel as {hi, lo} planes on an a-priori scale
#   CG_BOUNDED_NO_F32=0: ... and are written in fp32 as well
#   CG_FUSED_ACT_BWD=1: a layer's activation backward is folded into the data-gradient epilogue of its consumer
BOUNDED_NO_F32 = os.environ.get("CG_BOUNDED_NO_F32", "1") != "0"
FUSED_ACT_BWD = os.environ.get("CG_FUSED_ACT_BWD", "0") != "0"
BOUNDED_SPLIT = os.environ.get("CG_BOUNDED_SPLIT", "0") != "0"
X3_FORWARD = True     # module switches (Council_Trainer sets them from the config): split-precision forward convolutions,
X3_BACKWARD = True    # split-precision data gradients, dynamic (device-scaled) splitting of un-normalised conv inputs
X3_DYNAMIC_INPUT = True


def conv2d(x, weight, bias=None, stride=1, pad=0, act="none", x2=None, upsample=False, stats=None, wmgr=None,
           want_split=False, want_f32=True, skip=None):
    """Functional conv.  `weight` may be an nn.Parameter managed by a flat optimizer buffer.  `stats`: an empty
    list when an instance norm consumes the output next -- the conv appends (partials, rows) if its epilogue
    produced the norm's partial sums (pass the same list to instance_norm / adain).
    `wmgr` (SplitWeights): run the FORWARD on the split-precision kernel when the layer qualifies; the input's
    split form travels as the `_cg_split` attribute of `x` (set by the op that produced it), and with `want_split`
    the output gets one for the next convolution.  `want_f32=False`: the caller knows that ONLY a convolution reads the
    output -- an un-normalised output with a sign-only activation may then exist as {hi, lo} planes alone (bounded split)."""
    xsplit = wsplit = out_split = None
    no_f32 = bool(getattr(x, "_cg_no_f32", False))
    if upsample and _upconv_ok(tuple(x.shape), weight, stride, pad, act, x2, wmgr) and wmgr.get(weight) is not None:
        xsplit = getattr(x, "_cg_split", None)
        if xsplit is None:
            with torch.no_grad():
                xsplit = split_f16_dynamic(x.detach(), getattr(x, "_cg_amax", None))
        y = _UpConv2d.apply(x, weight, bias, getattr(weight, "_cg_grad", None),
                            getattr(bias, "_cg_grad", None) if bias is not None else None, stats, xsplit, wmgr,
                            (_grp(weight), weight))
        if stats is not None:
            y._cg_dz_split_ok = True      # both gradients of this layer take dz in split form
        return y
    if X3_FORWARD and wmgr is not None and x2 is None and x3_eligible(x.shape[1], 0) and weight.dim() == 4:
        # inputs whose producer emitted the split form are instance-normalised (or one fused conv+ReLU away from it),
        # i.e. O(1) activations inside fp16's accurate range, and travel unscaled; any other input is split here
        # with a per-tensor power-of-two scale chosen on the device
        xsplit = getattr(x, "_cg_split", None)
        wsplit = wmgr.get(weight) if (xsplit is not None or X3_DYNAMIC_INPUT) else None
        if wsplit is None:
            xsplit = None
        else:
            if xsplit is None:       # arbitrary-scale input (discriminator activations): per-tensor scale on the device
                with torch.no_grad():
                    # the convolution that produced x left its per-block maxima behind (amax_out below): no reduction pass
                    xsplit = split_f16_dynamic(x.detach(), getattr(x, "_cg_amax", None))
            if want_split and stats is None and xsplit.state is None:
                out_split = []       # O(1) chain (norm -> conv+ReLU -> conv): the epilogue emits the next operand
    # un-normalised outputs with a multiple of 32 channels may be split dynamically by the next convolution: let this
    # one's epilogue measure them (stats is None: no instance norm in between)
    amax_out = [] if (X3_FORWARD and X3_DYNAMIC_INPUT and stats is None and out_split is None and weight.dim() == 4 and
                      weight.shape[0] % 32 == 0) else None
    if no_f32 and (xsplit is None or wsplit is None):
        raise hip.HipError("an activation produced in split form only reached a convolution that reads fp32")
    # bounded split: an un-normalised output (no norm follows, the input itself carries a device-side scale) that the next
    # convolution will read leaves the kernel as {hi, lo} planes on an a-priori scale
    bounded = None
    if (BOUNDED_SPLIT and want_split and stats is None and out_split is None and xsplit is not None and wsplit is not None
            and xsplit.state is not None and weight.shape[0] % 32 == 0 and x3_interleaved()):
        bounded = wmgr.l1_bound(weight, bias, _grp(weight), _G.n, False)
        if bounded is not None:
            out_split, amax_out = [], None
    dz_ok = [] if (stats is not None and X3_BACKWARD and xsplit is not None) else None      # a norm follows
    bwd_want = [] if (stats is None and torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) else None
    y = _Conv2d.apply(x, x2, weight, bias, getattr(weight, "_cg_grad", None),
                      getattr(bias, "_cg_grad", None) if bias is not None else None,
                      int(stride), int(pad), ACT[act], bool(upsample), stats, xsplit, wsplit, out_split, amax_out, wmgr,
                      (_grp(weight), weight), dz_ok, no_f32, bounded, getattr(x, "_cg_bwd", None), bwd_want, bool(want_f32), skip)
    if dz_ok:
        y._cg_dz_split_ok = True
    if bwd_want:
        y._cg_bwd = bwd_want[0]      # (activation, dz wanted as planes, dz wanted in fp32) -- read by the consumer's backward
    if out_split:
        y._cg_split = out_split[0]
        if len(out_split) > 1:
            y._cg_no_f32 = True
    if amax_out:
        y._cg_amax = amax_out[0]
    return y


def linear(x, weight, bias=None, act="none"):
    """nn.Linear (networks.py:531) as a 1x1 conv on an [N, C, 1, 1] tensor."""
    n = x.shape[0]
    w4 = weight.view(weight.shape[0], weight.shape[1], 1, 1)
    y = _Conv2d.apply(x.reshape(n, -1, 1, 1), None, w4, bias, getattr(weight, "_cg_grad", None),
                      getattr(bias, "_cg_grad", None) if bias is not None else None, 1, 0, ACT[act], False, None, None, None,
                      None, None, None, (_grp(weight), weight))
    return y.reshape(n, -1)


# ------------------------------------------------------------------------------------------
# two 1x1 convolutions with nothing between them, composed (the council discriminator's tail, networks.py:142-143)
# ------------------------------------------------------------------------------------------
# CG_COMPOSE_1X1=0: the two layers run one after the other (A/B switch)
COMPOSE_1X1 = os.environ.get("CG_COMPOSE_1X1", "1") != "0"
_tail_groups = {}


def _tail_group(n, stride):
    if n <= 1:
        return None
    g = _tail_groups.get((n, stride))
    if g is None:
        g = _tail_groups[(n, stride)] = hip.Group(n, 0, stride)
    return byref(g)


class _ComposedTail(torch.autograd.Function):
    """conv1x1(conv1x1(y, W1, b1), W2, b2) with W2: C -> 1 and no activation in between, evaluated as ONE C -> 1 convolution
    with w_eff = W2 W1, b_eff = W2 b1 + b2 (cg_compose1x1_fwd): the C -> C convolution -- forward, data gradient, weight
    gradient -- is never run; its gradients follow from the composed layer's (cg_compose1x1_bwd).  fp32 throughout."""

    @staticmethod
    def forward(ctx, y, W1, b1, W2, b2):
        lib = _lib()
        y = nhwc(y)
        N, C, H, W = y.shape
        n = _G.n
        grp = _grp(W1)
        S = C + 32
        buf = torch.empty(n * S, dtype=torch.float32, device=y.device)
        check(lib.cg_compose1x1_fwd(grp, ptr(W1), ptr(b1), ptr(W2), ptr(b2), C, ptr(buf), S, stream()), "cg_compose1x1_fwd")
        g = fwd_geom(N, H, W, C, 0, 0, 1, 1, 1, 0, 1, 0)
        gt = _tail_group(n, S)
        out = empty_nhwc(N, 1, H, W, y)
        check(lib.cg_conv2d_fwd_g(byref(g), gt, ptr(y), None, ptr(buf), _off(buf, C), ptr(out), None, 0, None, None, None, stream()),
              "cg_conv2d_fwd (composed 1x1 tail)")
        ctx.save_for_backward(y, W1, b1, W2, buf)
        ctx.meta = (g, grp, gt, n, S, C)
        ctx.bufs = tuple(getattr(p, "_cg_grad", None) for p in (W1, b1, W2, b2))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib()
        y, W1, b1, W2, buf = ctx.saved_tensors
        g, grp, gt, n, S, C = ctx.meta
        dout = nhwc(dout)
        dy = None
        if ctx.needs_input_grad[0]:
            dy = torch.empty_like(y)
            ws = workspace(lib.cg_conv2d_dgrad_workspace_g(byref(g), gt, C))
            check(lib.cg_conv2d_dgrad_g(byref(g), gt, ptr(dout), ptr(buf), 0, C, ptr(dy), ptr(ws), ws.numel(), stream()),
                  "cg_conv2d_dgrad (composed 1x1 tail)")
        if any(ctx.needs_input_grad[1:]):
            if any(b is None for b in ctx.bufs):
                raise hip.HipError("composed 1x1 tail: the parameters need pool-backed gradient buffers")
            d = torch.empty(n * S, dtype=torch.float32, device=y.device)
            ws = workspace(lib.cg_conv2d_wgrad_workspace_g(byref(g), gt))
            check(lib.cg_conv2d_wgrad_g(byref(g), gt, ptr(y), None, ptr(dout), ptr(d), _off(d, C), 0, ptr(ws), ws.numel(), stream()),
                  "cg_conv2d_wgrad (composed 1x1 tail)")
            gW1, gb1, gW2, gb2 = ctx.bufs
            check(lib.cg_compose1x1_bwd(grp, ptr(d), S, ptr(W1), ptr(b1), ptr(W2), C, ptr(gW1), ptr(gb1), ptr(gW2), ptr(gb2), stream()),
                  "cg_compose1x1_bwd")
            for t in ctx.bufs:
                t._cg_touched = True
        return dy, None, None, None, None


def composed_tail_ok(c1, c2):
    """Can the two trailing nn.Conv2d holders run as one composed layer?  1x1, stride 1, C -> C -> 1, pool-managed parameters."""
    return (COMPOSE_1X1 and c1.kernel_size == (1, 1) and c2.kernel_size == (1, 1) and c1.stride == (1, 1) and c2.stride == (1, 1)
            and c1.out_channels == c1.in_channels == c2.in_channels and c2.out_channels == 1 and c1.bias is not None
            and c2.bias is not None and all(getattr(p, "_cg_grad", None) is not None for p in (c1.weight, c1.bias, c2.weight, c2.bias))
            and getattr(c1.weight, "_cg_pool", None) is not None)


def composed_tail(y, c1, c2):
    return _ComposedTail.apply(y, c1.weight, c1.bias, c2.weight, c2.bias)


# ------------------------------------------------------------------------------------------
# instance norm / AdaIN (+ activation + residual)
# ------------------------------------------------------------------------------------------
class _InstNormAct(torch.autograd.Function):
    """y = act(IN(x) * gamma + beta) + residual.  gamma/beta are columns [goff, goff+C) / [boff, boff+C)
    of `params` ([N, P], the MLP output, networks.py:303-312) or absent (plain nn.InstanceNorm2d)."""

    @staticmethod
    def forward(ctx, x, params, goff, boff, residual, act, eps, stats, out_split=None, dx_split_only=False, no_f32=False,
                skip=None, pgrad=None):
        lib = _lib()
        ctx.dx_split_only = bool(dx_split_only)
        ctx.pgrad = pgrad if params is not None else None      # ParamGrad: the shared gradient buffer of the AdaIN parameter matrix
        x, residual = nhwc(x), nhwc(residual)
        N, C, H, W = x.shape
        HW = H * W
        mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if stats:
            part, rows = stats[0]
            check(lib.cg_instnorm_stats_from_partials(ptr(part), N, HW, C, rows, eps, ptr(mean), ptr(rstd), stream()),
                  "cg_instnorm_stats_from_partials")
        else:
            ws = workspace(lib.cg_instnorm_workspace(N, HW, C))
            check(lib.cg_instnorm_stats(ptr(x), N, HW, C, eps, ptr(mean), ptr(rstd), ptr(ws), ws.numel(), stream()),
                  "cg_instnorm_stats")
        y = torch.empty_like(x)
        if params is not None:
            params = params.contiguous()
            if params.shape[0] != N or goff + C > params.shape[1] or boff + C > params.shape[1]:
                raise ValueError("AdaIN parameter slice out of range")
            gp, bp, gs = _off(params, goff), _off(params, boff), params.shape[1]
        else:
            gp, bp, gs = None, None, C
        if out_split is not None and (C & 3) == 0 and 256 % (C >> 2) == 0:
            ysp = SplitTensor(torch.empty(2 * y.numel(), dtype=torch.float16, device=y.device), y.shape)
            # no_f32: every consumer reads the {hi, lo} planes (a split-precision convolution, forward and weight gradient) --
            # the fp32 copy is not written; `y` stays an uninitialised carrier for the autograd graph (marked by the caller)
            check(lib.cg_instnorm_apply_split(ptr(x), ptr(mean), ptr(rstd), gp, bp, gs, ptr(residual), None if no_f32 else ptr(y),
                                              ysp.hi_ptr(), ysp.lo, N, HW, C, act, stream()), "cg_instnorm_apply_split")
            out_split.append(ysp)
            if no_f32:
                out_split.append(True)
        else:
            check(lib.cg_instnorm_apply(ptr(x), ptr(mean), ptr(rstd), gp, bp, gs, ptr(residual), ptr(y), N, HW, C, act,
                                        stream()), "cg_instnorm_apply")
        ctx.save_for_backward(x, mean, rstd, params)
        ctx.meta = (goff, boff, act, residual is not None)
        ctx.skip = skip if residual is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, mean, rstd, params = ctx.saved_tensors
        goff, boff, act, has_res = ctx.meta
        dy = nhwc(dy)
        N, C, H, W = x.shape
        HW = H * W
        dx = torch.empty_like(x)
        ws = workspace(lib.cg_instnorm_workspace(N, HW, C))
        dparams = None
        if params is not None:
            gp, bp, gs = _off(params, goff), _off(params, boff), params.shape[1]
            if ctx.needs_input_grad[1] and ctx.pgrad is not None:
                # this layer's 2C columns go straight into the buffer all AdaIN layers of the decoder share; the fork node
                # (adain_param_fork) reports it as the matrix's gradient once, after the last of them
                shared = ctx.pgrad.buffer(params)
                dgp, dbp = _off(shared, goff), _off(shared, boff)
            elif ctx.needs_input_grad[1]:
                dparams = torch.empty_like(params)
                check(lib.cg_fill(ptr(dparams), dparams.numel(), 0.0, stream()), "cg_fill")
                dgp, dbp = _off(dparams, goff), _off(dparams, boff)
            else:
                dgp = dbp = None
        else:
            gp = bp = dgp = dbp = None
            gs = C
        state, nslots = None, ctypes.c_int(0)
        if X3_BACKWARD:
            state = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=x.device)
        wsb = lib.cg_instnorm_bwd_split_workspace(N, HW, C) if (ctx.dx_split_only and X3_BACKWARD and DX_SPLIT) else 0
        if wsb:
            # the convolution in front of this norm takes dz in split form for both of its gradients: write the {hi, lo}
            # planes directly (no fp32 dx, no split pass); `dx` stays an uninitialised carrier of the right shape
            ws = workspace(wsb)
            buf = torch.empty(2 * x.numel(), dtype=torch.float16, device=x.device)
            check(lib.cg_instnorm_bwd_split(ptr(dy), ptr(x), ptr(mean), ptr(rstd), gp, bp, gs, ptr(buf), x3_lo(x.numel()),
                                            ptr(state), None, dgp, dbp, N, HW, C, act, ptr(ws), ws.numel(), stream()),
                  "cg_instnorm_bwd_split")
            dx._cg_dz_split = SplitTensor(buf, x.shape, state=state)
            dx._cg_dz_version = dx._version
        else:
            check(lib.cg_instnorm_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), gp, bp, gs, ptr(dx), dgp, dbp, N, HW, C, act,
                                      ptr(ws), ws.numel(), ptr(state), byref(nslots), stream()), "cg_instnorm_bwd")
            if nslots.value:
                dx._cg_amax = (state, nslots.value, dx._version)     # the conv before this norm splits dx without measuring it again
        dres = dy if has_res else None
        if dres is not None and ctx.skip is not None:
            # the first convolution of the block adds it to its data gradient in that kernel's epilogue (SkipLink): autograd sees
            # no gradient on the skip edge and the sum arrives on the convolution's edge
            ctx.skip.give(dres)
            dres = None
        return dx, dparams, None, None, dres, None, None, None, None, None, None, None, None


# CG_DX_SPLIT=0: instance-norm backward always writes fp32 dx and the convolution splits it in a pass of its own (A/B switch)
DX_SPLIT = os.environ.get("CG_DX_SPLIT", "1") != "0"


# CG_NORM_NO_F32=0: the norm apply always writes the fp32 activation next to its {hi, lo} planes (A/B switch)
NORM_NO_F32 = os.environ.get("CG_NORM_NO_F32", "1") != "0"


def _norm_apply(x, params, goff, boff, residual, act, eps, stats, want_split, want_f32=True, skip=None, pgrad=None):
    out_split = [] if (want_split and X3_FORWARD) else None
    y = _InstNormAct.apply(x, params, goff, boff, residual, ACT[act], float(eps), stats, out_split,
                           bool(getattr(x, "_cg_dz_split_ok", False)),
                           bool(out_split is not None and not want_f32 and NORM_NO_F32 and X3_BACKWARD), skip, pgrad)
    if out_split:
        y._cg_split = out_split[0]
        if len(out_split) > 1:
            y._cg_no_f32 = True      # fp32 values never written: only split-precision kernels may consume this tensor
    return y


def instance_norm(x, act="none", residual=None, eps=1e-5, stats=None, want_split=False, want_f32=True, skip=None):
    return _norm_apply(x, None, 0, 0, residual, act, eps, stats, want_split, want_f32, skip)


def adain(x, params, goff, boff, act="none", residual=None, eps=1e-5, stats=None, want_split=False, want_f32=True, skip=None,
          pgrad=None):
    return _norm_apply(x, params, int(goff), int(boff), residual, act, eps, stats, want_split, want_f32, skip, pgrad)


# CG_ADAIN_FORK=0: every AdaIN layer returns a full-size gradient of the parameter matrix and the autograd engine adds them
ADAIN_FORK = os.environ.get("CG_ADAIN_FORK", "1") != "0"


class ParamGrad:
    """Gradient of the AdaIN parameter matrix ([N, P], the MLP output; networks.py:303-312 hands column slices of it to the
    decoder's AdaIN layers) while a backward pass is collecting it.  The layers own DISJOINT columns, so instead of one zero-filled
    [N, P] gradient per layer plus the engine's additions, every layer's backward writes its columns into one shared buffer
    (`buffer`), and the fork node between the MLP and the layers (`adain_param_fork`) -- which the engine runs after the last of
    them -- reports it (`take`)."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def buffer(self, like):
        if self.buf is None:
            self.buf = torch.empty_like(like)
            check(_lib().cg_fill(ptr(self.buf), self.buf.numel(), 0.0, stream()), "cg_fill")
        return self.buf

    def take(self):
        b, self.buf = self.buf, None
        return b


class _ParamFork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, pgrad):
        ctx.pgrad = pgrad
        ctx.set_materialize_grads(False)
        return params.view_as(params)

    @staticmethod
    def backward(ctx, g):
        b = ctx.pgrad.take()
        if g is not None:        # somebody differentiated through the matrix outside the AdaIN layers (m.weight / m.bias views)
            g = g.contiguous()
            b = g if b is None else add(b, g)
        return b, None


def adain_param_fork(params):
    """(params', ParamGrad) for a parameter matrix that needs its gradient, (params, None) otherwise."""
    if not (ADAIN_FORK and torch.is_grad_enabled() and params.requires_grad):
        return params, None
    pg = ParamGrad()
    return _ParamFork.apply(params.contiguous(), pg), pg


# CG_SKIP_FUSE=0: the gradient of a ResBlock's skip connection is accumulated by the autograd engine (an ATen add per block)
SKIP_FUSE = os.environ.get("CG_SKIP_FUSE", "1") != "0"


class SkipLink:
    """One ResBlock's skip edge in the backward pass (networks.py:448-461, `out += residual`).  The block's input x has two
    consumers -- the first convolution and the residual add fused into the second norm -- so the autograd engine would add their
    two gradients in a pass of its own.  Instead the norm's backward hands the skip gradient over here (`give`) and reports
    None on the skip edge; the first convolution's backward, which by construction runs later, takes it (`take`) and adds it to
    its data gradient in the kernel's epilogue (cg_x3_epilogue.addend) -- or with cg_add where that kernel is not the
    split-precision one.  Same values, same rounding as the engine's addition."""
    __slots__ = ("grad",)

    def __init__(self):
        self.grad = None

    def give(self, g):
        if self.grad is not None:
            raise hip.HipError("skip link: a gradient was handed over twice before the convolution took it")
        self.grad = g

    def take(self):
        g, self.grad = self.grad, None
        return g


def skip_link(x):
    """A SkipLink for a block whose input is x, or None when no gradient flows to x (or the fusion is switched off)."""
    return SkipLink() if (SKIP_FUSE and torch.is_grad_enabled() and x.requires_grad) else None


def add(a, b):
    """a + b (fp32, same shape and layout) on the library's own kernel; no autograd."""
    if a.shape != b.shape or a.stride() != b.stride() or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise hip.HipError("add: operands differ in shape / layout / dtype")
    out = torch.empty_like(a)
    check(_lib().cg_add(ptr(a), ptr(b), ptr(out), a.numel(), stream()), "cg_add")
    return out


class _Activation(torch.autograd.Function):
    """Standalone activation (only the LayerNorm path needs it; convs and IN/AdaIN fuse theirs)."""

    @staticmethod
    def forward(ctx, x, act):
        lib = _lib()
        x = x.contiguous(memory_format=CL) if x.dim() == 4 else x.contiguous()
        y = torch.empty_like(x)
        check(lib.cg_act_fwd(ptr(x), ptr(y), x.numel(), act, stream()), "cg_act_fwd")
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        (y,) = ctx.saved_tensors
        dy = dy.contiguous(memory_format=CL) if dy.dim() == 4 else dy.contiguous()
        dz = torch.empty_like(y)
        check(lib.cg_act_bwd(ptr(dy), ptr(y), ptr(dz), y.numel(), ctx.act, stream()), "cg_act_bwd")
        return dz, None


def activation(x, act):
    return x if ACT[act] == 0 else _Activation.apply(x, ACT[act])


class _LayerNorm(torch.autograd.Function):
    """networks.py:670-686.  Under ops.members(n) the batch holds the n members' samples member-major and every member has
    its OWN gamma / beta (`stride` fp32 elements apart in the optimizer pool, like every other parameter): one launch per
    member on its rows -- LayerNorm is in no shipped configuration, so it stays a plain per-member loop."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _lib()
        x = nhwc(x)
        N, C, H, W = x.shape
        n = max(1, _G.n)
        stride = 0
        if n > 1:
            pool = getattr(gamma, '_cg_pool', None)
            if pool is None or getattr(beta, '_cg_pool', None) is not pool or N % n:
                raise hip.HipError("member-batched LayerNorm: gamma / beta must live in one optim.ParamPool and the batch "
                                   "must hold the members' samples member-major")
            stride = pool.stride
        ctx.bufs = (getattr(gamma, "_cg_grad", None), getattr(beta, "_cg_grad", None))
        y = torch.empty_like(x)
        mean = torch.empty(N, dtype=torch.float32, device=x.device)
        std = torch.empty_like(mean)
        b = N // n
        ws = workspace(lib.cg_layernorm_workspace(b, H * W, C))
        row = C * H * W
        for m in range(n):
            check(lib.cg_layernorm_fwd(_off(x, m * b * row), _off(gamma, m * stride), _off(beta, m * stride), _off(y, m * b * row),
                                       _off(mean, m * b), _off(std, m * b), b, H * W, C, eps, ptr(ws), ws.numel(), stream()),
                  "cg_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, std)
        ctx.eps, ctx.n, ctx.stride = eps, n, stride
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, gamma, mean, std = ctx.saved_tensors
        dy = nhwc(dy)
        N, C, H, W = x.shape
        n, stride = ctx.n, ctx.stride
        b, row = N // n, C * H * W
        dx = torch.empty_like(x)
        dg = torch.empty(n * C, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        ws = workspace(lib.cg_layernorm_workspace(b, H * W, C))
        gbuf, bbuf = ctx.bufs
        if n > 1 and (gbuf is None or bbuf is None):
            raise hip.HipError("member-batched LayerNorm backward needs the pool's gradient buffers")
        for m in range(n):
            check(lib.cg_layernorm_bwd(_off(dy, m * b * row), _off(x, m * b * row), _off(gamma, m * stride), _off(mean, m * b),
                                       _off(std, m * b), _off(dx, m * b * row), _off(dg, m * C), _off(db, m * C), b, H * W, C,
                                       ctx.eps, ptr(ws), ws.numel(), stream()), "cg_layernorm_bwd")
            if gbuf is not None and bbuf is not None:
                check(lib.cg_axpby(1.0, _off(dg, m * C), 1.0, _off(gbuf, m * stride), C, stream()), "cg_axpby")
                check(lib.cg_axpby(1.0, _off(db, m * C), 1.0, _off(bbuf, m * stride), C, stream()), "cg_axpby")
        if gbuf is not None and bbuf is not None:
            gbuf._cg_touched = bbuf._cg_touched = True
            return dx, None, None, None
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, float(eps))


# ------------------------------------------------------------------------------------------
# split-precision ("fp16 x 3") forward path -- tape-free passes only
# ------------------------------------------------------------------------------------------
_X3_IL = None


def x3_interleaved():
    """Layout of the library's {hi, lo} tensors (include/council_gan_hip.h, CG_X3_LO_ELEMS): interleaved per 32 elements
    (the default build) or two separate planes (A/B builds)."""
    global _X3_IL
    if _X3_IL is None:
        _X3_IL = bool(_lib().cg_x3_interleaved())
    return _X3_IL


def x3_lo(numel):
    """The `*_lo_elems` argument for a split tensor of `numel` elements."""
    return hip.X3_LO_ELEMS if x3_interleaved() else numel


class SplitTensor:
    """A tensor stored as fp16 halves hi = f16(s*v), lo = f16(s*v - hi), physical NHWC / [O][KH][KW][I].
    `buf` is a flat fp16 tensor of 2 x numel halves; the tensor starts at ELEMENT `off` of it (a multiple of 32) and its
    lo halves sit `lo` halves after the hi halves (x3_lo); `scale` is the power of two s the values were multiplied by
    (weights: hip.X3_WSCALE)."""

    def __init__(self, buf, shape, off=0, lo=None, scale=1.0, state=None, nslots=0):
        self.buf, self.shape, self.off, self.scale = buf, tuple(shape), off, scale
        self.state = state      # device floats of a dynamically scaled tensor: [0] = max |x| (or its bound), [1] = the scale in use
        self.nslots = nslots    # > 0: state[2 .. 2 + nslots) holds per-block maxima of x nobody has reduced into state[0]
        n = 1
        for d in shape:
            n *= d
        self.numel = n
        self.lo = x3_lo(n) if lo is None else lo

    def hi_ptr(self):
        if x3_interleaved():
            if self.off % 32:
                raise hip.HipError("a split tensor must start on a 32-element boundary of its buffer")
            return c_void_p(self.buf.data_ptr() + 4 * self.off)
        return c_void_p(self.buf.data_ptr() + 2 * self.off)

    def scale_ptr(self):
        return None if self.state is None else c_void_p(self.state.data_ptr() + 4)

    def to_float(self):
        """hi + lo as a flat fp32 tensor in the physical element order, still multiplied by the scale (debug / tests)."""
        n = self.numel
        if x3_interleaved():
            g = self.buf[2 * self.off:2 * (self.off + n)].view(-1, 64).float()
            return (g[:, :32] + g[:, 32:]).reshape(-1)
        return self.buf[self.off:self.off + n].float() + self.buf[self.off + self.lo:self.off + self.lo + n].float()


class SplitWeights:
    """{hi, lo} fp16 planes of ALL parameters of one optimizer (optim.FlatAdam) or of one pool of member optimizers
    (optim.ParamPool), re-split lazily -- one kernel over the flat storage -- whenever the owner's `version` moved (a step,
    a checkpoint load).  The power-of-two scale is chosen ON THE DEVICE from the storage's largest magnitude, capped at
    hip.X3_WSCALE = 2^10: ordinary weights (|w| < 8) get exactly the static scale of round 1, larger ones (a loaded
    checkpoint) the smaller scale that keeps their hi halves finite; consumers read it from `state` (scale_ptr).
    `get(weight)` returns the SplitTensor view of one conv weight, or None for tensors the owner does not hold.
    `dgrad_weights(...)` caches the re-laid-out, split data-gradient weights of a layer per weight version."""

    def __init__(self, owner):
        self.owner = owner
        self.version = None
        self.buf = None
        self.state = None
        self.views = {}
        self._dgrad = {}
        self._plan = {}
        # frozen: the mirror was refreshed at the start of an update (Council_Trainer._fresh_mirrors) and must not be
        # re-split inside it -- with several member groups on several streams, one group's Adam step moves the pool's
        # version while another group is still reading ITS members' (unchanged) slices of the mirror
        self.frozen = False

    def _storage(self):
        o = self.owner
        if hasattr(o, 'opts'):                       # ParamPool
            if o.data is None:
                return None
            return o.data, [(p, k * o.stride + off) for k, opt in enumerate(o.opts)
                            for p, off in zip(opt._params, opt.flat['offs'])]
        if o._flat is None:
            return None
        return o.flat['data'], list(zip(o._params, o.flat['offs']))

    def refresh(self):
        if self.frozen and self.version is not None:
            return True
        st = self._storage()
        if st is None:
            return False
        if self.version != self.owner.version:
            data, table = st
            total = data.numel()
            if self.buf is None or self.buf.numel() != 2 * total or self.buf.device != data.device:
                self.buf = torch.empty(2 * total, dtype=torch.float16, device=data.device)
                self.state = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=data.device)
            check(_lib().cg_split_f16_dynamic_capped(ptr(data), ptr(self.buf), total, x3_lo(total), ptr(self.state), 0,
                                                     hip.X3_WSCALE, stream()), "cg_split_f16_dynamic_capped")
            self.views = {}
            for p, o in table:
                if p.dim() == 4:
                    self.views[id(p)] = SplitTensor(self.buf, (p.shape[0], p.shape[2], p.shape[3], p.shape[1]), off=o,
                                                    lo=x3_lo(total), scale=1.0, state=self.state)
            self.version = self.owner.version
            self._dgrad = {}
        return True

    def get(self, weight):
        if not self.refresh():
            return None
        return self.views.get(id(weight))

    def dgrad_weights(self, weight, w, g, ci0, nci, grp, n):
        """Prepared ({hi, lo}, per-class [ci][tc][co]) data-gradient weights of `weight` for the current version and
        member scope: one cg_conv2d_dgrad_x3_prep launch per (layer, version) instead of one per backward launch."""
        if not self.refresh() or id(weight) not in self.views:
            return None
        key = (id(weight), ci0, nci, n)
        wt = self._dgrad.get(key)
        if wt is None:
            wt = self._dgrad_build(key, weight, w, g, ci0, nci, grp, n)
            self._plan[key] = (weight, g, ci0, nci, grp, n)       # what prefetch_dgrad prepares for the next weight version
        return wt

    def _dgrad_build(self, key, weight, w, g, ci0, nci, grp, n, on=None):
        """One cg_conv2d_dgrad_x3_prep launch (on stream `on`, default the current one); the buffer is allocated on the CURRENT
        stream either way -- that is where it is read."""
        lib = _lib()
        elems = lib.cg_conv2d_dgrad_x3_wt_elems(byref(g), nci)
        wt = torch.empty(2 * n * elems, dtype=torch.float16, device=w.device)
        check(lib.cg_conv2d_dgrad_x3_prep(byref(g), grp, ptr(w), ci0, nci, 1.0, self.scale_ptr(), ptr(wt), wt.numel() * 2,
                                          c_void_p(on.cuda_stream) if on is not None else stream()), "cg_conv2d_dgrad_x3_prep")
        self._dgrad[key] = wt
        return wt

    def prefetch_dgrad(self, side):
        """Prepare, on stream `side`, the data-gradient weights of every layer that asked for them under the previous weight
        version and has none for the current one.  These are 5 us launches, one per layer and version, that otherwise sit in the
        chain of the backward pass (44 per benchmark step, 62 on a rank with one member); issued at the start of the update they
        run beside the forward pass.  The caller orders `side` behind the current stream first (the weights' scale is written by
        refresh()) and makes the backward wait for the event it records on `side` afterwards.  Returns the number of launches."""
        if not self._plan or not self.refresh():
            return 0
        n_built = 0
        for key, (weight, g, ci0, nci, grp, n) in list(self._plan.items()):
            if key in self._dgrad:
                continue
            if id(weight) not in self.views:
                del self._plan[key]
                continue
            self._dgrad_build(key, weight, nhwc(weight), g, ci0, nci, grp, n, on=side)
            n_built += 1
        return n_built

    def l1_bound(self, weight, bias, grp, n, by_ci):
        """Device float[2] {largest row (by_ci = False) / column (True) 1-norm of `weight` over the members of the scope, largest
        |bias|} for the current weight version (cg_weight_l1_bound): the a-priori bound of a bounded-split epilogue."""
        if not self.refresh() or id(weight) not in self.views:
            return None
        key = (id(weight), 'l1', bool(by_ci), n)
        t = self._dgrad.get(key)
        if t is None:
            t = torch.empty(2, dtype=torch.float32, device=weight.device)
            ws = workspace(_lib().cg_weight_l1_workspace(weight.shape[1], n), slot=3) if by_ci else None
            check(_lib().cg_weight_l1_bound(grp, ptr(nhwc(weight)), weight.shape[0], weight.shape[2] * weight.shape[3],
                                            weight.shape[1], None if by_ci else ptr(bias), int(bool(by_ci)), ptr(t), ptr(ws),
                                            ws.numel() if ws is not None else 0, stream()), "cg_weight_l1_bound")
            self._dgrad[key] = t
        return t

    def upconv_weights(self, weight, grp, n):
        """(wt_fwd, wt_bwd): the summed-tap {hi, lo} weights of an upsample + 3x3 layer (cg_upconv_prep_x3) for the current
        weight version and member scope -- one launch pair per (layer, version)."""
        if not self.refresh() or id(weight) not in self.views:
            return None
        key = (id(weight), 'up', n)
        wt = self._dgrad.get(key)
        if wt is None:
            lib = _lib()
            Cout, Cin = weight.shape[0], weight.shape[1]
            elems = lib.cg_upconv_wt_elems(Cout, Cin)
            wt = (torch.empty(2 * n * elems, dtype=torch.float16, device=weight.device),
                  torch.empty(2 * n * elems, dtype=torch.float16, device=weight.device))
            check(lib.cg_upconv_prep_x3(grp, ptr(nhwc(weight)), Cout, Cin, UPC_WSCALE, self.scale_ptr(), ptr(wt[0]), ptr(wt[1]),
                                        stream()), "cg_upconv_prep_x3")
            self._dgrad[key] = wt
        return wt

    def scale_ptr(self):
        return c_void_p(self.state.data_ptr() + 4)


def x3_eligible(C1, C2):
    return C2 == 0 and C1 % 32 == 0


def split_f16(x, scale=1.0):
    """fp32 NCHW(channels_last) tensor -> SplitTensor"""
    if torch.is_grad_enabled() and x.requires_grad:
        raise hip.HipError("split-precision tensors carry no gradient: use them under torch.no_grad() only")
    x = nhwc(x)
    buf = torch.empty(2 * x.numel(), dtype=torch.float16, device=x.device)
    check(_lib().cg_split_f16(ptr(x), ptr(buf), x.numel(), x3_lo(x.numel()), float(scale), stream()), "cg_split_f16")
    return SplitTensor(buf, x.shape, scale=scale)


def split_f16_dynamic(x, amax=None):
    """fp32 tensor of ARBITRARY magnitude (a gradient, an un-normalised activation) -> SplitTensor whose planes hold
    scale*x, scale = the power of two that puts max|x| into [4096, 8192), chosen on the device (no host sync)."""
    x = nhwc(x) if x.dim() == 4 else x.contiguous()
    buf = torch.empty(2 * x.numel(), dtype=torch.float16, device=x.device)
    if amax is not None:            # (state, nslots) left behind by the kernel that produced x
        state, nslots = amax
    else:
        state, nslots = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=x.device), 0
    check(_lib().cg_split_f16_dynamic(ptr(x), ptr(buf), x.numel(), x3_lo(x.numel()), ptr(state), nslots, stream()),
          "cg_split_f16_dynamic")
    return SplitTensor(buf, x.shape, state=state)


# CG_ACT_BWD_AMAX=0: the activation backward always measures dz itself (A/B switch)
ACT_BWD_AMAX = os.environ.get("CG_ACT_BWD_AMAX", "1") != "0"


def act_bwd_split(dy, y, act, want_fp32, amax=None):
    """dz = dy * act'(y) as a dynamically scaled SplitTensor (+ the fp32 tensor when a non-split kernel still needs it).
    `amax`: (state, nslots) the producer of dy left behind (the data-gradient kernel of the next layer): its per-block
    maxima bound |dz| (|act'| <= 1), so the measuring pass over (dy, y) is skipped."""
    buf = torch.empty(2 * dy.numel(), dtype=torch.float16, device=dy.device)
    dz = torch.empty_like(dy) if want_fp32 else None
    if amax is not None and not want_fp32 and ACT_BWD_AMAX:
        state, nslots = amax
    else:
        state, nslots = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=dy.device), 0
    check(_lib().cg_act_bwd_split(ptr(dy), ptr(y), dy.numel(), act, ptr(buf), x3_lo(dy.numel()), ptr(state), nslots, ptr(dz),
                                  stream()), "cg_act_bwd_split")
    return dz, SplitTensor(buf, dy.shape, state=state)


def conv_dgrad_x3(g, dz, w, ci0, nci, grp=None, weight=None, wmgr=None, nm=1, fuse=None, xsplit=None, addend=None):
    """conv_dgrad on the split-precision kernel: dz is split with its device-side scale; needs Cout % 32 == 0.  The weights
    are re-laid-out and split per launch into the workspace -- or, for a parameter of a SplitWeights-managed optimizer
    (`weight`, `wmgr`), once per weight version (SplitWeights.dgrad_weights)."""
    lib = _lib()
    N, H, W, up = g.N, g.H, g.W, g.up
    dzs = dz if isinstance(dz, SplitTensor) else split_f16_dynamic(dz)
    dxl = torch.empty((N, nci, H << up, W << up), dtype=torch.float32, device=dzs.buf.device, memory_format=CL)
    wt = wmgr.dgrad_weights(weight, w, g, ci0, nci, grp, nm) if (wmgr is not None and weight is not None) else None
    if wt is not None and fuse is not None and xsplit is not None and dzs.state is not None and not up and x3_interleaved():
        # The layer below (the producer of this layer's input x) has a sign-only activation fused into its convolution and asked
        # for dz = dx * act'(x) directly: fold the activation backward into this data gradient's epilogue (the sign comes from
        # x's hi plane) and, when that layer reads dz as {hi, lo} planes, write them on the a-priori scale
        # L1(columns of W) * max|dz_in| -- no fp32 dx, no activation / split pass over it (cg_x3_epilogue)
        act_b, want_split, want_f32, cell = fuse
        ctl = wmgr.l1_bound(weight, None, grp, nm, True) if want_split else None
        if not want_split or ctl is not None:
            cell[0] += 1
            state = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=dxl.device)
            buf = torch.empty(2 * dxl.numel(), dtype=torch.float16, device=dxl.device) if want_split else None
            epi = hip.X3Epilogue(ctl.data_ptr() if want_split else None, dzs.state.data_ptr(), int(dzs.nslots), int(act_b),
                                 xsplit.hi_ptr(), state.data_ptr())
            nslots = ctypes.c_int(0)
            check(lib.cg_conv2d_dgrad_x3_run_e(byref(g), grp, dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(), ptr(wt), 1.0, wmgr.scale_ptr(),
                                               ci0, nci, ptr(dxl) if want_f32 else None, ptr(buf), x3_lo(dxl.numel()),
                                               byref(epi), ptr(state), byref(nslots), stream()), "cg_conv2d_dgrad_x3_run_e")
            if want_split:
                if nslots.value == 0:
                    raise hip.HipError("fused activation backward: the kernel reported no output maxima")
                dxl._cg_dz_split = SplitTensor(buf, dxl.shape, state=state, nslots=nslots.value)
                dxl._cg_dz_version = dxl._version
            elif nslots.value:
                dxl._cg_amax = (state, nslots.value, dxl._version)
            dxl._cg_act_applied = dxl._version
            dxl._cg_has_f32 = bool(want_f32)
            return dxl
    if wt is not None:
        # the kernel leaves the per-block maxima of dx behind: the layer below splits its dz without measuring it again
        state = torch.empty(hip.SPLIT_STATE_FLOATS, dtype=torch.float32, device=dxl.device) if not up else None
        nslots = ctypes.c_int(0)
        if addend is not None and not up and x3_interleaved():
            # the skip connection's gradient joins dx in the kernel's epilogue (cg_x3_epilogue.addend): no pass of its own
            if tuple(addend.shape) != tuple(dxl.shape) or addend.stride() != dxl.stride() or addend.dtype != torch.float32:
                raise hip.HipError("skip-connection gradient: shape / layout differs from the data gradient it joins")
            epi = hip.X3Epilogue(None, None, 0, 0, None, None, addend.data_ptr())
            check(lib.cg_conv2d_dgrad_x3_run_e(byref(g), grp, dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(), ptr(wt), 1.0, wmgr.scale_ptr(),
                                               ci0, nci, ptr(dxl), None, 0, byref(epi), ptr(state), byref(nslots), stream()),
                  "cg_conv2d_dgrad_x3_run_e")
            addend = None
        else:
            check(lib.cg_conv2d_dgrad_x3_run(byref(g), grp, dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(), ptr(wt), 1.0, wmgr.scale_ptr(),
                                             ci0, nci, ptr(dxl), ptr(state), byref(nslots) if state is not None else None, stream()),
                  "cg_conv2d_dgrad_x3_run")
        if nslots.value:
            dxl._cg_amax = (state, nslots.value, dxl._version)
    else:
        if grp is not None:
            raise hip.HipError("member-batched split-precision data gradient needs pool-managed weights")
        ws = workspace(lib.cg_conv2d_dgrad_workspace(byref(g), nci))
        check(lib.cg_conv2d_dgrad_x3(byref(g), dzs.hi_ptr(), dzs.lo, dzs.scale_ptr(), ptr(w), ci0, nci, ptr(dxl), ptr(ws),
                                     ws.numel(), stream()), "cg_conv2d_dgrad_x3")
    if not up:
        return dxl if addend is None else add(dxl, addend)
    dx = empty_nhwc(N, nci, H, W, dxl)
    check(lib.cg_upsample2x_bwd(ptr(dxl), ptr(dx), N, H, W, nci, stream()), "cg_upsample2x_bwd")
    return dx if addend is None else add(dx, addend)


def conv2d_x3(xs, wsplit, Cout, KH, KW, bias=None, stride=1, pad=0, act="none", upsample=False, stats=None, grp=None):
    """act(conv2d(zero_pad(x), W) + bias) on the fp16 MFMA with every product expanded as ah*bh + ah*bl + al*bh.
    xs: SplitTensor activation; wsplit: SplitTensor over the physical [Cout][KH][KW][Cin] weight.  No autograd."""
    if torch.is_grad_enabled() and (bias is not None and bias.requires_grad):
        raise hip.HipError("conv2d_x3 has no backward: call it under torch.no_grad()")
    lib = _lib()
    N, C1, H, W = xs.shape
    g = fwd_geom(N, H, W, C1, 0, int(bool(upsample)), KH, KW, stride, pad, Cout, ACT[act])
    y = torch.empty((N, Cout, g.Ho, g.Wo), dtype=torch.float32, device=xs.buf.device, memory_format=CL)
    rows = ctypes.c_int(0)
    sws, sbytes, rp = None, 0, None
    if stats is not None and ACT[act] == 0:
        m = N * g.Ho * g.Wo
        sws = workspace(((m + 63) // 64) * Cout * 16, slot=1)
        sbytes, rp = sws.numel(), byref(rows)
    if xs.scale != 1.0:
        raise hip.HipError("conv2d_x3: activations carry a static scale of 1 or a device-side one")
    check(lib.cg_conv2d_fwd_x3_g(byref(g), grp, xs.hi_ptr(), xs.lo, wsplit.hi_ptr(), wsplit.lo, float(wsplit.scale),
                                 wsplit.scale_ptr(), xs.scale_ptr(), ptr(bias), ptr(y), None, 0, ptr(sws), sbytes, rp, -1, None,
                                 None, stream()), "cg_conv2d_fwd_x3")
    if stats is not None and rows.value:
        stats.append((sws, rows.value))
    return y


# CG_FUSED_HEAD=0: the decoder's 1x1 head runs layer by layer in the tape-free passes too (A/B switch)
FUSED_HEAD = os.environ.get("CG_FUSED_HEAD", "1") != "0"


def decoder_head_x3(xs, convs, wmgr, im_in, out_dim, nmask):
    """The decoder's three 1x1 convolutions + mask / blend head (networks.py:393-407) as one kernel, for passes without a tape.
    xs: SplitTensor of the trunk output; convs: the three nn.Conv2d holders (pool-managed); returns (image, mask) or None when
    the fused kernel does not take this shape (the caller then runs the layers one by one)."""
    if not (FUSED_HEAD and x3_interleaved() and xs.state is None and xs.scale == 1.0):
        return None
    N, C, H, W = xs.shape
    if C != 64 or out_dim != 3 or nmask != 3 or [c.out_channels for c in convs] != [64, 64, out_dim * nmask + nmask] or \
            any(c.kernel_size != (1, 1) or c.in_channels != 64 for c in convs):
        return None
    ws = [wmgr.get(c.weight) for c in convs]
    if any(w is None for w in ws):
        return None
    im_in = nhwc(im_in)
    im_out = empty_nhwc(N, out_dim, H, W, im_in)
    mask = empty_nhwc(N, nmask, H, W, im_in)
    check(_lib().cg_decoder_head_fwd_x3(xs.hi_ptr(), xs.lo, ws[0].hi_ptr(), ws[1].hi_ptr(), ws[2].hi_ptr(), x3_lo(1),
                                        float(ws[0].scale), ws[0].scale_ptr(), ptr(convs[0].bias), ptr(convs[1].bias),
                                        ptr(convs[2].bias), _grp(convs[0].weight), ptr(im_in), ptr(im_out), ptr(mask),
                                        N * H * W, C, out_dim, nmask, stream()), "cg_decoder_head_fwd_x3")
    return im_out, mask


def instnorm_split(x, params, goff, boff, act="none", residual=None, eps=1e-5, stats=None, want_f32=False):
    """IN / AdaIN apply (no autograd) whose output is produced in split form (and in fp32 too when `want_f32`:
    the ResBlock skip connection and the fp32 consumers need it).  Returns (y_fp32_or_None, SplitTensor)."""
    lib = _lib()
    x, residual = nhwc(x), nhwc(residual)
    N, C, H, W = x.shape
    HW = H * W
    mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    if stats:
        part, rows = stats[0]
        check(lib.cg_instnorm_stats_from_partials(ptr(part), N, HW, C, rows, eps, ptr(mean), ptr(rstd), stream()),
              "cg_instnorm_stats_from_partials")
    else:
        ws = workspace(lib.cg_instnorm_workspace(N, HW, C))
        check(lib.cg_instnorm_stats(ptr(x), N, HW, C, eps, ptr(mean), ptr(rstd), ptr(ws), ws.numel(), stream()),
              "cg_instnorm_stats")
    if params is not None:
        params = params.contiguous()
        gp, bp, gs = _off(params, goff), _off(params, boff), params.shape[1]
    else:
        gp, bp, gs = None, None, C
    y = torch.empty_like(x) if want_f32 else None
    ys = SplitTensor(torch.empty(2 * x.numel(), dtype=torch.float16, device=x.device), x.shape)
    check(lib.cg_instnorm_apply_split(ptr(x), ptr(mean), ptr(rstd), gp, bp, gs, ptr(residual), ptr(y), ys.hi_ptr(), ys.lo,
                                      N, HW, C, ACT[act], stream()), "cg_instnorm_apply_split")
    return y, ys


# ------------------------------------------------------------------------------------------
# resampling
# ------------------------------------------------------------------------------------------
class _AvgPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib()
        x = nhwc(x)
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x)
        check(lib.cg_avgpool3s2_fwd(ptr(x), ptr(y), N, H, W, C, stream()), "cg_avgpool3s2_fwd")
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        N, C, H, W = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(N, C, H, W, dy)
        check(lib.cg_avgpool3s2_bwd(ptr(dy), ptr(dx), N, H, W, C, stream()), "cg_avgpool3s2_bwd")
        return dx


def avgpool3s2(x):
    return _AvgPool3s2.apply(x)


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib()
        x = nhwc(x)
        N, C, H, W = x.shape
        y = empty_nhwc(N, C, 2 * H, 2 * W, x)
        check(lib.cg_upsample2x_fwd(ptr(x), ptr(y), N, H, W, C, stream()), "cg_upsample2x_fwd")
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        N, C, H, W = ctx.shape
        dy = nhwc(dy)
        dx = empty_nhwc(N, C, H, W, dy)
        check(lib.cg_upsample2x_bwd(ptr(dy), ptr(dx), N, H, W, C, stream()), "cg_upsample2x_bwd")
        return dx


def upsample2x(x):
    return _Upsample2x.apply(x)


def global_avgpool(x):
    """nn.AdaptiveAvgPool2d(1), networks.py:347.  Forward only: the style code never receives a
    gradient on the shipped configs (recon_s_w = 0)."""
    lib = _lib()
    if x.requires_grad and torch.is_grad_enabled():
        x = x.detach()
    x = nhwc(x)
    N, C, H, W = x.shape
    y = empty_nhwc(N, C, 1, 1, x)
    check(lib.cg_global_avgpool_fwd(ptr(x), ptr(y), N, H * W, C, stream()), "cg_global_avgpool_fwd")
    return y


# ------------------------------------------------------------------------------------------
# mask / blend head (networks.py:398-407)
# ------------------------------------------------------------------------------------------
class _MaskBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, new_x, im_in, od, k):
        lib = _lib()
        new_x, im_in = nhwc(new_x), nhwc(im_in)
        N, CH, H, W = new_x.shape
        if CH != od * k + k or im_in.shape[1] != od:
            raise ValueError("mask/blend head expects %d channels" % (od * k + k))
        im_out = empty_nhwc(N, od, H, W, new_x)
        mask = empty_nhwc(N, k, H, W, new_x)
        check(lib.cg_mask_blend_fwd(ptr(new_x), ptr(im_in), ptr(im_out), ptr(mask), N * H * W, od, k, stream()),
              "cg_mask_blend_fwd")
        ctx.save_for_backward(new_x, im_in)
        ctx.meta = (od, k)
        ctx.set_materialize_grads(False)
        return im_out, mask

    @staticmethod
    def backward(ctx, d_im, d_mask):
        lib = _lib()
        new_x, im_in = ctx.saved_tensors
        od, k = ctx.meta
        N, CH, H, W = new_x.shape
        if d_im is None:
            d_im = empty_nhwc(N, od, H, W, new_x)
            check(lib.cg_fill(ptr(d_im), d_im.numel(), 0.0, stream()), "cg_fill")
        d_im, d_mask = nhwc(d_im), nhwc(d_mask)
        d_new = torch.empty_like(new_x)
        check(lib.cg_mask_blend_bwd(ptr(new_x), ptr(im_in), ptr(d_im), ptr(d_mask), ptr(d_new), N * H * W, od, k,
                                    stream()), "cg_mask_blend_bwd")
        return d_new, None, None, None


def mask_blend(new_x, im_in, od, k):
    return _MaskBlend.apply(new_x, im_in, int(od), int(k))


# ------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------
class _Lsgan(torch.autograd.Function):
    """sum over scales of  sum_s wt[s] * mean_hw (o - tgt[s])^2 / group   (networks.py:64,90,166,194).
    tgt / wt are device vectors with one entry per sample of the (batched) discriminator input.  Under ops.members(n)
    the samples are n consecutive member blocks and the result is a vector of n per-member losses."""

    @staticmethod
    def forward(ctx, tgt, wt, group, nm, *outs):
        lib = _lib()
        loss = torch.empty(nm, dtype=torch.float32, device=tgt.device)
        outs = [o.contiguous() for o in outs]
        for i, o in enumerate(outs):
            nb = o.shape[0]
            hw = o.numel() // nb
            check(lib.cg_lsgan_fwd_g(ptr(o), ptr(tgt), ptr(wt), nb, hw, group, nm, ptr(loss), int(i > 0), stream()),
                  "cg_lsgan_fwd")
        ctx.save_for_backward(tgt, wt, *outs)
        ctx.group, ctx.nm = group, nm
        return loss if nm > 1 else loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        tgt, wt, *outs = ctx.saved_tensors
        g = g.contiguous()
        grads = []
        for o in outs:
            nb = o.shape[0]
            hw = o.numel() // nb
            d = torch.empty_like(o)
            check(lib.cg_lsgan_bwd_g(ptr(o), ptr(tgt), ptr(wt), ptr(g), nb, hw, ctx.group, ctx.nm, ptr(d), stream()),
                  "cg_lsgan_bwd")
            grads.append(d)
        return (None, None, None, None) + tuple(grads)


def lsgan_loss(outs, tgt, wt, group):
    """`group` = samples per member that count as one batch mean (the reference's per-call batch size)."""
    return _Lsgan.apply(tgt, wt, int(group), _G.n, *outs)


class _FocusLoss(torch.autograd.Function):
    """w_zo * mask_zero_one + w_total * mask_small + w_tv * TV   (trainer_council.py:230-250).
    Returns (total, parts[3]) -- parts = the three unweighted criteria, for logging; under ops.members(n): total [n],
    parts [n, 3], every member's criteria over its own block of masks."""

    @staticmethod
    def forward(ctx, mask, center, eps, w_zo, w_total, w_tv, use_abs, use_square, reduce, nm):
        lib = _lib()
        mask = nhwc(mask)
        N, k, H, W = mask.shape
        sums = torch.empty(3 * nm, dtype=torch.float32, device=mask.device)
        out = torch.empty(4 * nm, dtype=torch.float32, device=mask.device)
        ws = workspace(nm * lib.cg_focus_workspace())
        check(lib.cg_focus_sums_g(ptr(mask), N, H, W, k, nm, center, eps, ptr(sums), ptr(ws), ws.numel(), stream()),
              "cg_focus_sums")
        if reduce is not None:
            # data parallelism inside a member: with the MEAN of the replicas' sums every rank evaluates the full-batch
            # criteria (the squared mask mean is not linear in the batch) and back-propagates world-size times its
            # share, which the gradient averaging turns into the full-batch gradient
            reduce(sums)
        check(lib.cg_focus_total_g(ptr(sums), mask.numel() // nm, nm, w_zo, w_total, w_tv, int(use_abs), int(use_square),
                                   ptr(out), stream()), "cg_focus_total")
        ctx.save_for_backward(mask, sums)
        ctx.meta = (center, eps, w_zo, w_total, w_tv, int(use_abs), int(use_square), nm)
        out = out.view(nm, 4)
        total, parts = (out[:, 0], out[:, 1:]) if nm > 1 else (out[0, 0], out[0, 1:])
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for `parts` (an ATen fill per backward)
        return total, parts

    @staticmethod
    def backward(ctx, g, _gparts):
        if g is None:
            return (None,) * 10
        lib = _lib()
        mask, sums = ctx.saved_tensors
        center, eps, w_zo, w_total, w_tv, use_abs, use_square, nm = ctx.meta
        N, k, H, W = mask.shape
        g = g.contiguous()
        d = torch.empty_like(mask)
        check(lib.cg_focus_bwd_g(ptr(mask), ptr(sums), ptr(g), N, H, W, k, nm, center, eps, w_zo, w_total, w_tv, use_abs,
                                 use_square, ptr(d), stream()), "cg_focus_bwd")
        return (d,) + (None,) * 9


def focus_loss(mask, center, eps, w_zo, w_total, w_tv, use_abs, use_square, reduce=None):
    return _FocusLoss.apply(mask, float(center), float(eps), float(w_zo), float(w_total), float(w_tv),
                            bool(use_abs), bool(use_square), reduce, _G.n)


class _L1Mean(torch.autograd.Function):
    """mean |a - b| (trainer_council.py:207-208); gradient flows to `a` only."""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib()
        if a.shape != b.shape:
            raise ValueError("l1_mean: shape mismatch")
        fmt = CL if a.dim() == 4 else torch.contiguous_format
        a, b = a.contiguous(memory_format=fmt), b.contiguous(memory_format=fmt)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        check(lib.cg_l1_mean_fwd(ptr(a), ptr(b), a.numel(), ptr(loss), stream()), "cg_l1_mean_fwd")
        ctx.save_for_backward(a, b)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        check(lib.cg_l1_mean_bwd(ptr(a), ptr(b), ptr(g.contiguous()), a.numel(), ptr(da), stream()), "cg_l1_mean_bwd")
        return da, None


def l1_mean(a, b):
    return _L1Mean.apply(a, b)


# ------------------------------------------------------------------------------------------
# utilities
# ------------------------------------------------------------------------------------------
def fill_(t, value):
    check(_lib().cg_fill(ptr(t), t.numel(), float(value), stream()), "cg_fill")
    return t


_idx_cache = {}


def _index_tensor(idx, device):
    key = (tuple(idx), str(device))
    t = _idx_cache.get(key)
    if t is None:
        # entries are never dropped once a hipGraph has been captured (graphs hold the addresses; a few hundred bytes per
        # distinct shape plan), bounded otherwise (hip.const_cache_put).  The upload completes before the tensor is published
        # -- the cache is read from every stream (hip.upload_const)
        t = hip.const_cache_put(_idx_cache, key, hip.upload_const(torch.tensor(list(idx), dtype=torch.int32)))
    return t


def take_rows(a, b, idx, out=None, idx_dev=None):
    """out[i] = a[idx[i]] if idx[i] >= 0 else b[-idx[i] - 1] along the batch dimension (NHWC rows): how the discriminator
    batches [own fake | real] / [own translation | colleagues' translations] of all members are assembled -- one copy
    kernel instead of torch.cat.  `idx` is a host sequence (cached on the device by value) -- or, with `idx_dev` (an int32
    device vector the caller keeps current, graphs.HostInputs), just its length."""
    a = nhwc(a) if a.dim() == 4 else a.contiguous()
    if b is not None:
        b = nhwc(b) if b.dim() == 4 else b.contiguous()
    row = a[0].numel()
    if row % 4:
        raise ValueError("take_rows: rows must be a multiple of 4 floats")
    if b is not None and b[0].numel() != row:
        raise ValueError("take_rows: sources differ in row size")
    if out is None:
        shape = (len(idx),) + tuple(a.shape[1:])
        out = torch.empty(shape, dtype=torch.float32, device=a.device, memory_format=CL) if a.dim() == 4 else \
            torch.empty(shape, dtype=torch.float32, device=a.device)
    elif out.shape[0] != len(idx) or out[0].numel() != row or not (out.is_contiguous(memory_format=CL) if out.dim() == 4
                                                                    else out.is_contiguous()):
        raise ValueError("take_rows: `out` must be a dense NHWC block of len(idx) rows")
    if idx_dev is not None and (idx_dev.dtype != torch.int32 or idx_dev.numel() != len(idx)):
        raise ValueError("take_rows: idx_dev must be an int32 vector of len(idx) entries")
    check(_lib().cg_gather_rows2(ptr(a), ptr(b), ptr(idx_dev if idx_dev is not None else _index_tensor(idx, a.device)),
                                 ptr(out), len(idx), row, stream()), "cg_gather_rows2")
    return out


def gen_total(focus4, adv, lc, w_match, gan_w, council_w, n):
    """Per-member generator objective and the pieces train.py logs (cg_gen_total): returns (total[n], council[n],
    gcouncil[n]); any of focus4 ([n, 4] from focus_loss), adv, lc, w_match may be None."""
    dev = next(t for t in (focus4, adv, lc) if t is not None).device
    total = torch.empty(n, dtype=torch.float32, device=dev)
    council = torch.empty(n, dtype=torch.float32, device=dev)
    gcouncil = torch.empty(n, dtype=torch.float32, device=dev)
    check(_lib().cg_gen_total(ptr(focus4), ptr(adv), ptr(lc), ptr(w_match), float(gan_w), float(council_w), ptr(total),
                              ptr(council), ptr(gcouncil), n, stream()), "cg_gen_total")
    return total, council, gcouncil


def gather_rows(src, idx_dev, nidx):
    """out[i] = src[idx[i]] along dim 0 (colleague pick after the council exchange)."""
    row = src[0].numel()
    out = torch.empty((nidx,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if src.dim() == 4 and src.is_contiguous(memory_format=CL):
        out = torch.empty((nidx,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device, memory_format=CL)
    check(_lib().cg_gather_rows(ptr(src), ptr(idx_dev), ptr(out), nidx, row, stream()), "cg_gather_rows")
    return out
