"""Autograd glue over the C ABI for the drop-in op surface (nn_core_operator / loss_terms_and_measures).

Tensors at this level are channels_last ``[B, T, C]`` float32 CUDA tensors, like the reference's TF tensors.
Every forward/backward below is one or a few libnsc_hip.so launches on the current stream; torch provides memory
and the autograd tape only.  No CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, check

ACT = {None: 0, "none": 0, "tanh": 1, "lrelu": 2}
# arithmetic of the gated block's weight gradients behind BlockFn: the engine's switch (engine.py: split_wgrad_arith)
SPLIT_ARITH = os.environ.get("NSC_BLOCK_ARITH", "split") == "split"


def _lib_():
    return _lib.load()


def _st():
    """Raw handle of the current HIP stream (torch.cuda.current_stream() builds a Stream object: ~8 us of host time per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _req(t, name="tensor"):
    """Type check only.  LAYOUT: the kernels work on time-contiguous [B,C,T] memory, the surface speaks channels_last [B,T,C].  An op
    hands its [B,C,T] result back as the TRANSPOSED VIEW `y.transpose(1, 2)` - shape and values of the [B,T,C] tensor the reference
    would return, no copy - and the next op recognises such a view (`to_bct`) and uses the memory as it lies: a chain of surface ops
    transposes nothing.  A tensor that really is channels_last in memory (user data, a torch op in between) is transposed once on
    the way in (nsc_transpose_last2)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise _lib.NscError(f"{name}: nsc_amd ops need float32 CUDA tensors (there is no CPU fallback)")
    return t


def _is_lazy_bct(x):
    return x.dim() == 3 and x.shape[-1] > 1 and x.shape[1] > 1 and not x.is_contiguous() and x.transpose(1, 2).is_contiguous()


def _mem(x):
    """The tensor as contiguous memory + whether that memory is [B,C,T] behind a [B,T,C] view (elementwise ops keep the layout)."""
    if _is_lazy_bct(x):
        return x.transpose(1, 2), True
    return x.contiguous(), False


def _same_mem(a, b):
    """Two same-shaped operands of an elementwise op in ONE memory order: [B,C,T] if either already lies that way."""
    ma, la = _mem(a)
    mb, lb = _mem(b)
    if la == lb:
        return ma, mb, la
    if la:
        return ma, to_bct(b), True
    return to_bct(a), mb, True


def _workspace(floats, dev):
    """Caller-owned scratch of the slab-flush weight-gradient kernels: one buffer per device, grown on demand (launches on one stream
    are ordered, so they can share it; the old buffer of a grow stays alive until the kernels using it have run - torch's
    stream-ordered allocator)."""
    #  one buffer per (device, STREAM): two backward passes on two streams would race on a shared slab; and never allocated
    # or grown while a hipGraph is being captured - the buffer would belong to the graph's private pool and be handed out again after
    # the graph is freed: the first eager use sizes it, a capture that needs more says so
    key = ("ws", str(dev), int(_st()))
    ws = _CACHE.get(key)
    if ws is None or ws.numel() < floats:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"nsc_amd.ops: the weight-gradient workspace ({floats} floats) must exist before graph capture: run the "
                               f"step once eagerly on this stream first")
        ws = torch.empty(max(floats, 1), dtype=torch.float32, device=dev)
        _CACHE[key] = ws
    return ws


def same_pad(T, k, dil=1, stride=1):
    t_out = -(-T // stride)
    pad = max((t_out - 1) * stride + (k - 1) * dil + 1 - T, 0)
    return t_out, pad // 2


def to_bct(x):
    """[B,T,C] -> contiguous [B,C,T]: free when C == 1 or when x is the transposed view an op returned; else one transpose launch."""
    B, T, Cc = x.shape
    if Cc == 1 or T == 1:
        return x.contiguous().reshape(B, Cc, T)
    if _is_lazy_bct(x):
        return x.transpose(1, 2)
    x = x.contiguous()
    y = torch.empty((B, Cc, T), dtype=x.dtype, device=x.device)
    check(_lib_().nsc_transpose_last2(x.data_ptr(), y.data_ptr(), B, T, Cc, _st()), "transpose")
    return y


def to_btc(x):
    """contiguous [B,C,T] -> the [B,T,C] tensor of the surface, as a VIEW of the same memory (see _req)."""
    B, Cc, T = x.shape
    if Cc == 1:
        return x.reshape(B, T, 1)
    return x.transpose(1, 2)


def _desc(B, Cin, Cout, Tin, Tout, K, dil, stride, padL, **kw):
    d = ConvDesc(B=B, Cin=Cin, Cout=Cout, Tin=Tin, Tout=Tout, K=K, dil=dil, stride=stride, padL=padL, act=0, res_mode=0,
                 mul_mode=0, out_mode=0, in_up=0, accumulate=0)
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def _split_conv_image(lib, which, d, w):
    """Kernel-ready image of a stride-2 k9 100 -> 100 kernel for the split-operand conv kernels (csrc/conv_split.hip): one gather launch
    from the kernel tensor itself; the index map is cached per device.  None: this shape is not served (or NSC_BLOCK_ARITH=exact)."""
    if not SPLIT_ARITH:
        return None
    n = int(lib.nsc_conv1d_simage_words(which, C.byref(d)))
    if n <= 0 or w.data_ptr() % 16:
        return None
    key = ("cs_idx", which, str(w.device), tuple(w.shape))
    if key not in _CACHE:
        idx = np.empty(n, np.int32)
        check(lib.nsc_conv1d_simage_index(which, C.byref(d), 0, idx.ctypes.data_as(C.c_void_p)), "conv1d_simage_index")
        _CACHE[key] = torch.from_numpy(idx).to(w.device)
    img = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.nsc_gather(w.data_ptr(), _CACHE[key].data_ptr(), img.data_ptr(), n, _st()), "gather")
    return img


class Conv1dFn(torch.autograd.Function):
    """tf.compat.v1.layers.conv1d(padding='SAME', channels_last) + bias + activation.  The stride-2 down-sampling conv takes the
    split-operand kernels of the engine (forward, data gradient, weight gradient) like the gated blocks do."""

    @staticmethod
    def forward(ctx, x, w, b, dil, stride, act):
        lib = _lib_()
        x, w, b = _req(x, "inputs"), _req(w, "kernel").contiguous(), _req(b, "bias").contiguous()
        B, T, Cin = x.shape
        K, Cin2, Cout = w.shape
        if Cin != Cin2:
            raise ValueError(f"conv1d: input has {Cin} channels, kernel expects {Cin2}")
        Tout, padL = same_pad(T, K, dil, stride)
        xb = to_bct(x)
        y = torch.empty((B, Cout, Tout), dtype=torch.float32, device=x.device)
        d = _desc(B, Cin, Cout, T, Tout, K, dil, stride, padL, act=ACT[act])
        img = _split_conv_image(lib, 0, d, w) if stride == 2 else None
        if img is not None and xb.data_ptr() % 16 == 0:
            check(lib.nsc_conv1d_fwd_simg(C.byref(d), xb.data_ptr(), img.data_ptr(), b.data_ptr(), y.data_ptr(), _st()), "conv1d (split)")
        else:
            fn = lib.nsc_conv1d_cout1_fwd if Cout == 1 else lib.nsc_conv1d_fwd
            check(fn(C.byref(d), xb.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), _st()), "conv1d")
        ctx.save_for_backward(xb, w, y)
        ctx.cfg = (dil, stride, act, padL, T, Tout)
        return to_btc(y)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_()
        xb, w, y = ctx.saved_tensors
        dil, stride, act, padL, T, Tout = ctx.cfg
        B, Cin, _ = xb.shape
        K, _, Cout = w.shape
        dz = to_bct(_req(dy, "grad"))
        if ACT[act]:
            dz2 = torch.empty_like(dz)
            check(lib.nsc_act_bwd(dz.data_ptr(), y.data_ptr(), dz2.data_ptr(), dz.numel(), ACT[act], _st()), "act_bwd")
            dz = dz2
        dwb = torch.zeros(w.numel() + Cout, dtype=torch.float32, device=w.device)      # dw | db
        dw, db = dwb[:w.numel()].view(w.shape), dwb[w.numel():]
        # partial sums of the (b,t) splits go to private slabs + one reduce launch (nsc_conv1d_wgrad_ws) instead of same-address atomics
        dfw = _desc(B, Cin, Cout, T, Tout, K, dil, stride, padL)
        split = (stride == 2 and SPLIT_ARITH and int(lib.nsc_conv1d_simage_words(0, C.byref(dfw))) > 0 and
                 (xb.data_ptr() | dz.data_ptr() | w.data_ptr()) % 16 == 0)
        if split:
            ws = _workspace(int(lib.nsc_conv1d_wgrad_split_workspace()), w.device)
            job = _lib.ConvWgradJob(dfw, xb.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0)
            check(lib.nsc_conv1d_wgrad_split((_lib.ConvWgradJob * 1)(job), 1, ws.data_ptr(), ws.numel(), _st()), "wgrad (split)")
        elif Cout == 1:
            d = _desc(B, 1, Cin, Tout, T, K, dil, 1, (K - 1) * dil - padL)
            ws = _workspace(int(lib.nsc_conv1d_wgrad_workspace(C.byref(d))), w.device)
            check(lib.nsc_conv1d_wgrad_ws(C.byref(d), dz.data_ptr(), xb.data_ptr(), dw.data_ptr(), None, 1, ws.data_ptr(), ws.numel(), _st()),
                  "wgrad")
            check(lib.nsc_sum_all(dz.data_ptr(), db.data_ptr(), dz.numel(), _st()), "bias grad")
        else:
            d = _desc(B, Cin, Cout, T, Tout, K, dil, stride, padL)
            ws = _workspace(int(lib.nsc_conv1d_wgrad_workspace(C.byref(d))), w.device)
            check(lib.nsc_conv1d_wgrad_ws(C.byref(d), xb.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, ws.data_ptr(),
                                          ws.numel(), _st()), "wgrad")
        dx = None
        img = _split_conv_image(lib, 1, dfw, w) if (split and ctx.needs_input_grad[0]) else None
        if img is not None:
            dxb = torch.empty((B, Cin, T), dtype=torch.float32, device=w.device)
            check(lib.nsc_conv1d_dgrad_simg(C.byref(dfw), dz.data_ptr(), img.data_ptr(), dxb.data_ptr(), _st()), "dgrad (split)")
            dx = to_btc(dxb)
        elif ctx.needs_input_grad[0]:
            wt = torch.empty((K, Cout, Cin), dtype=torch.float32, device=w.device)
            check(lib.nsc_weight_flip_transpose(w.data_ptr(), wt.data_ptr(), K, Cin, Cout, _st()), "flip")
            dxb = torch.empty((B, Cin, T), dtype=torch.float32, device=w.device)
            d = _desc(B, Cout, Cin, Tout, T, K, dil, 1, (K - 1) * dil - padL, in_up=1 if stride == 2 else 0)
            fn = lib.nsc_conv1d_cout1_fwd if Cin == 1 else lib.nsc_conv1d_fwd
            check(fn(C.byref(d), dz.data_ptr(), wt.data_ptr(), None, None, None, dxb.data_ptr(), _st()), "dgrad")
            dx = to_btc(dxb)
        return dx, dw, db, None, None, None


class DepthwiseFn(torch.autograd.Function):
    """Depthwise stage of Keras SeparableConv1D (multiplier 1, SAME, stride 1)."""

    @staticmethod
    def forward(ctx, x, wd):
        x, wd = _req(x), _req(wd).contiguous()
        B, T, Cc = x.shape
        K = wd.shape[0]
        xb = to_bct(x)
        y = torch.empty_like(xb)
        check(_lib_().nsc_depthwise_fwd(xb.data_ptr(), wd.data_ptr(), y.data_ptr(), B, Cc, T, K, _st()), "depthwise")
        ctx.save_for_backward(xb, wd)
        return to_btc(y)

    @staticmethod
    def backward(ctx, dy):
        xb, wd = ctx.saved_tensors
        B, Cc, T = xb.shape
        K = wd.shape[0]
        dyb = to_bct(_req(dy))
        dx = torch.empty_like(xb)
        dwd = torch.zeros_like(wd)
        check(_lib_().nsc_depthwise_bwd(xb.data_ptr(), wd.data_ptr(), dyb.data_ptr(), dx.data_ptr(), dwd.data_ptr(), B, Cc,
                                        T, K, _st()), "depthwise_bwd")
        return to_btc(dx), dwd


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        xm, lazy = _mem(_req(x))
        y = torch.empty_like(xm)
        check(_lib_().nsc_act_fwd(xm.data_ptr(), y.data_ptr(), xm.numel(), ACT[act], _st()), "act")
        ctx.save_for_backward(y)
        ctx.act, ctx.lazy = act, lazy
        return y.transpose(1, 2) if lazy else y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dym = to_bct(_req(dy)) if ctx.lazy else _req(dy).contiguous()
        dx = torch.empty_like(dym)
        check(_lib_().nsc_act_bwd(dym.data_ptr(), y.data_ptr(), dx.data_ptr(), dym.numel(), ACT[ctx.act], _st()), "act_bwd")
        return (dx.transpose(1, 2) if ctx.lazy else dx), None


class MulFn(torch.autograd.Function):
    """tf.multiply of two same-shaped tensors (the GLU gate)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b, lazy = _same_mem(_req(a), _req(b))
        out = torch.empty_like(a)
        check(_lib_().nsc_mul(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _st()), "mul")
        ctx.save_for_backward(a, b)
        ctx.lazy = lazy
        return out.transpose(1, 2) if lazy else out

    @staticmethod
    def backward(ctx, dg):
        a, b = ctx.saved_tensors
        dg = to_bct(_req(dg)) if ctx.lazy else _req(dg).contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        lib = _lib_()
        check(lib.nsc_mul(dg.data_ptr(), b.data_ptr(), da.data_ptr(), a.numel(), _st()), "mul")
        check(lib.nsc_mul(dg.data_ptr(), a.data_ptr(), db.data_ptr(), a.numel(), _st()), "mul")
        return (da.transpose(1, 2), db.transpose(1, 2)) if ctx.lazy else (da, db)


class AddFn(torch.autograd.Function):
    """Residual add with channel broadcast ([B,T,C] + [B,T,1])."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _req(a), _req(b)
        ctx.bshape = tuple(b.shape)
        lib = _lib_()
        if a.shape == b.shape:
            am, bm, lazy = _same_mem(a, b)
            out = torch.empty_like(am)
            check(lib.nsc_axpby(am.data_ptr(), bm.data_ptr(), out.data_ptr(), 1.0, 1.0, am.numel(), _st()), "add")
            if lazy:
                out = out.transpose(1, 2)
        else:  # broadcast over channels: run the add in [B,C,T] through a 1-tap identity-free path
            assert b.shape[-1] == 1 and a.shape[:2] == b.shape[:2]
            B, T, Cc = a.shape
            b = b.contiguous()
            ab = to_bct(a)
            ob = torch.empty_like(ab)
            # one launch: an identity 1x1 conv whose epilogue adds the broadcast residual (res_mode 2)
            ident = torch.zeros((1, Cc, Cc), dtype=torch.float32, device=a.device)
            ident[0].fill_diagonal_(1.0)
            d = _desc(B, Cc, Cc, T, T, 1, 1, 1, 0, res_mode=2)
            check(lib.nsc_conv1d_fwd(C.byref(d), ab.data_ptr(), ident.data_ptr(), None, b.reshape(B, 1, T).data_ptr(),
                                     None, ob.data_ptr(), _st()), "broadcast add")
            out = to_btc(ob)
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _req(dy)
        if tuple(dy.shape) == ctx.bshape:
            return dy, dy
        B, T, Cc = dy.shape
        db = torch.empty((B, 1, T), dtype=torch.float32, device=dy.device)
        check(_lib_().nsc_channel_sum(to_bct(dy).data_ptr(), db.data_ptr(), B, Cc, T, 0, _st()), "channel_sum")
        return dy, db.reshape(B, T, 1)


class ShuffleFn(torch.autograd.Function):
    """Sub-pixel shuffle (nsc_module:158-167): out[b, 2t+j, c] = in[b, t, 2c+j] - in [B,C,T] memory: out[b, c, 2t+j] = in[b, 2c+j, t]
    (nsc_shuffle2 / nsc_unshuffle2)."""

    @staticmethod
    def forward(ctx, x):
        xb = to_bct(_req(x))
        B, Cc, T = xb.shape
        y = torch.empty((B, Cc // 2, 2 * T), dtype=torch.float32, device=x.device)
        check(_lib_().nsc_shuffle2(xb.data_ptr(), y.data_ptr(), B, Cc, T, _st()), "shuffle2")
        return to_btc(y)

    @staticmethod
    def backward(ctx, dy):
        dyb = to_bct(_req(dy))
        B, C2, T2 = dyb.shape
        dx = torch.empty((B, 2 * C2, T2 // 2), dtype=torch.float32, device=dy.device)
        check(_lib_().nsc_unshuffle2(dyb.data_ptr(), dx.data_ptr(), B, 2 * C2, T2 // 2, _st()), "unshuffle2")
        return to_btc(dx)


class BlockFn(torch.autograd.Function):
    """gated_bottleneck (nn_core_operator.py:82-112) as ONE function call, like the reference's: the fused persistent kernels
    nsc_gated_block_fwd[_cin1] / nsc_gated_block_dgrad[_cin1] / nsc_gated_block_wgrad instead of four convs + multiply + add +
    activations.  narrow 20, k9 9, dilation 1 | 2, wide <= 112 (one input channel: wide in {100, 50, 25}); other shapes keep the
    composed form (nn_core_operator.gated_bottleneck decides)."""

    @staticmethod
    def forward(ctx, x, w1, b1, wl, bl, wr, br, w9, b9, dil, flat):
        lib = _lib_()
        xb = to_bct(_req(x, "the_input"))
        ws = [_req(t).contiguous() for t in (w1, b1, wl, bl, wr, br, w9, b9)]
        B, Cin, T = xb.shape
        C_ = ws[6].shape[2]
        out = torch.empty((B, C_, T), dtype=torch.float32, device=x.device)
        h, lin, th, g = torch.empty((4, B, 20, T), dtype=torch.float32, device=x.device).unbind(0)
        fn = lib.nsc_gated_block_fwd_cin1 if Cin == 1 else lib.nsc_gated_block_fwd
        check(fn(xb.data_ptr(), *[t.data_ptr() for t in ws], out.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), g.data_ptr(),
                 B, C_, T, 20, 9, int(dil), int(bool(flat)), _st()), "gated_block_fwd")
        ctx.save_for_backward(xb, h, lin, th, g, out, *ws)
        ctx.cfg = (int(dil), bool(flat))
        return to_btc(out)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_()
        xb, h, lin, th, g, out, w1, b1, wl, bl, wr, br, w9, b9 = ctx.saved_tensors
        dil, flat = ctx.cfg
        B, Cin, T = xb.shape
        C_ = w9.shape[2]
        dev = xb.device
        dz = to_bct(_req(dy, "grad"))
        if not flat:                                   # through the block's output leaky-relu
            dz2 = torch.empty_like(dz)
            check(lib.nsc_act_bwd(dz.data_ptr(), out.data_ptr(), dz2.data_ptr(), dz.numel(), ACT["lrelu"], _st()), "act_bwd")
            dz = dz2
        # flipped / transposed kernels of the four convs (what tf.gradients' conv backprop reads): one launch, one buffer
        n1, n15, n9 = Cin * 20, 15 * 20 * 20, 9 * 20 * C_
        wt = torch.empty(n1 + 2 * n15 + n9, dtype=torch.float32, device=dev)
        check(lib.nsc_gated_block_flip_weights(w1.data_ptr(), wl.data_ptr(), wr.data_ptr(), w9.data_ptr(), wt.data_ptr(), C_, Cin, 20, 9,
                                               _st()), "flip")
        p0 = wt.data_ptr()
        wts = [p0, p0 + 4 * n1, p0 + 4 * (n1 + n15), p0 + 4 * (n1 + 2 * n15)]
        dx = torch.empty((B, Cin, T), dtype=torch.float32, device=dev)
        da = torch.empty((B, 40, T), dtype=torch.float32, device=dev)
        dz1 = torch.empty((B, 20, T), dtype=torch.float32, device=dev)
        if Cin == 1:
            check(lib.nsc_gated_block_dgrad_cin1(h.data_ptr(), lin.data_ptr(), th.data_ptr(), dz.data_ptr(), *wts,
                                                 dx.data_ptr(), da.data_ptr(), da.data_ptr() + 4 * 20 * T, dz1.data_ptr(), B, C_, T, 20, 9,
                                                 dil, 40, _st()), "gated_block_dgrad_cin1")
        else:
            check(lib.nsc_gated_block_dgrad(xb.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), dz.data_ptr(),
                                            *wts, dx.data_ptr(), da.data_ptr(), dz1.data_ptr(), B, C_, T, 20, 9,
                                            dil, ACT[None], _st()), "gated_block_dgrad")
        # parameter gradients: the batched launch with one job (it serves both block forms); the eight gradients are one contiguous
        # range in creation order, handed back as views of it
        sizes = [Cin * 20, 20, n15, 20, n15, 20, n9, C_]
        flat_g = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        job = _lib.BlockWgradJob(xb.data_ptr(), h.data_ptr(), g.data_ptr(), dz.data_ptr(), da.data_ptr(), dz1.data_ptr(),
                                 flat_g.data_ptr(), C_, T, dil, Cin)
        ws_ = _workspace(int(lib.nsc_gated_block_wgrad_batch_workspace(112)), dev)
        fn = lib.nsc_gated_block_wgrad_batch_split if SPLIT_ARITH else lib.nsc_gated_block_wgrad_batch
        check(fn((_lib.BlockWgradJob * 1)(job), 1, B, 20, 9, ws_.data_ptr(), ws_.numel(), _st()), "gated_block_wgrad_batch")
        grads = [gr.view(t.shape) for gr, t in zip(flat_g.split(sizes), (w1, b1, wl, bl, wr, br, w9, b9))]
        return (to_btc(dx), *grads, None, None)


class QuantizeFn(torch.autograd.Function):
    """scalar_softmax_quantization: returns (soft p [B,L,nb], bit_code [B,L,1])."""

    @staticmethod
    def forward(ctx, code, alpha, bins, is_quan_on, soft):
        code, bins = _req(code, "floating_code").contiguous(), _req(bins, "bins").contiguous()
        alpha = _req(alpha.reshape(1), "alpha").contiguous()
        B, L, _ = code.shape
        nb = bins.numel()
        p = torch.empty((B, L, nb), dtype=torch.float32, device=code.device)
        out = torch.empty_like(code)
        check(_lib_().nsc_quantize_fwd(code.data_ptr(), alpha.data_ptr(), bins.data_ptr(), float(is_quan_on), int(bool(soft)),
                                       B, L, nb, p.data_ptr(), out.data_ptr(), None, None, _st()), "quantize")
        ctx.save_for_backward(code, alpha, bins)
        ctx.cfg = (float(is_quan_on), int(bool(soft)))
        return p, out

    @staticmethod
    def backward(ctx, dp, dout):
        code, alpha, bins = ctx.saved_tensors
        on, soft = ctx.cfg
        B, L, _ = code.shape
        nb = bins.numel()
        dcode = torch.empty_like(code)
        dalpha = torch.zeros(1, dtype=torch.float32, device=code.device)
        dbins = torch.zeros_like(bins)
        dp = _req(dp).contiguous() if dp is not None else None
        dout = _req(dout).contiguous() if dout is not None else None
        check(_lib_().nsc_quantize_bwd(code.data_ptr(), alpha.data_ptr(), bins.data_ptr(), on, soft, B, L, nb,
                                       _lib.ptr(dout), _lib.ptr(dp), 0.0, None, 0.0, 0, dcode.data_ptr(), dalpha.data_ptr(),
                                       dbins.data_ptr(), _st()), "quantize_bwd")
        return dcode, dalpha.reshape(()), dbins, None, None


class ReconLossFn(torch.autograd.Function):
    """(mse_loss, mfcc_loss) of decoded vs original, both [B,512] -> two [B] vectors; gradient wrt decoded only."""

    @staticmethod
    def forward(ctx, decoded, original):
        from .loss_terms_and_measures import mel_matrix_cat
        decoded, original = _req(decoded).contiguous(), _req(original).contiguous()
        B = decoded.shape[0]
        dev = decoded.device
        key = ("mel", str(dev))
        if key not in _CACHE:
            import numpy as np
            m = mel_matrix_cat()
            _CACHE[key] = (torch.from_numpy(m).to(dev), torch.from_numpy(np.ascontiguousarray(m.T)).to(dev))
        mel, melT = _CACHE[key]
        t, f = torch.empty(B, device=dev), torch.empty(B, device=dev)
        check(_lib_().nsc_recon_loss(decoded.data_ptr(), original.data_ptr(), B, 0.0, 0.0, None, None, mel.data_ptr(),
                                     melT.data_ptr(), t.data_ptr(), f.data_ptr(), None, _st()), "recon_loss")
        ctx.save_for_backward(decoded, original)
        return t, f

    @staticmethod
    def backward(ctx, gt, gf):
        decoded, original = ctx.saved_tensors
        B = decoded.shape[0]
        dev = decoded.device
        mel, melT = _CACHE[("mel", str(dev))]
        gt = _req(gt).contiguous() if gt is not None else torch.zeros(B, device=dev)
        gf = _req(gf).contiguous() if gf is not None else torch.zeros(B, device=dev)
        g = torch.empty_like(decoded)
        t, f = torch.empty(B, device=dev), torch.empty(B, device=dev)
        check(_lib_().nsc_recon_loss(decoded.data_ptr(), original.data_ptr(), B, 0.0, 0.0, gt.data_ptr(), gf.data_ptr(),
                                     mel.data_ptr(), melT.data_ptr(), t.data_ptr(), f.data_ptr(), g.data_ptr(), _st()),
              "recon_loss bwd")
        return g, None


_CACHE = {}


class QuanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        p = _req(p).contiguous()
        B, L, nb = p.shape
        q = torch.empty(B, device=p.device)
        check(_lib_().nsc_p_stats(p.data_ptr(), B, L, nb, q.data_ptr(), None, _st()), "p_stats")
        ctx.save_for_backward(p)
        return q

    @staticmethod
    def backward(ctx, gq):
        (p,) = ctx.saved_tensors
        B, L, nb = p.shape
        dp = torch.empty_like(p)
        check(_lib_().nsc_p_stats_bwd(p.data_ptr(), _req(gq).contiguous().data_ptr(), None, dp.data_ptr(), B, L, nb, _st()), "p_stats_bwd")
        return dp


class EntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        p = _req(p).contiguous()
        B, L, nb = p.shape
        hist = torch.zeros(nb, device=p.device)
        ent = torch.empty(1, device=p.device)
        gh = torch.empty(nb, device=p.device)
        lib = _lib_()
        check(lib.nsc_p_stats(p.data_ptr(), B, L, nb, None, hist.data_ptr(), _st()), "p_stats")
        check(lib.nsc_entropy_from_hist(hist.data_ptr(), nb, ent.data_ptr(), gh.data_ptr(), _st()), "entropy")
        ctx.save_for_backward(p, gh)
        return ent.reshape(())

    @staticmethod
    def backward(ctx, g):
        p, gh = ctx.saved_tensors
        B, L, nb = p.shape
        lib = _lib_()
        ghs = torch.empty_like(gh)
        check(lib.nsc_mul(gh.data_ptr(), _req(g.reshape(1)).expand(nb).contiguous().data_ptr(), ghs.data_ptr(), nb, _st()), "mul")
        dp = torch.empty_like(p)
        check(lib.nsc_p_stats_bwd(p.data_ptr(), None, ghs.data_ptr(), dp.data_ptr(), B, L, nb, _st()), "p_stats_bwd")
        return dp


# ---- functional wrappers used by nn_core_operator / loss_terms_and_measures ----
def recon_losses(decoded, original):
    return ReconLossFn.apply(decoded.reshape(-1, 512), original.reshape(-1, 512))


def rfft512(sig):
    sig = _req(sig).contiguous().reshape(-1, 512)
    B = sig.shape[0]
    re, im, mag = (torch.empty((B, 257), device=sig.device) for _ in range(3))
    check(_lib_().nsc_rfft512(sig.data_ptr(), B, re.data_ptr(), im.data_ptr(), mag.data_ptr(), _st()), "rfft512")
    return torch.complex(re, im), mag


def quan_loss(p):
    return QuanLossFn.apply(p)


def entropy_coding_loss(p):
    return EntropyFn.apply(p)
