"""Explicit forward/backward engine of the NSC/CMRL codec cascade on MI355X.

Host logic only: this file sequences calls into libnsc_hip.so (C ABI, include/nsc_hip.h) on caller-owned,
pre-allocated device buffers (time-contiguous ``[B, C, T]``).  PyTorch is used for device memory, streams and
``torch.distributed`` - no torch arithmetic, no autograd, no CPU fallback.  Every launch is on the current
stream and allocation-free after the first step, so a whole train step can be captured in a hipGraph.

Reference structure being replaced (file:line in cocosci/NSC):
  neural_speech_coding_module.py:152-295  encoder / decoder / codec graph builders
  nn_core_operator.py:82-112, 140-164     gated bottleneck, soft-to-hard quantizer
  loss_terms_and_measures.py:77-183, 257-267  losses
  cmrl.py:22-135, 295-511                 cascade wiring, follower / finetune phases
  neural_speech_coding_module.py:908-926  loss assembly + two TF1 Adam optimizers
"""
from __future__ import annotations

import ctypes as C
import os
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_TANH, ConvDesc, check
from .constants import (beta_boundary, frame_length, init_alpha, lpc_coeff_lsf_bins)
from .loss_terms_and_measures import mel_band_ranges, mel_matrix_cat

KIND_MUL = {"none": 0, "lrelu": 1, "tanh": 2}
KIND_ACT = {"none": ACT_NONE, "lrelu": ACT_LRELU, "tanh": ACT_TANH}


def same_pad(T, k, dil=1, stride=1):
    """TF 'SAME' (SURVEY 8a a1): returns (T_out, padL)."""
    t_out = -(-T // stride)
    pad = max((t_out - 1) * stride + (k - 1) * dil + 1 - T, 0)
    return t_out, pad // 2


_POISON_BUFS = os.environ.get("NSC_POISON_BUFS", "") not in ("", "0")

class ParamLayout:
    """Flat fp32 parameter buffer layout; creation order == TF trainable_variables order inside a scope."""

    def __init__(self):
        self.entries = OrderedDict()  # name -> (offset, shape)
        self.size = 0
        self._counts = {}

    def uniq(self, scope, base):
        n = self._counts.get((scope, base), 0)
        self._counts[(scope, base)] = n + 1
        return f"{scope}/{base}" if n == 0 else f"{scope}/{base}_{n}"

    def add(self, name, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        off = self.size
        self.entries[name] = (off, tuple(shape))
        self.size += n
        return off

    def scope_range(self, scope):
        offs = [(o, o + (int(np.prod(s)) if len(s) else 1)) for k, (o, s) in self.entries.items()
                if k.startswith(scope + "/")]
        return min(a for a, _ in offs), max(b for _, b in offs)


class _Conv:
    """One conv layer: offsets into the flat buffers + fwd / dgrad / wgrad launchers."""

    def __init__(self, eng, scope, K, Cin, Cout, dil=1, stride=1, T_in=frame_length, pointwise_of=None):
        self.eng, self.K, self.Cin, self.Cout, self.dil, self.stride, self.Tin = eng, K, Cin, Cout, dil, stride, T_in
        self.Tout, self.padL = same_pad(T_in, K, dil, stride)
        lay = eng.layout
        if pointwise_of is None:
            name = lay.uniq(scope, "conv1d")
            self.w_off = lay.add(name + "/kernel", (K, Cin, Cout))
            self.b_off = lay.add(name + "/bias", (Cout,))
        else:  # separable conv: depthwise declared by the caller, then pointwise + bias
            self.w_off = lay.add(pointwise_of + "/pointwise_kernel", (1, Cin, Cout))
            self.b_off = lay.add(pointwise_of + "/bias", (Cout,))
        self.name = name if pointwise_of is None else pointwise_of
        eng.convs.append(self)

    # ---- pointer helpers ----
    def _p(self, base, off):
        return base + 4 * off

    def desc(self, **kw):
        d = ConvDesc(B=self.eng.B, Cin=self.Cin, Cout=self.Cout, Tin=self.Tin, Tout=self.Tout, K=self.K, dil=self.dil,
                     stride=self.stride, padL=self.padL, act=0, res_mode=0, mul_mode=0, out_mode=0, in_up=0,
                     accumulate=0)
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    def flops(self):
        """Algorithmic FLOPs of one pass over this layer (2*MAC), identical for fwd, dgrad and wgrad."""
        return 2.0 * self.eng.B * self.Tout * self.Cout * self.K * self.Cin

    def fwd(self, x, y, act="none", res=None, res_mode=0, out_mode=0, chain=None):
        """chain (Cout = 1 only): a _lib.Cout1Chain applied elementwise to the [B,1,T] result in the conv's own epilogue."""
        e = self.eng
        d = self.desc(act=KIND_ACT[act], res_mode=res_mode, out_mode=out_mode)
        split = (e.split_conv and getattr(self, "simg_off", None) is not None and res is None and out_mode == 0 and e.use_images and
                 e.images_valid)
        tok = e.prof_begin("conv_cout1" if self.Cout == 1 else ("conv_split" if split else "conv_mfma"), self.flops())
        if self.Cout == 1:
            check(e.lib.nsc_conv1d_cout1_fwd_chain(C.byref(d), x.data_ptr(), self._p(e.p_ptr, self.w_off), self._p(e.p_ptr, self.b_off),
                                                   _lib.ptr(res), None, y.data_ptr(), C.byref(chain) if chain is not None else None,
                                                   e.stream()), f"conv fwd {self.name}")
        elif split:
            # the stride-2 down-sampling conv on split operands (bf16 matrix cores, csrc/conv_split.hip)
            assert chain is None
            check(e.lib.nsc_conv1d_fwd_simg(C.byref(d), x.data_ptr(), e.wtp(self.simg_off[0]), self._p(e.p_ptr, self.b_off),
                                            y.data_ptr(), e.stream()), f"conv fwd (split) {self.name}")
        else:
            assert chain is None
            check(e.lib.nsc_conv1d_fwd(C.byref(d), x.data_ptr(), self._p(e.p_ptr, self.w_off), self._p(e.p_ptr, self.b_off),
                                       _lib.ptr(res), None, y.data_ptr(), e.stream()), f"conv fwd {self.name}")
        e.prof_end(tok)

    def poly_ok(self):
        """stride-2 k9 SAME conv (padL 3): its data gradient splits into two phases with 4 / 5 taps (see wtpoly_index)."""
        return self.stride == 2 and self.K == 9 and self.dil == 1 and self.padL == 3 and 2 * self.Cin <= 224 and self.Cin > 1

    def wtpoly_index(self):
        """Polyphase form of the stride-2 data gradient.  Forward: y[o,m] = sum_k W[k,ci,o] x[ci, 2m+k-3], so
        dx[ci, 2n+p] = sum_{t'} W[7-2t'+p, ci, o] dy[o, n+t'-2]: a stride-1 conv over dy with 5 taps (4 for p = 0) whose
        output channels 2ci+p interleave in time (sub-pixel shuffle).  The zero-upsampled form spends half of its MFMAs on
        zeros.  Returns source offsets for W'[t', o, 2ci+p] (-1 = structural zero)."""
        K, Ci, Co = self.K, self.Cin, self.Cout
        src = np.arange(K * Ci * Co, dtype=np.int64).reshape(K, Ci, Co) + self.w_off
        out = np.full((5, Co, 2 * Ci), -1, dtype=np.int64)
        for tp in range(5):
            for par in range(2):
                k = 7 - 2 * tp + par
                if 0 <= k < K:
                    out[tp, :, par::2] = src[k].T          # [Co, Ci]
        return out.reshape(-1).astype(np.int32)

    def dgrad(self, dz, dx, res=None, res_mode=0, mul_kind="none", aux=None, chain=None):
        """dx = conv^T(dz) (+res) (* act'(aux)); runs the forward kernel on the flipped/transposed weights.
        chain (Cin = 1 only): a _lib.Cout1Chain applied to the [B,1,T] result in the kernel's epilogue."""
        e = self.eng
        if e.split_conv and getattr(self, "simg_off", None) is not None and res is None and mul_kind == "none":
            tok = e.prof_begin("conv_split", self.flops())
            check(e.lib.nsc_conv1d_dgrad_simg(C.byref(self.desc()), dz.data_ptr(), e.wtp(self.simg_off[1]), dx.data_ptr(), e.stream()),
                  f"conv dgrad (split) {self.name}")
            e.prof_end(tok)
            return
        if self.poly_ok() and e.poly_dgrad:
            d = ConvDesc(B=e.B, Cin=self.Cout, Cout=2 * self.Cin, Tin=self.Tout, Tout=self.Tout, K=5, dil=1, stride=1, padL=2,
                         act=0, res_mode=res_mode, mul_mode=KIND_MUL[mul_kind], out_mode=1, in_up=0, accumulate=0)
            tok = e.prof_begin("conv_mfma", self.flops())
            check(e.lib.nsc_conv1d_fwd(C.byref(d), dz.data_ptr(), e.wtp(self.wtpoly_off), None, _lib.ptr(res),
                                       _lib.ptr(aux) if mul_kind != "none" else None, dx.data_ptr(), e.stream()),
                  f"conv dgrad (polyphase) {self.name}")
            e.prof_end(tok)
            return
        padl = (self.K - 1) * self.dil - self.padL
        d = ConvDesc(B=e.B, Cin=self.Cout, Cout=self.Cin, Tin=self.Tout, Tout=self.Tin, K=self.K, dil=self.dil,
                     stride=1, padL=padl, act=0, res_mode=res_mode, mul_mode=KIND_MUL[mul_kind], out_mode=0,
                     in_up=1 if self.stride == 2 else 0, accumulate=0)
        tok = e.prof_begin("conv_cout1" if self.Cin == 1 else "conv_mfma", self.flops())
        if self.Cin == 1:
            check(e.lib.nsc_conv1d_cout1_fwd_chain(C.byref(d), dz.data_ptr(), e.wtp(self.w_off), None, _lib.ptr(res),
                                                   _lib.ptr(aux) if mul_kind != "none" else None, dx.data_ptr(),
                                                   C.byref(chain) if chain is not None else None, e.stream()), f"conv dgrad {self.name}")
        else:
            assert chain is None
            check(e.lib.nsc_conv1d_fwd(C.byref(d), dz.data_ptr(), e.wtp(self.w_off), None, _lib.ptr(res),
                                       _lib.ptr(aux) if mul_kind != "none" else None, dx.data_ptr(), e.stream()), f"conv dgrad {self.name}")
        e.prof_end(tok)

    def wgrad(self, x, dz):
        e = self.eng
        dw, db = self._p(e.g_ptr, self.w_off), self._p(e.g_ptr, self.b_off)
        if e.batch_conv_wgrad:
            e._cw_flops += self.flops()
            self._wgrad(x, dz, dw, db)
            return
        tok = e.prof_begin("wgrad_mfma", self.flops())
        self._wgrad(x, dz, dw, db)
        e.prof_end(tok)

    def wgrad_desc(self):
        """Descriptor of the weight-gradient launch (roles swapped for Cout == 1: "input" = dz, "grad" = x, flipped taps)."""
        e = self.eng
        if self.Cout == 1:
            assert self.stride == 1
            return ConvDesc(B=e.B, Cin=1, Cout=self.Cin, Tin=self.Tout, Tout=self.Tin, K=self.K, dil=self.dil, stride=1,
                            padL=(self.K - 1) * self.dil - self.padL, act=0, res_mode=0, mul_mode=0, out_mode=0, in_up=0,
                            accumulate=0)
        return self.desc()

    def _wgrad(self, x, dz, dw, db):
        e = self.eng
        if e.batch_conv_wgrad:
            # deferred to the end of the backward pass (x, dz are private to this conv and live until then)
            d = self.wgrad_desc()
            if self.Cout == 1:
                e.defer_conv_wgrad(d, dz, x, dw, None, 1)
                e._sum_jobs.append(_lib.SumJob(dz.data_ptr(), db, dz.numel()))     # bias gradient: one launch for all of them
            else:
                e.defer_conv_wgrad(d, x, dz, dw, db, 0)
            return
        st = e.side_fork()
        ws = e.wgrad_workspace(slot=e.side_idx)
        d = self.wgrad_desc()
        if self.Cout == 1:
            check(e.lib.nsc_conv1d_wgrad_ws(C.byref(d), dz.data_ptr(), x.data_ptr(), dw, None, 1, ws, e._ws_floats, st),
                  f"conv wgrad(swapped) {self.name}")
            check(e.lib.nsc_sum_all(dz.data_ptr(), db, dz.numel(), st), "bias grad")
        else:
            check(e.lib.nsc_conv1d_wgrad_ws(C.byref(d), x.data_ptr(), dz.data_ptr(), dw, db, 0, ws, e._ws_floats, st),
                  f"conv wgrad {self.name}")

    def wt_index(self):
        """Index map e -> source offset so that wt[w_off + ((K-1-k)*Cout + o)*Cin + i] = p[w_off + (k*Cin+i)*Cout + o]."""
        K, Ci, Co = self.K, self.Cin, self.Cout
        src = np.arange(K * Ci * Co, dtype=np.int64).reshape(K, Ci, Co)
        return (self.w_off + src[::-1].transpose(0, 2, 1).reshape(-1)).astype(np.int32)


class _Block:
    """gated_bottleneck (nn_core_operator.py:82-112): 1x1 -> lrelu -> {k15 dil, k15 dil + tanh} -> mul -> k9 -> +x -> lrelu."""

    def __init__(self, eng, scope, uid, Cin, wide, narrow, k9, dil, flat, T):
        self.eng, self.uid, self.Cin, self.wide, self.narrow, self.flat, self.T = eng, uid, Cin, wide, narrow, flat, T
        self.c1 = _Conv(eng, scope, 1, Cin, narrow, 1, 1, T)
        self.cl = _Conv(eng, scope, 15, narrow, narrow, dil, 1, T)   # kernel size 15 hard-coded (:92, :97)
        self.cr = _Conv(eng, scope, 15, narrow, narrow, dil, 1, T)
        self.c9 = _Conv(eng, scope, k9, narrow, wide, 1, 1, T)
        self.out_kind = "none" if flat else "lrelu"

    def fwd(self, x):
        e, u = self.eng, self.uid
        B, n, T = e.B, self.narrow, self.T
        self.x = x
        self.h = e.buf(u + ".h", (B, n, T))
        self.lin = e.buf(u + ".lin", (B, n, T))
        self.th = e.buf(u + ".th", (B, n, T))
        self.g = e.buf(u + ".g", (B, n, T))
        self.out = e.buf(u + ".out", (B, self.wide, T))
        cin1 = self.Cin == 1 and self.wide in (100, 50, 25) and self.cl.dil in (1, 2)   # first block of a decoder stage
        if e.fused_fwd and (self.Cin > 1 or cin1) and n == 20 and self.c9.K == 9 and self.wide <= 112 and self.cl.dil <= 4:
            # (other reference-legal shapes, e.g. wide 128 or dilation 8, take the per-conv path below)
            P = lambda c, base=e.p_ptr: (base + 4 * c.w_off, base + 4 * c.b_off)
            (w1, b1), (wl, bl), (wr, br), (w9, b9) = P(self.c1), P(self.cl), P(self.cr), P(self.c9)
            tok = e.prof_begin("block_fwd", self.c1.flops() + self.cl.flops() + self.cr.flops() + self.c9.flops())
            # the fused backward recomputes the intermediates from x; an inference-only engine never reads them
            save = e.keep_activations
            sv = [t.data_ptr() if save else None for t in (self.h, self.lin, self.th, self.g)]
            if e.use_images and e.images_valid and e.split_fwd and self.simg_fwd_off is not None and T % 4 == 0:
                check(e.lib.nsc_gated_block_fwd_simg(e.wtp(self.simg_fwd_off), x.data_ptr(), self.out.data_ptr(), *sv, B,
                                                     self.wide, self.Cin, T, self.cl.dil, int(self.flat), e.stream()),
                      "gated_block_fwd_simg")
            elif e.use_images and e.images_valid and self.img_fwd_off is not None:
                check(e.lib.nsc_gated_block_fwd_img(e.wtp(self.img_fwd_off), x.data_ptr(), self.out.data_ptr(), *sv, B,
                                                    self.wide, self.Cin, T, self.cl.dil, int(self.flat), e.stream()),
                      "gated_block_fwd_img")
            else:
                fn = e.lib.nsc_gated_block_fwd_cin1 if cin1 else e.lib.nsc_gated_block_fwd
                check(fn(x.data_ptr(), w1, b1, wl, bl, wr, br, w9, b9, self.out.data_ptr(), *sv, B, self.wide, T, n, 9, self.cl.dil,
                         int(self.flat), e.stream()), "gated_block_fwd")
            e.prof_end(tok)
            return self.out
        self.c1.fwd(x, self.h, "lrelu")
        self.cl.fwd(self.h, self.lin, "none")
        self.cr.fwd(self.h, self.th, "tanh")
        check(e.lib.nsc_mul(self.lin.data_ptr(), self.th.data_ptr(), self.g.data_ptr(), self.g.numel(), e.stream()), "mul")
        self.c9.fwd(self.g, self.out, self.out_kind, res=x, res_mode=2 if self.Cin == 1 else 1)
        return self.out

    def bwd(self, dz, in_kind, need_dx=True):
        """dz = dL/d(pre-activation of out).  Returns dL/d(pre-activation of the producer of x)."""
        e, u = self.eng, self.uid
        B, n, T = e.B, self.narrow, self.T
        dg = e.buf(u + ".dg", (B, n, T))
        dlin = e.buf(u + ".dlin", (B, n, T))
        dgate = e.buf(u + ".dgate", (B, n, T))
        dh = e.buf(u + ".dh", (B, n, T))
        block_wgrad = (e.fused_wgrad and self.Cin > 1 and n == 20 and self.c9.K == 9 and self.cl.dil <= 4
                       and self.wide <= 112)
        if block_wgrad:
            # data path: k9 dgrad -> GLU backward (dlin | dgate in one tensor) -> ONE k15 dgrad over 40 input channels;
            # then a persistent kernel does all eight parameter gradients AND the 1x1 data gradient (it has x, dy, dz1
            # staged anyway).
            da = e.buf(u + ".da", (B, 2 * n, T))
            fused_dgrad = e.fused_dgrad and self.cl.dil in (1, 2)
            if fused_dgrad:
                # one 8-wave kernel for the whole data path of the block: dx, da (dlin | dgate) and dz1
                dxf = e.buf(u + ".dx", (B, self.Cin, T))
                WT = lambda c: e.wtp(c.w_off)
                tok = e.prof_begin("block_dgrad", self.c1.flops() + self.cl.flops() + self.cr.flops() + self.c9.flops())
                if e.use_images and e.split_dgrad and self.simg_bwd_off is not None and T % 4 == 0 and self.Cin == self.wide:
                    # three launches on the bf16 matrix cores (csrc/block_bwd_split.hip): k9^T + GLU', k15^T + lrelu', 1x1^T + residual
                    check(e.lib.nsc_gated_block_dgrad_simg2(e.wtp(self.simg_bwd_off), e.p_ptr + 4 * self.c1.w_off,
                                                            self.x.data_ptr(), self.h.data_ptr(), self.lin.data_ptr(),
                                                            self.th.data_ptr(), dz.data_ptr(), dxf.data_ptr(), da.data_ptr(),
                                                            dh.data_ptr(), B, self.wide, self.Cin, T, self.cl.dil, KIND_ACT[in_kind],
                                                            e.stream()), "gated_block_dgrad_simg2")
                elif e.use_images and self.img_bwd_off is not None:     # (the data-gradient images are rebuilt with wt every step)
                    check(e.lib.nsc_gated_block_dgrad_img(e.wtp(self.img_bwd_off), self.x.data_ptr(), self.h.data_ptr(),
                                                          self.lin.data_ptr(), self.th.data_ptr(), dz.data_ptr(), dxf.data_ptr(),
                                                          da.data_ptr(), da.data_ptr() + 4 * n * T, dh.data_ptr(), B, self.wide,
                                                          self.Cin, T, self.cl.dil, KIND_ACT[in_kind], 2 * n, e.stream()),
                          "gated_block_dgrad_img")
                else:
                    check(e.lib.nsc_gated_block_dgrad(self.x.data_ptr(), self.h.data_ptr(), self.lin.data_ptr(),
                                                      self.th.data_ptr(), dz.data_ptr(), WT(self.c1), WT(self.cl), WT(self.cr),
                                                      WT(self.c9), dxf.data_ptr(), da.data_ptr(), dh.data_ptr(), B, self.Cin, T,
                                                      n, 9, self.cl.dil, KIND_ACT[in_kind], e.stream()), "gated_block_dgrad")
                e.prof_end(tok)
            else:
                self.c9.dgrad(dz, dg)
                check(e.lib.nsc_glu_bwd_cat(self.lin.data_ptr(), self.th.data_ptr(), dg.data_ptr(), da.data_ptr(), B, n, T,
                                            e.stream()), "glu_bwd_cat")
                cl = self.cl
                d = ConvDesc(B=B, Cin=2 * n, Cout=n, Tin=T, Tout=T, K=cl.K, dil=cl.dil, stride=1,
                             padL=(cl.K - 1) * cl.dil - cl.padL, act=0, res_mode=0, mul_mode=KIND_MUL["lrelu"], out_mode=0,
                             in_up=0, accumulate=0)
                tok = e.prof_begin("conv_mfma", self.cl.flops() + self.cr.flops())
                check(e.lib.nsc_conv1d_fwd(C.byref(d), da.data_ptr(), e.wtp(self.wtlr_off), None, None,
                                           self.h.data_ptr(), dh.data_ptr(), e.stream()), "gate dgrad (fused lin|gate)")
                e.prof_end(tok)
            dx = e.buf(u + ".dx", (B, self.Cin, T)) if need_dx else None
            G = lambda c: (e.g_ptr + 4 * c.w_off, e.g_ptr + 4 * c.b_off)
            (dw1, db1), (dwl, dbl), (dwr, dbr), (dw9, db9) = G(self.c1), G(self.cl), G(self.cr), G(self.c9)
            fuse_d1 = need_dx and not e.overlap_wgrad and not fused_dgrad   # with overlap the 1x1 dgrad stays on the critical stream
            fl = self.c1.flops() * (2 if fuse_d1 else 1) + self.cl.flops() + self.cr.flops() + self.c9.flops()
            if e.batch_wgrad and fused_dgrad:
                # defer: every buffer the job reads is private to this block and lives until the end of the step
                e.defer_block_wgrad(self, dz, da, dh, dw1, fl)
                return dxf if need_dx else None
            tok = e.prof_begin("block_wgrad", fl)
            st = e.side_fork()
            args = (self.x.data_ptr(), self.h.data_ptr(), self.g.data_ptr(), dz.data_ptr(), da.data_ptr(), dh.data_ptr(),
                    dw1, db1, dwl, dbl, dwr, dbr, dw9, db9, e.wtp(self.c1.w_off))
            tail = (KIND_ACT[in_kind], B, self.Cin, T, n, 9, self.cl.dil)
            if e.overlap_wgrad and e.split_wgrad:
                # two light persistent launches (dW9 | dWl, dWr, dW1) that can share CUs with the main-stream kernels
                for part in (1, 2):
                    if part == 2:
                        st = e.side_fork()     # the two parts are independent: next side stream
                    check(e.lib.nsc_gated_block_wgrad(*args, None, *tail, 4, part, e.wgrad_workspace(slot=e.side_idx), st),
                          "gated_block_wgrad")
            else:
                check(e.lib.nsc_gated_block_wgrad(*args, _lib.ptr(dx) if fuse_d1 else None, *tail, e.wgrad_waves, 0,
                                                  e.wgrad_workspace(slot=e.side_idx), st), "gated_block_wgrad")
            e.prof_end(tok)
            if fused_dgrad:
                return dxf if need_dx else None
            if need_dx and not fuse_d1:
                self.c1.dgrad(dh, dx, res=dz, res_mode=1, mul_kind=in_kind, aux=self.x)
            return dx
        if (self.Cin == 1 and e.fused_fwd and e.fused_dgrad and n == 20 and self.c9.K == 9 and self.wide in (100, 50, 25)
                and self.cl.dil in (1, 2)):
            # first block of a decoder stage: the whole data path in one persistent kernel (one input channel: the 1x1
            # gradient is a dot product, the residual branch sums dy over its channels); its weight gradients join the
            # step's batched block launch (job.Cin = 1), which reads dlin | dgate as one [B,40,T] tensor
            assert in_kind == "none"
            dx = e.buf(u + ".dx", (B, 1, T))
            batched = e.batch_wgrad and e.batch_cin1_wgrad
            if batched:
                da = e.buf(u + ".da", (B, 2 * n, T))
                dlin, dgate = da[:, :n], da[:, n:]      # data_ptr of the halves: dgate = dlin + 20 T floats
            WT = lambda c: e.wtp(c.w_off)
            tok = e.prof_begin("block_dgrad", self.c1.flops() + self.cl.flops() + self.cr.flops() + self.c9.flops())
            if e.use_images and e.split_dgrad and self.simg_bwd_off is not None and T % 4 == 0 and batched:
                check(e.lib.nsc_gated_block_dgrad_simg2(e.wtp(self.simg_bwd_off), e.p_ptr + 4 * self.c1.w_off, None,
                                                        self.h.data_ptr(), self.lin.data_ptr(), self.th.data_ptr(), dz.data_ptr(),
                                                        dx.data_ptr(), da.data_ptr(), dh.data_ptr(), B, self.wide, 1, T, self.cl.dil,
                                                        KIND_ACT["none"], e.stream()), "gated_block_dgrad_simg2 (one input channel)")
            elif e.use_images and self.img_bwd_off is not None:
                check(e.lib.nsc_gated_block_dgrad_img(e.wtp(self.img_bwd_off), None, self.h.data_ptr(), self.lin.data_ptr(),
                                                      self.th.data_ptr(), dz.data_ptr(), dx.data_ptr(), dlin.data_ptr(),
                                                      dgate.data_ptr(), dh.data_ptr(), B, self.wide, 1, T, self.cl.dil,
                                                      KIND_ACT["none"], 2 * n if batched else n, e.stream()), "gated_block_dgrad_img")
            else:
                check(e.lib.nsc_gated_block_dgrad_cin1(self.h.data_ptr(), self.lin.data_ptr(), self.th.data_ptr(), dz.data_ptr(),
                                                       WT(self.c1), WT(self.cl), WT(self.cr), WT(self.c9), dx.data_ptr(),
                                                       dlin.data_ptr(), dgate.data_ptr(), dh.data_ptr(), B, self.wide, T, n, 9,
                                                       self.cl.dil, 2 * n if batched else n, e.stream()), "gated_block_dgrad_cin1")
            e.prof_end(tok)
            if batched:
                e.defer_block_wgrad(self, dz, da, dh, e.g_ptr + 4 * self.c1.w_off,
                                    self.c1.flops() + self.cl.flops() + self.cr.flops() + self.c9.flops())
                return dx if need_dx else None
            self.c9.wgrad(self.g, dz)
            self.cl.wgrad(self.h, dlin)
            self.cr.wgrad(self.h, dgate)
            self.c1.wgrad(self.x, dh)
            return dx if need_dx else None
        self.c9.wgrad(self.g, dz)
        self.c9.dgrad(dz, dg)
        check(e.lib.nsc_glu_bwd(self.lin.data_ptr(), self.th.data_ptr(), dg.data_ptr(), dlin.data_ptr(),
                                dgate.data_ptr(), dg.numel(), e.stream()), "glu_bwd")
        self.cl.wgrad(self.h, dlin)
        self.cr.wgrad(self.h, dgate)
        dh0 = e.buf(u + ".dh0", (B, n, T))
        self.cl.dgrad(dlin, dh0)
        self.cr.dgrad(dgate, dh, res=dh0, res_mode=1, mul_kind="lrelu", aux=self.h)
        self.c1.wgrad(self.x, dh)
        if not need_dx:
            return None
        dx = e.buf(u + ".dx", (B, self.Cin, T))
        if self.Cin == 1:
            assert in_kind == "none"
            self.c1.dgrad(dh, dx)
            check(e.lib.nsc_channel_sum(dz.data_ptr(), dx.data_ptr(), B, self.wide, T, 1, e.stream()), "channel_sum")
        else:
            self.c1.dgrad(dh, dx, res=dz, res_mode=1, mul_kind=in_kind, aux=self.x)
        return dx


class _Codec:
    """One neural codec (neural_speech_coding_module.py:262-295): encoder -> quantizer -> decoder."""

    def __init__(self, eng, scope, bkd, strides, nb):
        self.eng, self.scope, self.strides, self.nb = eng, scope, list(strides), nb
        self.index = int(scope.rsplit("_", 1)[1]) - 1          # position in the cascade (scope_1 -> 0)
        lay = eng.layout
        self.alpha_off = lay.add(scope + "/alpha", ())
        self.bins_off = lay.add(scope + "/bins", (nb,))
        wide, narrow, k9 = bkd[2], bkd[3], bkd[1]
        dils = bkd[4:]
        T = frame_length
        uid = [0]

        def stack(Cin, T):
            blocks = []
            w = wide if Cin == 1 else Cin
            c = Cin
            for i, dl in enumerate(dils):
                uid[0] += 1
                blocks.append(_Block(eng, scope, f"{scope}.b{uid[0]}", c, w, narrow, k9, dl, i == len(dils) - 1, T))
                c = w
            return blocks, w

        # ---- encoder (:219-237) ----
        self.in_conv = _Conv(eng, scope, 55, 1, wide, 1, 1, T)
        self.enc_stages = []  # (blocks, down_conv)
        C_ = wide
        for s in self.strides:
            blocks, C_ = stack(C_, T)
            down = _Conv(eng, scope, 9, C_, wide, 1, s, T)
            T = down.Tout
            C_ = wide
            self.enc_stages.append((blocks, down))
        self.enc_tail, C_ = stack(C_, T)
        self.enc_out = _Conv(eng, scope, 55, C_, 1, 1, 1, T)
        self.L = T
        # ---- decoder (:239-260) ----
        self.dec_stages = []  # (blocks, depthwise_off, pointwise conv, T_before, C)
        C_ = 1
        for s in self.strides:
            assert s == 2, "sub-pixel up-sampling is implemented for stride 2"
            blocks, C_ = stack(C_, T)
            name = lay.uniq(scope, "separable_conv1d")
            dw_off = lay.add(name + "/depthwise_kernel", (9, C_, 1))
            pw = _Conv(eng, scope, 1, C_, C_, 1, 1, T, pointwise_of=name)
            self.dec_stages.append((blocks, dw_off, pw, T, C_))
            T, C_ = T * s, C_ // s
        self.dec_tail, C_ = stack(C_, T)
        self.dec_out = _Conv(eng, scope, 55, C_, 1, 1, 1, T)

    # ---- a stack of blocks (one resolution): the dil-1 / dil-2 pair of C -> C blocks runs as ONE launch when it can ----
    def _pair_ok(self, blocks):
        e = self.eng
        if not (e.fused_pairs and e.fused_fwd and e.fused_dgrad and e.use_images and len(blocks) == 2):
            return False
        # A pair launch needs every one of its workgroups resident at once.  With per-scope gradient messages under the backward
        # pass (dp_overlap) a collective's kernel can hold compute units while a 256-workgroup pair launch waits for them: its
        # neighbour waits would spin into the time-out.  The default tail message overlaps nothing and keeps the pairs.
        if e.dp_overlap and e._dp_comm_attached:
            return False
        b0, b1 = blocks
        return ((b0.Cin == b0.wide or (b0.Cin == 1 and e.batch_cin1_wgrad)) and b0.wide == b1.Cin == b1.wide and
                b0.wide in (100, 50, 25) and b0.cl.dil == 1 and b1.cl.dil == 2 and
                b0.narrow == 20 and b0.c9.K == 9 and b0.img_fwd_off is not None and b1.img_fwd_off is not None and
                b0.img_bwd_off is not None and b1.img_bwd_off is not None and not b0.flat and b0.T % 4 == 0 and
                e.B * b0.wide * b0.T * 4 < 2 ** 31 and
                min(e.B * ((b0.T + 63) // 64), 256) <= e.cu_count)

    def stack_fwd(self, blocks, h):
        e = self.eng
        if not (self._pair_ok(blocks) and e.images_valid):
            for blk in blocks:
                h = blk.fwd(h)
            return h
        b0, b1 = blocks
        B, n, T = e.B, b0.narrow, b0.T
        for blk, xin in ((b0, h), (b1, None)):
            u = blk.uid
            blk.x = xin
            blk.h = e.buf(u + ".h", (B, n, T)); blk.lin = e.buf(u + ".lin", (B, n, T))
            blk.th = e.buf(u + ".th", (B, n, T)); blk.g = e.buf(u + ".g", (B, n, T))
            blk.out = e.buf(u + ".out", (B, blk.wide, T))
        b1.x = b0.out
        save = e.keep_activations
        sv = lambda blk: [t.data_ptr() if save else None for t in (blk.h, blk.lin, blk.th, blk.g)]
        fl = sum(c.flops() for blk in blocks for c in (blk.c1, blk.cl, blk.cr, blk.c9))
        tok = e.prof_begin("block_fwd", fl)
        if e.split_fwd and b0.simg_fwd_off is not None and b1.simg_fwd_off is not None:
            check(e.lib.nsc_gated_block_pair_fwd_simg(e.wtp(b0.simg_fwd_off), e.wtp(b1.simg_fwd_off), h.data_ptr(),
                                                      b0.out.data_ptr(), *sv(b0), b1.out.data_ptr(), *sv(b1), B, b0.wide, b0.Cin, T,
                                                      int(b1.flat), e.pair_flags(), e.pair_timeouts_ptr(), e.stream()),
                  "gated_block_pair_fwd_simg")
        else:
            check(e.lib.nsc_gated_block_pair_fwd_img(e.wtp(b0.img_fwd_off), e.wtp(b1.img_fwd_off), h.data_ptr(),
                                                     b0.out.data_ptr(), *sv(b0), b1.out.data_ptr(), *sv(b1), B, b0.wide, b0.Cin, T,
                                                     int(b1.flat), e.pair_flags(), e.pair_timeouts_ptr(), e.stream()), "gated_block_pair_fwd_img")
        e.prof_end(tok)
        return b1.out

    def stack_bwd(self, blocks, dz, in_kind_first, in_kind_rest="lrelu"):
        """Backward through a stack: dz = dL/d(pre-activation of the last block's output).  in_kind_first: what produced the
        first block's input.  Returns dL/d(pre-activation of that producer)."""
        e = self.eng
        # (the split-operand data gradient has no pair form yet: a stack with a block it serves runs block by block)
        split_any = e.split_dgrad and e.use_images and any(b.simg_bwd_off is not None and b.T % 4 == 0 for b in blocks)
        if split_any or not (self._pair_ok(blocks) and e.pair_bwd and e.batch_wgrad and e.fused_wgrad and in_kind_first in ("lrelu", "none") and
                             (blocks[0].Cin > 1 or in_kind_first == "none")):
            for j in range(len(blocks) - 1, -1, -1):
                dz = blocks[j].bwd(dz, in_kind_rest if j > 0 else in_kind_first)
            return dz
        b0, b1 = blocks
        B, n, T = e.B, b0.narrow, b0.T
        bufs = {}
        for blk in blocks:
            u = blk.uid
            bufs[blk] = (e.buf(u + ".dx", (B, blk.Cin, T)), e.buf(u + ".da", (B, 2 * n, T)), e.buf(u + ".dh", (B, n, T)))
        (dx0, da0, dh0), (dx1, da1, dh1) = bufs[b0], bufs[b1]
        fl = sum(c.flops() for blk in blocks for c in (blk.c1, blk.cl, blk.cr, blk.c9))
        tok = e.prof_begin("block_dgrad", fl)
        P = lambda t: t.data_ptr()
        check(e.lib.nsc_gated_block_pair_dgrad_img(e.wtp(b1.img_bwd_off), P(b1.x), P(b1.h), P(b1.lin), P(b1.th), P(dz), P(dx1),
                                                   P(da1), P(dh1), e.wtp(b0.img_bwd_off), P(b0.x), P(b0.h), P(b0.lin),
                                                   P(b0.th), P(dx0), P(da0), P(dh0), B, b0.wide, b0.Cin, T, KIND_ACT[in_kind_first],
                                                   e.pair_flags(), e.pair_timeouts_ptr(), e.stream()), "gated_block_pair_dgrad_img")
        e.prof_end(tok)
        for blk, dzb, da, dh in ((b1, dz, da1, dh1), (b0, dx1, da0, dh0)):
            e.defer_block_wgrad(blk, dzb, da, dh, e.g_ptr + 4 * blk.c1.w_off,
                                blk.c1.flops() + blk.cl.flops() + blk.cr.flops() + blk.c9.flops())
        return dx0

    def all_blocks(self):
        out = []
        for blocks, _ in self.enc_stages:
            out += blocks
        out += self.enc_tail
        for st in self.dec_stages:
            out += st[0]
        return out + self.dec_tail

    # -------------------------------------------------------------------------------------------
    def forward(self, x, is_quan_on, soft, want_p=False, chain=None):
        """chain: elementwise follow-up of the decoded frame in the output conv's epilogue (the cascade step)."""
        e, s = self.eng, self.scope
        B = e.B
        self.x_in = x
        self.h0 = e.buf(s + ".h0", (B, self.in_conv.Cout, frame_length))
        self.in_conv.fwd(x, self.h0, "lrelu")
        h = self.h0
        self.down_out = []
        for i, (blocks, down) in enumerate(self.enc_stages):
            h = self.stack_fwd(blocks, h)
            d = e.buf(f"{s}.down{i}", (B, down.Cout, down.Tout))
            down.fwd(h, d, "lrelu")
            self.down_out.append((h, d))
            h = d
        h = self.stack_fwd(self.enc_tail, h)
        self.enc_feat = h
        self.code = e.buf(s + ".code", (B, 1, self.L))
        # ---- encoder output conv + quantizer + fused quan/entropy partials ----
        self.qcode = e.buf(s + ".qcode", (B, 1, self.L))
        self.hist = e.hist_view(s, self.nb)     # a slice of ONE flat buffer: the data-parallel exchange is a single all-reduce
        self.p = e.buf(s + ".p", (B, self.L, self.nb)) if want_p else None
        e.zero_hists_once()
        self.is_quan_on, self.soft = float(is_quan_on), int(bool(soft))
        eo = self.enc_out
        if (e.fused_quant and not want_p and self.nb == 32 and eo.K == 55 and eo.dil == 1 and eo.stride == 1 and 8 <= eo.Cin <= 104):
            # the quantizer rides in the conv's launch (nsc_conv1d_cout1_fwd_quant); quan_loss accumulates tile by tile into the
            # zeroed accumulator behind the gradients
            self.quan = e._quan_acc[self.index]
            qz = _lib.Cout1Quant(e.p_ptr + 4 * self.alpha_off, e.p_ptr + 4 * self.bins_off, self.is_quan_on, self.soft, self.nb,
                                 self.qcode.data_ptr(), self.quan.data_ptr(), self.hist.data_ptr())
            d = eo.desc(act=KIND_ACT["tanh"], res_mode=0, out_mode=0)
            tok = e.prof_begin("conv_cout1", eo.flops())
            check(e.lib.nsc_conv1d_cout1_fwd_quant(C.byref(d), h.data_ptr(), eo._p(e.p_ptr, eo.w_off), eo._p(e.p_ptr, eo.b_off),
                                                   self.code.data_ptr(), C.byref(qz), e.stream()), "conv1d_cout1_fwd_quant")
            e.prof_end(tok)
        else:
            self.enc_out.fwd(h, self.code, "tanh")
            self.quan = e.buf(s + ".quan", (B,))
            check(e.lib.nsc_quantize_fwd(self.code.data_ptr(), e.p_ptr + 4 * self.alpha_off, e.p_ptr + 4 * self.bins_off,
                                         self.is_quan_on, self.soft, B, self.L, self.nb, _lib.ptr(self.p),
                                         self.qcode.data_ptr(), self.quan.data_ptr(), self.hist.data_ptr(), e.stream()),
                  "quantize_fwd")
        # ---- decoder ----
        h = self.qcode
        self.up_saved = []
        for i, (blocks, dw_off, pw, T, C_) in enumerate(self.dec_stages):
            h = self.stack_fwd(blocks, h)
            dwo = e.buf(f"{s}.dw{i}", (B, C_, T))
            up = e.buf(f"{s}.up{i}", (B, C_ // 2, T * 2))
            if e.fused_up and C_ in (100, 50) and B * C_ * T < 2 ** 29:    # (32-bit buffer offsets in the fused kernels)
                # depthwise -> pointwise -> leaky-relu -> shuffle in one kernel (dwo kept only when a backward pass follows)
                tok = e.prof_begin("upsample", pw.flops())
                check(e.lib.nsc_upsample_fwd(h.data_ptr(), e.p_ptr + 4 * dw_off, e.p_ptr + 4 * pw.w_off, e.p_ptr + 4 * pw.b_off,
                                             dwo.data_ptr() if e.keep_activations else None, up.data_ptr(), B, C_, T, 9,
                                             KIND_ACT["lrelu"], e.stream()), "upsample_fwd")
                e.prof_end(tok)
            else:
                check(e.lib.nsc_depthwise_fwd(h.data_ptr(), e.p_ptr + 4 * dw_off, dwo.data_ptr(), B, C_, T, 9, e.stream()),
                      "depthwise_fwd")
                pw.fwd(dwo, up, "lrelu", out_mode=1)
            self.up_saved.append((h, dwo, up))
            h = up
        h = self.stack_fwd(self.dec_tail, h)
        self.dec_feat = h
        self.dec = e.buf(s + ".dec", (B, 1, frame_length))
        self.dec_out.fwd(h, self.dec, "none", chain=chain)
        return self.dec

    def entropy(self):
        """entropy_coding_loss from the (possibly all-reduced) histogram; also prepares d ent / d hist."""
        e, s = self.eng, self.scope
        self.ent = e.buf(s + ".ent", (1,))
        self.ghist = e.buf(s + ".ghist", (self.nb,))
        check(e.lib.nsc_entropy_from_hist(self.hist.data_ptr(), self.nb, self.ent.data_ptr(), self.ghist.data_ptr(),
                                          e.stream()), "entropy_from_hist")
        return self.ent

    def backward(self, ddec, c_quan, ent_scale, need_dx=False, chain=None):
        """ddec [B,1,512] = dL/d dec.  c_quan: coefficient on sum_b quan_loss[b]; ent_scale: coefficient on the
        entropy scalar (tau * global batch).  Accumulates parameter grads; returns dL/dx_in if need_dx."""
        e, s = self.eng, self.scope
        B = e.B
        # decoder
        dz = e.buf(s + ".d_decfeat", tuple(self.dec_feat.shape))
        self.dec_out.wgrad(self.dec_feat, ddec)
        self.dec_out.dgrad(ddec, dz)  # dec_feat comes from a flat block: no activation gradient
        dz = self.stack_bwd(self.dec_tail, dz, "lrelu" if self.dec_stages else "none")
        for i in range(len(self.dec_stages) - 1, -1, -1):
            blocks, dw_off, pw, T, C_ = self.dec_stages[i]
            xin, dwo, up = self.up_saved[i]
            dzp = e.buf(f"{s}.dzp{i}", (B, C_, T))
            ddw = e.buf(f"{s}.ddw{i}", (B, C_, T))
            dxu = e.buf(f"{s}.dxup{i}", (B, C_, T))
            if e.fused_up and C_ in (100, 50) and B * C_ * T < 2 ** 29:
                # un-shuffle -> pointwise^T -> depthwise^T in one kernel; the two weight gradients read dzp / ddw afterwards
                tok = e.prof_begin("upsample", pw.flops())
                check(e.lib.nsc_upsample_bwd(dz.data_ptr(), e.p_ptr + 4 * dw_off, e.p_ptr + 4 * pw.w_off, dzp.data_ptr(),
                                             ddw.data_ptr(), dxu.data_ptr(), B, C_, T, 9, e.stream()), "upsample_bwd")
                e.prof_end(tok)
                pw.wgrad(dwo, dzp)
                check(e.lib.nsc_depthwise_bwd(xin.data_ptr(), e.p_ptr + 4 * dw_off, ddw.data_ptr(), None,
                                              e.g_ptr + 4 * dw_off, B, C_, T, 9, e.stream()), "depthwise wgrad")
            else:
                check(e.lib.nsc_unshuffle2(dz.data_ptr(), dzp.data_ptr(), B, C_, T, e.stream()), "unshuffle2")
                pw.wgrad(dwo, dzp)
                pw.dgrad(dzp, ddw)
                check(e.lib.nsc_depthwise_bwd(xin.data_ptr(), e.p_ptr + 4 * dw_off, ddw.data_ptr(), dxu.data_ptr(),
                                              e.g_ptr + 4 * dw_off, B, C_, T, 9, e.stream()), "depthwise_bwd")
            dz = self.stack_bwd(blocks, dxu, "none" if i == 0 else "lrelu")   # stage 0 input is the quantized code
        dq = dz  # [B,1,L]: dL/d qcode
        # quantizer (+ fused quan / entropy loss gradients), returns dL/d(pre-tanh code)
        dcode = e.buf(s + ".dcode", (B, 1, self.L))
        check(e.lib.nsc_quantize_bwd(self.code.data_ptr(), e.p_ptr + 4 * self.alpha_off, e.p_ptr + 4 * self.bins_off,
                                     self.is_quan_on, self.soft, B, self.L, self.nb, dq.data_ptr(), None,
                                     float(c_quan), self.ghist.data_ptr() if ent_scale != 0.0 else None,
                                     float(ent_scale), 1, dcode.data_ptr(), e.g_ptr + 4 * self.alpha_off,
                                     e.g_ptr + 4 * self.bins_off, e.stream()), "quantize_bwd")
        # encoder
        self.enc_out.wgrad(self.enc_feat, dcode)
        dz = e.buf(s + ".d_encfeat", tuple(self.enc_feat.shape))
        self.enc_out.dgrad(dcode, dz)
        dz = self.stack_bwd(self.enc_tail, dz, "lrelu")   # input of block 0 is the lrelu output of the down conv / in conv
        for i in range(len(self.enc_stages) - 1, -1, -1):
            blocks, down = self.enc_stages[i]
            hin, dout = self.down_out[i]
            down.wgrad(hin, dz)
            dzi = e.buf(f"{s}.d_down{i}", tuple(hin.shape))
            down.dgrad(dz, dzi)                       # hin comes from a flat block
            dz = self.stack_bwd(blocks, dzi, "lrelu")
        self.in_conv.wgrad(self.x_in, dz)
        if need_dx:
            dx = e.buf(s + ".dx_in", (B, 1, frame_length))
            self.in_conv.dgrad(dz, dx, chain=chain)
            return dx
        return None


class _SegmentRecorder:
    """Captures consecutive hipGraph segments on the current (side) stream; cut() ends one and starts the next.  A cut with
    no launch since the previous one adds its action to the previous segment instead of capturing an empty graph."""

    def __init__(self, eng):
        self.eng, self.segs, self.cur, self.mark = eng, [], None, 0
        self.pool = torch.cuda.graph_pool_handle()

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool)
        self.mark = self.eng._nlaunch

    def cut(self, action):
        # "empty" is decided from the engine's launch counter: every GPU launch inside a captured step must go through
        # eng.stream() (a torch op or a raw-stream launch issued here would not be counted and could land in the segment that
        # replays AFTER the collective)
        if self.eng._nlaunch == self.mark and self.segs:
            self.segs[-1][1].append(action)
            return
        self.cur.capture_end()
        self.segs.append((self.cur, [action]))
        self.begin()

    def finish(self):
        self.cur.capture_end()
        self.segs.append((self.cur, []))


class SegmentedStep:
    """A train step as [graph, eager collective, graph, ...]: replay() launches the segments in order on the current stream;
    "sync" collectives block the stream behind them, "async" ones (gradient all-reduces on RCCL's own stream) overlap with
    the next segment and are waited for at the "wait" cut (just before the optimizer segment)."""

    def __init__(self, segs):
        self.segs = segs
        self.nseg = len(segs)
        self.ncoll = sum(1 for _, acts in segs for a in acts if a[1] is not None)

    def replay(self):
        handles = []
        for g, acts in self.segs:
            g.replay()
            for kind, fn in acts:
                if kind == "sync":
                    fn()
                elif kind == "async":
                    handles.append(fn())
                elif kind == "wait":
                    for h in handles:
                        if h is not None:
                            h.wait()
                    handles = []


class CascadeEngine:
    """N-codec CMRL cascade + losses + TF1 Adam on flat buffers (cmrl.py:22-135, 295-511)."""

    def __init__(self, batch, num_codecs=1, bkd=(9, 9, 100, 20, 1, 2), strides=None, num_bins=None, res_scalar=1.0,
                 scale_first=False, lpc=False, device="cuda", seed=20200504, layout_only=False):
        if not layout_only:
            if not torch.cuda.is_available():
                raise _lib.NscError("nsc_amd.engine needs a GPU: the HIP path has no CPU fallback")
            self.lib = _lib.load()
        self.B, self.N = int(batch), int(num_codecs)
        self.device = torch.device(device)
        self.bkd = list(bkd)
        strides = strides or [[2]] * self.N
        num_bins = num_bins or [32] * self.N
        self.res_scalar, self.scale_first, self.lpc = float(res_scalar), bool(scale_first), bool(lpc)
        self.layout = ParamLayout()
        self.convs = []
        self._bufs = {}
        self._wg_jobs, self._wg_keep, self._wg_flops = [], [], 0.0
        self._cw_jobs, self._cw_flops = [], 0.0
        self._sum_jobs = []
        self._live_tables = {}
        if self.lpc:  # 'lpc_quan' scope is created before scope_1 (nsc_module:993-996)
            self.lpc_alpha_off = self.layout.add("lpc_quan/alpha", ())
            self.lpc_bins_off = self.layout.add("lpc_quan/bins", (len(lpc_coeff_lsf_bins),))
        self.codecs = [_Codec(self, f"scope_{i + 1}", self.bkd, strides[i], num_bins[i]) for i in range(self.N)]
        n = self.layout.size
        # gather indices (nsc_gather / nsc_step_begin) keep bits 26..29 for the split-image word modes: offsets must fit 26 bits
        assert n < (1 << 26), "flat parameter buffer too large for the gather index encoding (bits 26..29 are mode bits)"
        if layout_only:
            return
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(n, **f32)
        # gradients and, right behind them, every quantizer's soft histogram live in ONE allocation padded to 64 KB: a
        # training step clears both with a single aligned memset (an unaligned size costs a second fill dispatch for the tail)
        nh = sum(nb_ for nb_ in num_bins) + 4 * len(lpc_coeff_lsf_bins)
        # ... and the neighbour flags of the pair launches (int32 behind the same memset): one slot per pair launch of a step
        self.cu_count = torch.cuda.get_device_properties(self.device).multi_processor_count
        # Pair launches need every workgroup of a launch resident at once, i.e. the GPU to this process (the deployment model: one
        # process per GPU).  Ranks that SHARE a device (tests: more local ranks than GPUs, gloo) would interleave two such launches,
        # neither complete: their neighbour waits would run into the time-out.  They launch the blocks one by one.
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        if lws > torch.cuda.device_count():
            self.fused_pairs = False
        self._flag_ints = int(self.lib.nsc_gated_block_pair_flag_ints())
        npairs = sum(len(c.enc_stages) + len(c.dec_stages) + 2 for c in self.codecs)    # enc / dec stages + enc tail + dec tail
        self._flag_slots = 2 * npairs
        nfl = (self._flag_ints + 3) // 4 * 4 * self._flag_slots
        nquan = self.N * self.B
        self._gh_floats = (n + nh + nfl + nquan + 16383) // 16384 * 16384
        self._gh = torch.zeros(self._gh_floats, **f32)
        self.grads = self._gh[:n]
        self._hist_flat = self._gh[n:n + nh]
        self._flags = self._gh[n + nh:n + nh + nfl].view(torch.int32)
        self._flag_next = 0
        self._pair_to = torch.zeros(4, dtype=torch.int32, device=self.device)
        self._quan_acc = self._gh[n + nh + nfl:n + nh + nfl + nquan].view(self.N, self.B)     # quan_loss per frame, accumulated by tiles
        self._hist_slots, self._hist_used = {}, 0
        # wt = flipped/transposed kernels of every conv at the same offsets as the parameters, followed by one extra
        # region per gated block holding the two k15 gate kernels concatenated along their OUTPUT channels
        # (wt_lr[tap', c' in 0..39, ci]) so that both gate data-gradients run as a single 40-channel conv.
        blocks = [b for c in self.codecs for b in c.all_blocks()]
        extra = 0
        self._wt_regions = {}                                   # first word -> words, of every region a kernel may be pointed at
        for c in self.convs:
            self._wt_regions[c.w_off] = c.K * c.Cin * c.Cout      # (the flipped / transposed copy of each conv kernel, at its own offset)
        for b in blocks:
            b.wtlr_off = n + extra
            self._wt_regions[b.wtlr_off] = b.cl.K * 2 * b.narrow * b.narrow
            extra += b.cl.K * 2 * b.narrow * b.narrow
        # polyphase data-gradient kernels of the stride-2 k9 convs: W'[t', o, 2 ci + p] (5 taps, structural zero at p=0, t'=4)
        # ... or, on split operands (csrc/conv_split.hip), kernel-ready images of the forward and the data-gradient GEMM instead
        poly = [c for c in self.convs if c.poly_ok()]
        for c in poly:
            c.simg_off = None
            nw = [int(self.lib.nsc_conv1d_simage_words(which, C.byref(c.desc()))) for which in (0, 1)] if self.split_conv else [0, 0]
            if nw[0] > 0 and nw[1] > 0:
                extra = (n + extra + 3) // 4 * 4 - n
                c.simg_off = (n + extra, n + extra + nw[0])
                c.simg_words = tuple(nw)
                self._wt_regions[c.simg_off[0]], self._wt_regions[c.simg_off[1]] = nw[0], nw[1]
                extra += nw[0] + nw[1]
            c.wtpoly_off = n + extra            # (kept beside the images: split_conv can be switched off at run time for A/B)
            self._wt_regions[c.wtpoly_off] = 5 * c.Cout * 2 * c.Cin
            extra += 5 * c.Cout * 2 * c.Cin
        # kernel-ready parameter images of the gated blocks the persistent kernels serve (fast prologue: include/nsc_hip.h,
        # nsc_gated_block_image_index): one forward and one data-gradient image per block, 16-byte aligned, rebuilt by the
        # same gather launch as the flipped kernels
        for b in blocks:
            b.img_fwd_off = b.img_bwd_off = b.simg_fwd_off = b.simg_bwd_off = None
            if b.narrow == 20 and b.c9.K == 9:
                for which, attr in ((0, "img_fwd_off"), (1, "img_bwd_off")):
                    nf = int(self.lib.nsc_gated_block_image_floats(which, int(b.wide), int(b.Cin), int(b.cl.dil)))
                    if nf > 0:
                        extra = (n + extra + 3) // 4 * 4 - n
                        setattr(b, attr, n + extra)
                        self._wt_regions[n + extra] = nf
                        extra += nf
                # ... and the SPLIT images of the bf16-matrix-core kernels (csrc/block_split.hip): 32-bit words of packed bf16 pieces
                # (the data-gradient images only when that kernel is switched on: they are re-gathered every step)
                for which, attr in ((0, "simg_fwd_off"), (2, "simg_bwd_off")):
                    if which == 2 and not self.split_dgrad:
                        continue
                    nf = int(self.lib.nsc_gated_block_simage_words(which, int(b.wide), int(b.Cin), int(b.cl.dil)))
                    if nf > 0:
                        extra = (n + extra + 3) // 4 * 4 - n
                        setattr(b, attr, n + extra)
                        self._wt_regions[n + extra] = nf
                        extra += nf
        self.wt = torch.zeros(n + extra, **f32)
        self.p_ptr, self.g_ptr, self.wt_ptr = self.params.data_ptr(), self.grads.data_ptr(), self.wt.data_ptr()
        assert self.wt_ptr % 16 == 0
        idx = np.arange(n + extra, dtype=np.int32)
        idx[n:] = -1                                  # alignment gaps between the extra regions: never read (gather writes 0)
        for c in self.convs:
            idx[c.w_off:c.w_off + c.K * c.Cin * c.Cout] = c.wt_index()
        for b in blocks:
            K, nn = b.cl.K, b.narrow
            il = b.cl.wt_index().reshape(K, nn, nn)      # [tap', c, ci] source offsets
            ir = b.cr.wt_index().reshape(K, nn, nn)
            idx[b.wtlr_off:b.wtlr_off + K * 2 * nn * nn] = np.concatenate([il, ir], axis=1).reshape(-1)
        for c in poly:
            if c.simg_off is not None:
                for which in (0, 1):
                    im = np.empty(c.simg_words[which], dtype=np.int32)
                    check(self.lib.nsc_conv1d_simage_index(which, C.byref(c.desc()), int(c.w_off), im.ctypes.data_as(C.c_void_p)),
                          "conv1d_simage_index")
                    idx[c.simg_off[which]:c.simg_off[which] + c.simg_words[which]] = im
            idx[c.wtpoly_off:c.wtpoly_off + 5 * c.Cout * 2 * c.Cin] = c.wtpoly_index()
        for b in blocks:
            for which, off in ((0, b.img_fwd_off), (1, b.img_bwd_off)):
                if off is None:
                    continue
                nf = int(self.lib.nsc_gated_block_image_floats(which, int(b.wide), int(b.Cin), int(b.cl.dil)))
                if which == 0:      # straight from the parameters
                    offs = [b.c1.w_off, b.c1.b_off, b.cl.w_off, b.cl.b_off, b.cr.w_off, b.cr.b_off, b.c9.w_off, b.c9.b_off]
                else:               # from the flipped / transposed kernels: compose with their own index map
                    offs = [b.c1.w_off, b.cl.w_off, b.cr.w_off, b.c9.w_off]
                im = np.empty(nf, dtype=np.int32)
                check(self.lib.nsc_gated_block_image_index(which, int(b.wide), int(b.Cin), int(b.cl.dil), (C.c_long * len(offs))(*offs),
                                                           im.ctypes.data_as(C.c_void_p)), "gated_block_image_index")
                if which == 1:
                    im = np.where(im >= 0, idx[np.maximum(im, 0)], -1).astype(np.int32)
                idx[off:off + nf] = im
            for which, off in ((0, b.simg_fwd_off), (2, b.simg_bwd_off)):
                if off is None:
                    continue     # split images, straight from the parameters (entries carry a mode in bits 26..29)
                nf = int(self.lib.nsc_gated_block_simage_words(which, int(b.wide), int(b.Cin), int(b.cl.dil)))
                offs = [b.c1.w_off, b.c1.b_off, b.cl.w_off, b.cl.b_off, b.cr.w_off, b.cr.b_off, b.c9.w_off, b.c9.b_off]
                im = np.empty(nf, dtype=np.int32)
                check(self.lib.nsc_gated_block_simage_index(which, int(b.wide), int(b.Cin), int(b.cl.dil), (C.c_long * len(offs))(*offs),
                                                            im.ctypes.data_as(C.c_void_p)), "gated_block_simage_index")
                idx[off:off + nf] = im
        self.wt_idx = torch.from_numpy(idx).to(self.device)
        # two Adam slot sets (no-quan op / quan op) with independent state (nsc_module:922-926)
        self.adam = [dict(m=torch.zeros(n, **f32), v=torch.zeros(n, **f32), t=0,
                          t_dev=torch.zeros(1, dtype=torch.int32, device=self.device)) for _ in range(2)]
        mel = mel_matrix_cat().astype(np.float32)
        self.mel = torch.from_numpy(mel).to(self.device).contiguous()
        self.melT = torch.from_numpy(np.ascontiguousarray(mel.T)).to(self.device)
        self.mel_ranges = torch.from_numpy(mel_band_ranges(mel)).to(self.device)
        self.init_params(seed)

    # ---- plumbing ----
    def stream(self):
        """Raw handle of the current stream, cached for the duration of a public call (the lookup costs ~8 us of host
        time and an engine step makes ~700 of them).  Every launch asks for it: _nlaunch counts them."""
        self._nlaunch += 1
        return self._st if self._st is not None else torch.cuda.current_stream().cuda_stream

    _st = None
    _nlaunch = 0

    def _enter(self):
        self._st = torch.cuda.current_stream().cuda_stream

    def _leave(self):
        self._st = None

    keep_activations = True   # False for inference-only use: the block forward then skips its 4 saved [B,20,T] tensors
    fused_fwd = True   # gated blocks run as one kernel (csrc/block.hip); False = one launch per conv
    fused_wgrad = True # all eight parameter gradients of a block in one persistent kernel (csrc/block.hip)
    fused_dgrad = True   # whole data path of a block (k9 -> GLU -> k15 -> 1x1 data gradients) in one persistent,
                         # weight-stationary kernel (v2; the per-tile v1 lost to the per-conv launches: 1.9 vs 1.2 ms/step)
    # per-launch HIP-event timing of the conv kernels (bench.py roofline); events sit on the launch stream
    prof = None

    # ---- weight gradients on a side stream: they are off the critical path (nothing downstream reads them before the
    # optimizer), and both they and the data-gradient chain are latency- rather than throughput-bound, so running them
    # concurrently on the same CUs is nearly additive.  fork = event on the main stream, join before Adam / all-reduce.
    overlap_wgrad = True
    wgrad_waves = 8      # waves per workgroup of the persistent block-wgrad kernel (8 fills the register file of a CU)
    split_wgrad = True   # under overlap: two light 4-wave launches per block instead of one heavy 8-wave launch

    n_side = 1   # side streams, used round-robin (2 measured no faster: 6.04 vs 5.96 ms/step)

    # ---- block weight gradients deferred to the end of the backward pass and produced by ONE persistent launch per
    # block width (nsc_gated_block_wgrad_batch): per-block launches walk only 2-4 tiles per workgroup at batch 128, so
    # their prologue, accumulator flush and slab reduction cost more than the MFMA work itself.
    fused_quant = True   # the training-shape quantizer forward in the launch of the encoder's output conv (nsc_conv1d_cout1_fwd_quant)
    # Arithmetic of the gated blocks' long contractions: False = the exact fp32 matrix instruction (csrc/block.hip); True = the bf16
    # matrix cores on operands split into three bf16 pieces, six products, fp32 accumulation (csrc/block_split.hip: fp32-class
    # error, the vector ALU left free: the default since round 5 - numerics gate: tests/test_fullsize_gpu.py::_numerics_gate, profiles/r06_numerics_gate_*.txt).
    # NSC_BLOCK_ARITH=exact|split overrides the default for A/B runs.
    split_fwd = os.environ.get("NSC_BLOCK_ARITH", "split") == "split"
    split_wgrad_arith = os.environ.get("NSC_BLOCK_ARITH", "split") == "split"   # (split_wgrad is the two-light-launches switch above)
    # the split-operand DATA gradient (gated_block_dgrad3_kernel) is correct and tested but no faster than the exact pair launches
    # yet (profiles/r05h_dgrad_split_time.txt: 0.8-0.98x): off unless asked for
    split_dgrad = os.environ.get("NSC_SPLIT_DGRAD", "0") == "1"
    # the stride-2 down-sampling convs (forward + data gradient) on split operands too (csrc/conv_split.hip); the images are laid out
    # at construction if this is on then - later it can only be switched OFF (A/B runs)
    split_conv = os.environ.get("NSC_BLOCK_ARITH", "split") == "split"
    fused_pairs = True   # the dil-1 / dil-2 blocks of a stack in ONE launch (nsc_gated_block_pair_fwd_img / _dgrad_img: neighbour flags
                         # between workgroups instead of a kernel boundary); False: one launch per block
    pair_bwd = os.environ.get("NSC_PAIR_BWD", "1") == "1"   # A/B: the data gradients of a stack as one pair launch (False: block by block)
    fused_chain = True   # the cascade step / output-gradient arithmetic between codecs rides in the epilogue of the Cout = 1 convs
                         # (nsc_conv1d_cout1_fwd_chain) instead of nsc_cascade_step / nsc_axpby launches
    poly_dgrad = True    # stride-2 data gradients in polyphase form (half the MFMAs of the zero-upsampled form)
    fused_up = True      # decoder up-sampling stage as one kernel per direction (nsc_upsample_fwd / _bwd)
    batch_wgrad = True
    batch_cin1_wgrad = True   # the one-input-channel decoder blocks join the batched launch too (False: per-conv launches)
    batch_conv_wgrad = True   # the same for the convs outside gated blocks (nsc_conv1d_wgrad_batch).  With the block kernels
                              # persistent at one workgroup per CU and a static split of the tiles, a weight-gradient workgroup
                              # that holds a CU's LDS when such a kernel starts delays that CU's whole share: per-conv launches
                              # on the side stream measured 3.57 ms/step when their timing happened to fall well and 3.82 when
                              # it did not; everything deferred to the tail of the step: 3.55, independent of timing

    # ---- data parallel: how the step meets its collectives ----
    dp_overlap = False   # True: one gradient all-reduce per trainable scope, started as soon as that codec's backward pass and
                         # weight gradients are done (runs under the earlier codecs' backward pass); False: one message for
                         # all scopes at the tail of the step (the batched weight-gradient launches stay whole).  Measured on
                         # one MI355X with the collectives as no-ops (bench.py --dp-selftest): 3.496 ms/step with per-scope
                         # flushes, 3.360 at the tail, 3.352 for the single-GPU graph - splitting the batched launches costs
                         # 0.14 ms, more than the ~0.05 ms an exposed 2.8 MB all-reduce takes
    _rec = None          # segment recorder while a step is being captured (capture_train_step)
    _dp_comm_attached = False   # a communicator is attached to the step being run / captured (train_step, capture_train_step)

    def _collective(self, kind, fn):
        """Every collective of a step goes through here.  Eager: run it.  While a step is being captured as hipGraph
        segments: close the current segment, remember the collective as the eager action that follows it, open the next."""
        if self._rec is None:
            assert self._st is None or self._st == torch.cuda.current_stream().cuda_stream, \
                "collective issued from a different stream than the engine's launches"
            return fn() if fn is not None else None
        self._rec.cut((kind, fn))
        return None

    def capture_train_step(self, x, target, cfg, lpc_x=None, comm=None):
        """Capture one train step as hipGraph SEGMENTS cut at the collectives (forward + loss | per-scope backward + weight
        gradients | Adam), so that the data-parallel step replays like the single-GPU one: RCCL calls stay eager between
        two graph launches (no collective is captured).  The caller has run the step eagerly at least once (buffers allocated).
        Returns a SegmentedStep; .replay() runs one step on the current stream."""
        rec = _SegmentRecorder(self)
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._rec = rec
            try:
                rec.begin()
                self.train_step(x, target, cfg, lpc_x=lpc_x, comm=comm)
                rec.finish()
            except BaseException:
                # leave the stream out of capture mode and drop the half-built segments, so that the caller's fallback
                # (eager launches in this process) starts from a clean stream
                try:
                    if rec.cur is not None:
                        rec.cur.capture_end()
                except Exception:
                    pass
                rec.segs.clear()
                raise
            finally:
                self._rec = None
        torch.cuda.current_stream().wait_stream(s)
        return SegmentedStep(rec.segs)

    def defer_block_wgrad(self, blk, dz, da, dz1, dw1_ptr, flops):
        self._wg_jobs.append(_lib.BlockWgradJob(blk.x.data_ptr(), blk.h.data_ptr(), blk.g.data_ptr(), dz.data_ptr(),
                                                da.data_ptr(), dz1.data_ptr(), dw1_ptr, blk.wide, blk.T, blk.cl.dil, blk.Cin))
        self._wg_keep += [blk.x, blk.h, blk.g, dz, da, dz1]
        self._wg_flops += flops

    def defer_conv_wgrad(self, desc, x, dz, dw_ptr, db_ptr, flip):
        self._cw_jobs.append(_lib.ConvWgradJob(desc, x.data_ptr(), dz.data_ptr(), dw_ptr, db_ptr, flip))
        self._wg_keep += [x, dz]

    def flush_conv_wgrads(self):
        for lo in range(0, len(self._sum_jobs), 8):
            chunk = self._sum_jobs[lo:lo + 8]
            check(self.lib.nsc_sum_all_batch((_lib.SumJob * len(chunk))(*chunk), len(chunk), self.stream()), "sum_all_batch")
        self._sum_jobs = []
        if not self._cw_jobs:
            return
        tok = self.prof_begin("wgrad_mfma", self._cw_flops)
        if self.split_conv:
            # the stride-2 down-sampling convs: their own launch on split operands (csrc/conv_split.hip), up to 8 jobs per call
            sp = [j for j in self._cw_jobs if not j.flip_taps and int(self.lib.nsc_conv1d_simage_words(0, C.byref(j.d))) > 0]
            if sp:
                self._cw_jobs = [j for j in self._cw_jobs if not any(j is q for q in sp)]
                need = int(self.lib.nsc_conv1d_wgrad_split_workspace())
                ws = self._bufs.get("cwgrad.split.ws")
                if ws is None:
                    ws = torch.empty(need, dtype=torch.float32, device=self.device)
                    self._bufs["cwgrad.split.ws"] = ws
                for lo in range(0, len(sp), 8):
                    chunk = sp[lo:lo + 8]
                    check(self.lib.nsc_conv1d_wgrad_split((_lib.ConvWgradJob * len(chunk))(*chunk), len(chunk), ws.data_ptr(), ws.numel(),
                                                          self.stream()), "conv1d_wgrad_split")
        if not self._cw_jobs:
            self.prof_end(tok)
            self._cw_flops = 0.0
            return
        n = len(self._cw_jobs)
        jobs = (_lib.ConvWgradJob * n)(*self._cw_jobs)
        need = int(self.lib.nsc_conv1d_wgrad_batch_workspace(jobs, n))
        ws = self._bufs.get("cwgrad.ws")
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.float32, device=self.device)
            self._bufs["cwgrad.ws"] = ws
        check(self.lib.nsc_conv1d_wgrad_batch(jobs, n, ws.data_ptr(), ws.numel(), self.stream()), "conv1d_wgrad_batch")
        self.prof_end(tok)
        self._cw_jobs, self._cw_flops = [], 0.0

    # The two batches of deferred weight gradients at the tail of the step (convs | gated blocks) are independent: the convs' launches go
    # to a second stream beside the blocks' (joined before Adam / the gradient message).  Their persistent kernels mostly cannot share
    # a CU (LDS, registers), but each launch's ramp-down runs under the next one's start: 2.445 -> 2.401 ms per step, hipGraph replay
    # (same box, two runs each).  Safe here and only here: no pair launch (all workgroups resident, neighbour flags) runs at the tail.
    tail_overlap = os.environ.get("NSC_TAIL_OVERLAP", "1") == "1"

    def _tail_two_streams(self):
        """The second stream at the tail is for the step that overlaps nothing else: with per-scope gradient messages under the backward
        pass (dp_overlap + a communicator) a collective's kernel may already hold compute units, and two persistent launches beside it
        are the residency squeeze _pair_ok avoids - one stream then."""
        return self.tail_overlap and self.prof is None and not (self.dp_overlap and self._dp_comm_attached)

    def flush_block_wgrads(self):
        if self._tail_two_streams() and self._cw_jobs and self._wg_jobs:
            if self._tail_stream is None:
                self._tail_stream = torch.cuda.Stream(device=self.device)
            side, cur = self._tail_stream, torch.cuda.current_stream()
            side.wait_stream(cur)
            keep = self._st
            self._st = side.cuda_stream
            try:
                self.flush_conv_wgrads()
            finally:
                self._st = keep
            self._tail_join = side
        else:
            self.flush_conv_wgrads()
        if not self._wg_jobs:
            self._wg_keep = []
            return
        jobs = (_lib.BlockWgradJob * len(self._wg_jobs))(*self._wg_jobs)
        # private (per-conv wgrads may still be running on the side stream) and twice the size: the launches of the two block widths
        # keep their slabs side by side and ONE reduce launch sums both
        ws = self.wgrad_workspace(slot="batch", mult=2)
        tok = self.prof_begin("block_wgrad", self._wg_flops)
        # main stream: ordered after every data-gradient kernel, and after nothing else that matters (tail of the step)
        fn = self.lib.nsc_gated_block_wgrad_batch_split if self.split_wgrad_arith else self.lib.nsc_gated_block_wgrad_batch
        check(fn(jobs, len(self._wg_jobs), self.B, 20, 9, ws, 2 * self._ws_floats, self.stream()), "gated_block_wgrad_batch")
        self.prof_end(tok)
        self._wg_jobs, self._wg_keep, self._wg_flops = [], [], 0.0
        if self._tail_join is not None:
            torch.cuda.current_stream().wait_stream(self._tail_join)
            self._tail_join = None

    _tail_stream = None
    _tail_join = None

    def side_fork(self):
        """Returns the stream handle weight-gradient kernels should be launched on (a side stream ordered after everything
        enqueued so far on the current stream), or the current stream when overlap is off."""
        if not self.overlap_wgrad:
            self.side_idx = 0
            return self.stream()
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.n_side)]
        self.side_idx = self._side_rr % len(self._side)    # also selects the slab workspace of launches on this stream
        s = self._side[self.side_idx]
        self._side_rr += 1
        if not isinstance(self._fork_evs, list):
            self._fork_evs = []
        if self._fork_ev_i == len(self._fork_evs):
            self._fork_evs.append(torch.cuda.Event())      # created once, reused every step
        ev = self._fork_evs[self._fork_ev_i]
        self._fork_ev_i += 1
        ev.record(torch.cuda.current_stream())
        s.wait_event(ev)
        self._side_used = True
        return s.cuda_stream

    def side_join(self):
        if self._side is not None and self._side_used:
            for s in self._side:
                torch.cuda.current_stream().wait_stream(s)
            self._side_used = False

    _side = None
    _fork_evs = ()
    _fork_ev_i = 0
    _side_used = False
    _side_rr = 0
    side_idx = 0

    def wgrad_workspace(self, slot=0, mult=1):
        """Scratch for the store+reduce flush of nsc_gated_block_wgrad / nsc_conv1d_wgrad_ws.  One buffer per slot:
        launches that may run concurrently (different side streams) must not share a slab; launches on one stream are
        ordered.  Sized ONCE for the largest user so it is never reallocated under a running kernel."""
        if self._ws_floats is None:
            n = 0
            for c in self.codecs:
                for blk in c.all_blocks():
                    if blk.Cin > 1:
                        n = max(n, int(self.lib.nsc_gated_block_wgrad_workspace(int(blk.Cin))))
            for cv in self.convs:
                n = max(n, int(self.lib.nsc_conv1d_wgrad_workspace(C.byref(cv.wgrad_desc()))))
            self._ws_floats = n
        key = f"wgrad.ws{slot}"
        ws = self._bufs.get(key)
        if ws is None:
            ws = torch.empty(mult * self._ws_floats, dtype=torch.float32, device=self.device)
            self._bufs[key] = ws
        return ws.data_ptr()

    _ws_floats = None

    def prof_begin(self, tag, flops):
        if self.prof is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        return (tag, flops, a, b)

    def prof_end(self, tok):
        if tok is not None:
            tok[3].record()
            self.prof.append(tok)

    def prof_summary(self):
        """tag -> (launches, total ms, total algorithmic FLOPs); call after torch.cuda.synchronize()."""
        out = {}
        for tag, fl, a, b in self.prof or []:
            n, ms, f = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, ms + a.elapsed_time(b), f + fl)
        return out

    def pair_flags(self):
        """Pointer to the next zeroed flag slot of this step / forward (reset in forward(); zeroed by the step's opening launch, or
        by forward() itself outside a training step)."""
        k = self._flag_next
        assert k < self._flag_slots, "more pair launches than flag slots"
        self._flag_next += 1
        stride = (self._flag_ints + 3) // 4 * 4
        return self._flags.data_ptr() + 4 * stride * k

    def pair_timeouts_ptr(self):
        """The STICKY time-out counter of the pair launches: one int outside everything a step zeroes; the kernels only add to it
        (allocated with the engine - ADVICE r5: a lazy allocation inside a graph capture would have captured its zero-fill)."""
        return self._pair_to.data_ptr()

    def pair_timeouts(self):
        """Neighbour waits of pair launches that timed out since this engine was created (must be 0: the launch that counted one
        read unpublished data).  Synchronises the device."""
        return int(self._pair_to[0].item())

    def hist_view(self, key, nb):
        """[nb] slice of the flat histogram buffer (codecs' 32-bin and the LSF quantizer's 256-bin histograms side by side)."""
        if key not in self._hist_slots:
            assert self._hist_used + nb <= self._hist_flat.numel()
            self._hist_slots[key] = (self._hist_used, nb)
            self._hist_used += nb
        o, n = self._hist_slots[key]
        assert n == nb
        return self._hist_flat[o:o + n]

    _hist_flat = None
    _hist_clean = False

    def zero_hists_once(self):
        """All quantizers' histograms live in one flat buffer: zero it once per forward (first quantizer to ask)."""
        if not self._hist_clean and self._hist_flat is not None:
            check(self.lib.nsc_zero(self._hist_flat.data_ptr(), self._hist_flat.numel(), self.stream()), "zero hists")
            self._hist_clean = True

    def buf(self, name, shape):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=torch.float32, device=self.device)
            if _POISON_BUFS:      # debugging aid (NSC_POISON_BUFS=1): a kernel that reads an element nobody wrote meets NaN, not
                t.fill_(float("nan"))   # the zeros of a fresh allocation or last step's value
            self._bufs[name] = t
        return t

    def view(self, name, which="params"):
        off, shape = self.layout.entries[name]
        n = int(np.prod(shape)) if len(shape) else 1
        base = getattr(self, which)
        return base[off:off + n].view(shape if len(shape) else (1,))

    def init_params(self, seed=20200504):
        """Glorot-uniform kernels, zero biases, alpha=-300, bins=linspace (nsc_module:268-269; SURVEY 8c seeds)."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.layout.size, np.float32)
        for name, (off, shape) in self.layout.entries.items():
            n = int(np.prod(shape)) if len(shape) else 1
            if name.endswith("/alpha"):
                host[off] = init_alpha
            elif name == "lpc_quan/bins":
                host[off:off + n] = np.asarray(lpc_coeff_lsf_bins, np.float32)
            elif name.endswith("/bins"):
                host[off:off + n] = np.linspace(-beta_boundary, beta_boundary, n)
            elif name.endswith("/bias"):
                pass
            else:
                K, Ci, Co = shape
                if name.endswith("depthwise_kernel"):
                    fi, fo = K * Ci, K
                elif name.endswith("pointwise_kernel"):
                    fi, fo = Ci, Co
                else:
                    fi, fo = K * Ci, K * Co
                lim = math.sqrt(6.0 / (fi + fo))
                host[off:off + n] = rng.uniform(-lim, lim, size=n).astype(np.float32)
        self.params.copy_(torch.from_numpy(host))
        self.images_valid = False

    def load_named(self, named):
        """Copy a name->array dict (oracle ParamStore layout, TF shapes) into the flat buffer."""
        host = self.params.cpu().numpy()
        for name, (off, shape) in self.layout.entries.items():
            if name in named:
                a = np.asarray(named[name], np.float32).reshape(-1)
                host[off:off + a.size] = a
        self.params.copy_(torch.from_numpy(host))
        self.images_valid = False

    def named(self, which="params"):
        host = getattr(self, which).detach().cpu().numpy()
        out = OrderedDict()
        for name, (off, shape) in self.layout.entries.items():
            n = int(np.prod(shape)) if len(shape) else 1
            out[name] = host[off:off + n].reshape(shape).copy()
        return out

    # The forward images mirror self.params.  They are valid iff refresh_wt has run since the last write to the parameters:
    # writes by this engine's own kernels (Adam) clear the flag, writes through torch (params.copy_, eng.view(name)[...] = ..,
    # optimizers stepping on views) bump the tensor's version counter, which refresh_wt remembers.  A hipGraph captured with
    # valid images bakes the image path in: a captured forward must contain refresh_wt() or be re-captured after a weight load.
    _img_ok = False
    _img_version = -1
    use_images = True      # fast prologue of the persistent block kernels from the images (False: A/B and tests)

    @property
    def images_valid(self):
        return (self._img_ok and self.params._version == self._img_version and
                (self._img_live_flags is None or self._img_live_flags == self._flags_key()))

    @images_valid.setter
    def images_valid(self, v):
        self._img_ok = bool(v)
        self._img_version = self.params._version if v else -1

    def set_params(self, flat):
        """Overwrite the flat parameter buffer (another engine's .params, a checkpoint) and invalidate everything derived from it."""
        self.params.copy_(flat)
        self.images_valid = False

    # ---- which words of wt a step reads: most of the index map serves the OTHER arithmetic arm and the unfused paths (exact forward
    # images, the k15 gate concatenation, the polyphase kernels, the flipped copies of the block convs: 1.7 M of the headline step's 3.8 M
    # words).  Every pointer into wt goes through wtp(); a step's FIRST run with a given (flags, trainable pattern) gathers everything and
    # records the regions it was pointed at, later runs gather those only (nsc_step_begin_chunks).  A forward outside a training step sees
    # images_valid only while the flags are the ones the last gather was recorded for. ----
    live_gather = os.environ.get("NSC_LIVE_GATHER", "1") == "1"
    _wt_rec = None
    _img_live_flags = None

    def wtp(self, off):
        if self._wt_rec is not None:
            self._wt_rec.add(int(off))
        return self.wt_ptr + 4 * off

    def _flags_key(self):
        return (self.split_fwd, self.split_wgrad_arith, self.split_conv, self.split_dgrad, self.fused_fwd, self.fused_dgrad, self.fused_wgrad,
                self.fused_pairs, self.pair_bwd, self.use_images, self.poly_dgrad, self.fused_up, self.batch_wgrad, self.batch_cin1_wgrad,
                self.batch_conv_wgrad, self.overlap_wgrad, self.fused_chain, self.fused_quant,
                self.dp_overlap and self._dp_comm_attached)

    def _chunk_table(self, offs):
        rows = []
        for off in sorted(offs):
            nw = self._wt_regions[off]
            for lo in range(0, nw, 1024):
                rows.append((off + lo, min(1024, nw - lo)))
        return torch.tensor(np.asarray(rows, dtype=np.int32).reshape(-1, 2), device=self.device), len(rows)

    def refresh_wt(self):
        """Rebuild everything derived from the parameters: the flipped / transposed data-gradient kernels and the kernel-ready
        block images.  train_step does it every step; an inference user calls it once after loading weights (without it the
        forward simply takes the slower prologue that reads the parameters themselves)."""
        check(self.lib.nsc_gather(self.p_ptr, self.wt_idx.data_ptr(), self.wt_ptr, self.wt.numel(), self.stream()),
              "gather wt")
        self.images_valid = True
        self._img_live_flags = None

    # ---- forward ----
    def forward(self, x, is_quan_on=1.0, soft=True, lpc_x=None, want_p=False, hists_clean=False, first_needed=0):
        """x [B,1,512] (time-domain frame, or the fed LPC residual).  Returns decoded [B,1,512] (sum of codecs).
        hists_clean: the caller has just zeroed the histogram buffer (train_step's memset covers it).
        first_needed: index of the first codec a backward pass will go through (train_step: the first trainable scope; the
        frozen codecs in front of it - a follower step's earlier scopes, cmrl.py:106-113 - run forward-only and keep nothing)."""
        e = self
        B, rs = self.B, self.res_scalar
        assert tuple(x.shape) == (B, 1, frame_length) and x.dtype == torch.float32 and x.is_contiguous()
        self._hist_clean = bool(hists_clean)
        self._flag_next = 0
        if not hists_clean:
            # outside a training step (whose opening launch zeroes all of it): histograms, pair flags and the quan accumulators of
            # the fused quantizer stage sit side by side behind the gradients - one memset
            n = self.layout.size
            check(self.lib.nsc_zero(self._gh.data_ptr() + 4 * n, self._gh_floats - n, self.stream()), "zero hists + flags + quan")
            self._hist_clean = True
        self.x = x
        n = B * frame_length
        self.decoded = self.buf("decoded", (B, 1, frame_length))
        self.xin = []
        for i, c in enumerate(self.codecs):
            scaled = (i > 0) or self.scale_first
            if i == 0:
                if scaled and rs != 1.0:
                    xin = self.buf("xin0", (B, 1, frame_length))
                    check(self.lib.nsc_axpby(x.data_ptr(), None, xin.data_ptr(), rs, 0.0, n, self.stream()), "axpby")
                else:
                    xin = x
            else:
                xin = xin_next        # written by the cascade step that closed the previous codec
            self.xin.append(xin)
            keep = self.keep_activations
            self.keep_activations = keep and i >= first_needed
            sc = (1.0 / rs) if scaled else 1.0
            # decoded (+)= sc * dec, and the next codec's input rs * (x - decoded): in the epilogue of the codec's output conv
            # (fused_chain; otherwise nsc_cascade_step, one launch per codec)
            xin_next = self.buf(f"xin{i + 1}", (B, 1, frame_length)) if i + 1 < self.N else None
            chain = None
            if self.fused_chain:
                chain = _lib.Cout1Chain(self.decoded.data_ptr() if i > 0 else None, self.decoded.data_ptr(), 1.0, sc,
                                        x.data_ptr() if xin_next is not None else None, _lib.ptr(xin_next), rs, -rs)
            try:
                dec = c.forward(xin, is_quan_on, soft, want_p, chain=chain)
            finally:
                self.keep_activations = keep
            if chain is None:
                check(self.lib.nsc_cascade_step(dec.data_ptr(), self.decoded.data_ptr(), int(i > 0), x.data_ptr(), _lib.ptr(xin_next),
                                                sc, rs, n, self.stream()), "cascade_step")
        if self.lpc and lpc_x is not None:
            # LSF quantizer (nsc_module:993-1005): only its soft assignment enters the loss (py_func has no grad)
            L, nb = lpc_x.shape[1], len(lpc_coeff_lsf_bins)
            self.lpc_x = lpc_x
            self.lpc_q = self.buf("lpc.q", (B, L, 1))
            self.lpc_quan = self.buf("lpc.quan", (B,))
            self.lpc_hist = self.hist_view("lpc", nb)
            self.zero_hists_once()
            self.lpc_p = self.buf("lpc.p", (B, L, nb)) if want_p else None
            check(self.lib.nsc_quantize_fwd(lpc_x.data_ptr(), self.p_ptr + 4 * self.lpc_alpha_off,
                                            self.p_ptr + 4 * self.lpc_bins_off, float(is_quan_on), int(bool(soft)), B, L,
                                            nb, _lib.ptr(self.lpc_p), self.lpc_q.data_ptr(), self.lpc_quan.data_ptr(),
                                            self.lpc_hist.data_ptr(), self.stream()), "lpc quantize_fwd")
        return self.decoded

    # ---- losses + backward ----
    def loss_backward(self, target, c_time, c_freq, c_quan, c_ent, trainable, c_quan_lpc=0.0, c_ent_lpc=0.0,
                      global_batch=None, hist_allreduce=None, train_lpc=None, grad_allreduce=None):
        """loss = sum_b [c_time*time + c_freq*freq + sum_i c_quan[i]*quan_i[b]] + Bglobal * sum_i c_ent[i]*ent_i
        (vector loss => implicit sum over the batch).  trainable: list of bool per codec.
        grad_allreduce (data parallel): t -> handle with .wait(); called once per trainable scope with that scope's
        range of the flat gradient buffer as soon as it is complete (SUM over ranks), all handles waited on before returning.
        Returns dict of loss terms (device tensors)."""
        B, rs = self.B, self.res_scalar
        Bg = float(global_batch or B)
        self._fork_ev_i = 0          # the fork events are reused from the start of every backward pass
        self.time = self.buf("loss.time", (B,))
        self.freq = self.buf("loss.freq", (B,))
        G = self.buf("loss.G", (B, 1, frame_length))
        first_needed = min([i for i, t in enumerate(trainable) if t], default=self.N)
        # d dec_i = sc_i * (G - rs * sum_{j>i} dxin_j): when every codec that takes part has the same sc (the usual case), the
        # loss kernel writes sc * G directly and the last codec's d dec is that buffer - one launch less
        scs = {(1.0 / rs) if (i > 0 or self.scale_first) else 1.0 for i in range(first_needed, self.N)}
        gsc = scs.pop() if len(scs) == 1 else 1.0
        check(self.lib.nsc_recon_loss_banded(self.decoded.data_ptr(), target.data_ptr(), B, float(c_time) * gsc, float(c_freq) * gsc,
                                             None, None, self.mel.data_ptr(), self.melT.data_ptr(),
                                             self.mel_ranges.data_ptr(), self.time.data_ptr(), self.freq.data_ptr(),
                                             G.data_ptr(), self.stream()), "recon_loss")
        if hist_allreduce is not None:
            # every quantizer's soft histogram in one message
            self._collective("sync", lambda: hist_allreduce([self._hist_flat[:self._hist_used]]))
        ents = self.entropies()
        n = B * frame_length
        dsum = None  # running sum over later codecs of dL/dxin_j
        ddec_ready = False
        pending = []
        # LSF quantizer first: its loss terms do not depend on the codecs' backward pass, and its two gradients then travel in
        # the same all-reduce message as scope_1 (adjacent in the flat buffer) instead of a third one
        lpc_rng = None
        ent_lpc = None
        if train_lpc is None:
            train_lpc = c_quan_lpc != 0.0 or c_ent_lpc != 0.0
        if self.lpc and train_lpc and hasattr(self, "lpc_x"):
            L, nb = self.lpc_x.shape[1], len(lpc_coeff_lsf_bins)
            ent_lpc, gh = self.buf("lpc.ent", (1,)), self.buf("lpc.ghist", (nb,))      # filled by entropies() above
            check(self.lib.nsc_quantize_bwd(self.lpc_x.data_ptr(), self.p_ptr + 4 * self.lpc_alpha_off,
                                            self.p_ptr + 4 * self.lpc_bins_off, self.codecs[0].is_quan_on,
                                            self.codecs[0].soft, B, L, nb, None, None, float(c_quan_lpc),
                                            gh.data_ptr() if c_ent_lpc != 0.0 else None, float(c_ent_lpc) * Bg, 0, None,
                                            self.g_ptr + 4 * self.lpc_alpha_off, self.g_ptr + 4 * self.lpc_bins_off,
                                            self.stream()), "lpc quantize_bwd")
            if grad_allreduce is not None:
                lpc_rng = self.layout.scope_range("lpc_quan")    # sent together with the adjacent scope_1 range below
        for i in range(self.N - 1, -1, -1):
            c = self.codecs[i]
            if i < first_needed:
                break
            scaled = (i > 0) or self.scale_first
            sc = (1.0 / rs) if scaled else 1.0
            ddec = self.buf(f"ddec{i}", (B, 1, frame_length))
            # d yhat_i = G - rs * sum_{j>i} dxin_j ; d dec_i = d yhat_i * sc   (G already carries gsc)
            if ddec_ready:
                pass                                   # written by the previous codec's input-conv data gradient (fused_chain)
            elif dsum is None:
                if sc == gsc:
                    ddec = G
                else:
                    check(self.lib.nsc_axpby(G.data_ptr(), None, ddec.data_ptr(), sc / gsc, 0.0, n, self.stream()), "axpby")
            else:
                check(self.lib.nsc_axpby(G.data_ptr(), dsum.data_ptr(), ddec.data_ptr(), sc / gsc, -rs * sc, n, self.stream()),
                      "axpby")
            need_dx = i > first_needed
            chain, ddec_ready = None, False
            if need_dx and self.fused_chain:
                # S_i = S_{i+1} + dxin_i and d dec_{i-1} = sc' G / gsc - rs sc' S_i, both in the epilogue of this codec's
                # input-conv data gradient (they were one or two nsc_axpby launches)
                sc_prev = (1.0 / rs) if ((i - 1 > 0) or self.scale_first) else 1.0
                S = self.buf(f"dsum{i}", (B, 1, frame_length))
                chain = _lib.Cout1Chain(dsum.data_ptr() if dsum is not None else None, S.data_ptr(), 1.0, 1.0,
                                        G.data_ptr(), self.buf(f"ddec{i - 1}", (B, 1, frame_length)).data_ptr(),
                                        sc_prev / gsc, -rs * sc_prev)
            if trainable[i]:
                dx = c.backward(ddec, c_quan[i], c_ent[i] * Bg, need_dx=need_dx, chain=chain)
            else:
                raise NotImplementedError("a frozen codec between trainable ones is not a reference configuration")
            if chain is not None:
                dsum, ddec_ready = S, True
            if grad_allreduce is not None and self.dp_overlap:
                # data parallel: this codec's gradients are complete once its deferred weight gradients have run; send them
                # now, under the backward pass of the earlier codecs (scope = one contiguous range of the flat buffer)
                self.flush_block_wgrads()
                self.side_join()
                a, b = self.layout.scope_range(f"scope_{i + 1}")
                if lpc_rng is not None and lpc_rng[1] == a:
                    a, lpc_rng = lpc_rng[0], None
                pending.append(self._collective("async", lambda a=a, b=b: grad_allreduce(self.grads[a:b])))
            if need_dx and chain is None:
                if dsum is None:
                    dsum = dx
                else:
                    acc = self.buf("dsum", (B, 1, frame_length))
                    check(self.lib.nsc_axpby(dx.data_ptr(), dsum.data_ptr(), acc.data_ptr(), 1.0, 1.0, n, self.stream()), "axpby")
                    dsum = acc
        if lpc_rng is not None and self.dp_overlap:
            pending.append(self._collective("async", lambda r=lpc_rng: grad_allreduce(self.grads[r[0]:r[1]])))
        self.flush_block_wgrads()
        self.side_join()
        if grad_allreduce is not None and not self.dp_overlap:
            # one message for everything trainable, after the batched weight-gradient launches at the tail of the step (the
            # launches stay whole: per-codec flushes split them); frozen scopes in front of the first trainable one are not sent
            # (the LSF quantizer's two gradients adjoin scope_1: they ride along when scope_1 trains, and travel as a small
            # message of their own when it is frozen - never the frozen scope's zeros; nothing trainable: nothing is sent)
            if first_needed < self.N:
                lo = self.layout.scope_range(f"scope_{first_needed + 1}")[0]
                hi = self.layout.scope_range(f"scope_{self.N}")[1]
                if lpc_rng is not None and lpc_rng[1] == lo:
                    lo, lpc_rng = lpc_rng[0], None
                pending.append(self._collective("async", lambda lo=lo, hi=hi: grad_allreduce(self.grads[lo:hi])))
            if lpc_rng is not None:
                pending.append(self._collective("async", lambda r=lpc_rng: grad_allreduce(self.grads[r[0]:r[1]])))
        if grad_allreduce is not None:
            self._collective("wait", None)
        for w in pending:
            if w is not None:
                w.wait()
        return dict(time=self.time, freq=self.freq, quan=[c.quan for c in self.codecs], ent=ents, ent_lpc=ent_lpc,
                    quan_lpc=getattr(self, "lpc_quan", None))

    def entropies(self):
        """entropy_coding_loss of every quantizer from its (possibly all-reduced) batch histogram, and d ent / d hist, in ONE
        launch: the codecs' (-> c.ent, c.ghist; returned) and, on the LPC path, the LSF quantizer's (-> lpc.ent, lpc.ghist)."""
        jobs = []
        for c in self.codecs:
            c.ent = self.buf(c.scope + ".ent", (1,))
            c.ghist = self.buf(c.scope + ".ghist", (c.nb,))
            jobs.append(_lib.EntropyJob(c.hist.data_ptr(), c.ent.data_ptr(), c.ghist.data_ptr(), c.nb))
        if self.lpc and hasattr(self, "lpc_x"):
            nb = len(lpc_coeff_lsf_bins)
            jobs.append(_lib.EntropyJob(self.lpc_hist.data_ptr(), self.buf("lpc.ent", (1,)).data_ptr(),
                                        self.buf("lpc.ghist", (nb,)).data_ptr(), nb))
        for lo in range(0, len(jobs), 8):
            chunk = jobs[lo:lo + 8]
            check(self.lib.nsc_entropy_from_hist_batch((_lib.EntropyJob * len(chunk))(*chunk), len(chunk), self.stream()),
                  "entropy_from_hist_batch")
        return [c.ent for c in self.codecs]

    def frame_entropies(self, x, lpc_x=None):
        """Per-frame entropy (bits) of every codec's soft assignment, as the reference's validation loop measures it:
        frames are fed one at a time with the_share=False, is_quan_on=1 (nsc_module:685-722), so entropy_coding_loss sees
        the histogram of a single frame.  Here the whole batch goes through one forward and nsc_frame_entropy evaluates
        each frame's own histogram.  Returns a [num_codecs, B] tensor."""
        self._enter()
        try:
            self.forward(x, 1.0, False, lpc_x=lpc_x, want_p=True)
            out = self.buf("val.frame_ent", (self.N, self.B))
            for i, c in enumerate(self.codecs):
                check(self.lib.nsc_frame_entropy(c.p.data_ptr(), self.B, c.L, c.nb, out[i].data_ptr(), self.stream()),
                      "frame_entropy")
            if self.lpc and lpc_x is not None:
                # the LSF quantizer's per-frame entropy (end2end_eval_lpc's ent_loss_list[0], nsc_module:1040-1041)
                self.lpc_frame_ent = self.buf("val.lpc_frame_ent", (self.B,))
                check(self.lib.nsc_frame_entropy(self.lpc_p.data_ptr(), self.B, lpc_x.shape[1], len(lpc_coeff_lsf_bins),
                                                 self.lpc_frame_ent.data_ptr(), self.stream()), "frame_entropy lpc")
            return out
        finally:
            self._leave()

    def adam_step(self, scopes, lr, slot=1, beta1=0.9, beta2=0.999, eps=1e-8, counted=False):
        """TF1 Adam on the flat ranges of the given scopes (independent state per optimizer slot).
        counted: the device step counter of this slot was already advanced for this step (train_step does it in nsc_step_begin)."""
        st = self.adam[slot]
        st["t"] += 1   # host mirror; the kernel reads the device counter so a captured graph stays valid
        if not counted:
            check(self.lib.nsc_increment(st["t_dev"].data_ptr(), self.stream()), "increment")
        # adjacent scopes are adjacent ranges of the flat buffers: merge them (one launch for the joint step)
        ranges = []
        for sc in scopes:
            a, b = self.layout.scope_range(sc)
            if ranges and ranges[-1][1] == a:
                ranges[-1][1] = b
            else:
                ranges.append([a, b])
        self.images_valid = False
        for a, b in ranges:
            check(self.lib.nsc_adam_tf1_step(self.p_ptr + 4 * a, self.g_ptr + 4 * a, st["m"].data_ptr() + 4 * a,
                                             st["v"].data_ptr() + 4 * a, b - a, float(lr), beta1, beta2, eps, st["t"],
                                             st["t_dev"].data_ptr(), self.stream()), "adam")

    def reset_adam(self):
        for st in self.adam:
            st["m"].zero_(); st["v"].zero_(); st["t_dev"].zero_(); st["t"] = 0

    def train_step(self, x, target, cfg, lpc_x=None, comm=None):
        """One optimizer step (nsc_module:455-458 sess.run(trainop)): zero grads, refresh dgrad weights, forward,
        losses, backward, [gradient all-reduce], TF1 Adam.  cfg: dict(is_quan_on, c_time, c_freq, c_quan, c_ent,
        trainable, lr, slot, c_quan_lpc, c_ent_lpc).  comm: nsc_amd.dist.Comm or None."""
        self._enter()
        try:
            return self._train_step(x, target, cfg, lpc_x, comm)
        finally:
            self._leave()

    def _train_step(self, x, target, cfg, lpc_x, comm):
        self._dp_comm_attached = comm is not None
        # ONE launch opens the step: zero gradients + histograms, rebuild the data-gradient kernels / parameter images (refresh_wt),
        # advance the optimizer's device step counter (read by the Adam launch at the end of this step)
        slot = cfg.get("slot", 1)
        flags = self._flags_key()
        key = (flags, tuple(bool(t) for t in cfg["trainable"]), lpc_x is not None)
        live = self._live_tables.get(key) if self.live_gather else None
        if live is not None:
            check(self.lib.nsc_step_begin_chunks(self.p_ptr, self.wt_idx.data_ptr(), self.wt_ptr, live[0].data_ptr(), live[1], self.g_ptr,
                                                 self._gh_floats, self.adam[slot]["t_dev"].data_ptr(), self.stream()), "step_begin_chunks")
            self._img_live_flags = flags
        else:
            check(self.lib.nsc_step_begin(self.p_ptr, self.wt_idx.data_ptr(), self.wt_ptr, self.wt.numel(), self.g_ptr, self._gh_floats,
                                          self.adam[slot]["t_dev"].data_ptr(), self.stream()), "step_begin")
            self._img_live_flags = None
            if self.live_gather and self._rec is None and not torch.cuda.is_current_stream_capturing():
                self._wt_rec = set()                         # this run records what it is pointed at
        self.images_valid = True
        self.forward(x, cfg["is_quan_on"], True, lpc_x=lpc_x, hists_clean=True,
                     first_needed=min([i for i, t in enumerate(cfg["trainable"]) if t], default=self.N))
        gb = self.B * (comm.world if comm else 1)
        terms = self.loss_backward(target, cfg["c_time"], cfg["c_freq"], cfg["c_quan"], cfg["c_ent"], cfg["trainable"],
                                   c_quan_lpc=cfg.get("c_quan_lpc", 0.0), c_ent_lpc=cfg.get("c_ent_lpc", 0.0),
                                   global_batch=gb, train_lpc=cfg.get("train_lpc"),
                                   # the decision must not depend on a host value that can differ between ranks (tau):
                                   # every quan-op step exchanges the (tiny) histograms, no-quan steps never do
                                   hist_allreduce=(comm.allreduce_list if comm and cfg.get("global_entropy", True) and
                                                   cfg.get("quan_op", cfg.get("slot", 1) == 1) else None),
                                   # SUM, not mean: the reference's vector loss sums over the batch (a18)
                                   grad_allreduce=comm.allreduce_async if comm else None)
        scopes = [f"scope_{i + 1}" for i, t in enumerate(cfg["trainable"]) if t]
        train_lpc = cfg.get("train_lpc")
        if train_lpc is None:
            train_lpc = cfg.get("c_quan_lpc", 0.0) != 0.0 or cfg.get("c_ent_lpc", 0.0) != 0.0
        if self.lpc and train_lpc:
            scopes = ["lpc_quan"] + scopes
        self.adam_step(scopes, cfg["lr"], slot, counted=True)
        if self._wt_rec is not None:
            self._live_tables[key] = self._chunk_table(self._wt_rec)
            self._wt_rec = None
        return terms
