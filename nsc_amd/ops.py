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


def _capture_id():
    """Identity of the hipGraph capture the current stream records into (0: none)."""
    if not torch.cuda.is_current_stream_capturing():
        return 0
    cid = C.c_ulonglong(0)
    check(_lib_().nsc_stream_capture_id(_st(), C.byref(cid)), "stream_capture_id")
    return int(cid.value)


ZERO_POOL_FLOATS = 1 << 20


def _zeros(n, dev):
    """n zeroed floats (16-byte aligned) for a gradient the kernels ACCUMULATE into: a slice of a pool that ONE fill launch zeroed,
    instead of one fill per gradient (21 fills per step of the codec before).  A slice is handed out once, never recycled; a pool is
    dropped when it is used up (its memory lives as long as the gradients cut from it).  A pool belongs to one (stream, capture): a
    graph being captured gets its own pool - and with it its own fill node - on first use."""
    n4 = (int(n) + 3) // 4 * 4
    owner = (int(_st()), _capture_id())
    key = ("zero_pool", str(dev))
    st = _CACHE.get(key)
    if st is None or st[2] != owner or st[1] + n4 > st[0].numel():
        st = [torch.zeros(max(ZERO_POOL_FLOATS, n4), dtype=torch.float32, device=dev), 0, owner]
        _CACHE[key] = st
    out = st[0][st[1]:st[1] + int(n)]
    st[1] += n4
    return out


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


def _flip_index(K, Cin, Cout, off):
    """Source offsets of the flipped / transposed kernel wt [K][Cout][Cin] of a conv kernel w [K][Cin][Cout] at `off`:
    wt[k', o, i] = w[K - 1 - k', i, o] (what nsc_gated_block_flip_weights / nsc_weight_flip_transpose write)."""
    kt, o, i = np.meshgrid(np.arange(K), np.arange(Cout), np.arange(Cin), indexing="ij")
    return (((K - 1 - kt) * Cin + i) * Cout + o).reshape(-1).astype(np.int64) + int(off)


class _ImageSet:
    """Everything the kernels DERIVE from the parameters of one scope.VariableStore arena - kernel-ready images of the gated blocks
    (forward + data gradient), of the stride-2 convs, flipped / transposed kernels of the other convs - rebuilt by ONE nsc_gather
    launch per pass, the engine's arrangement (engine.py: wt_idx / nsc_step_begin; a gather launch is latency: 16 us for the 0.1 M
    words of one block, 25 us for the 1.2 M words of a whole codec).  An item is registered the first time an op asks for it (that request, and any
    other before the next pass, is served by a gather of its own); from the next store.begin_pass() on, the first request of a pass
    gathers the whole set into a fresh buffer and every request is a slice of it.  A graph capture counts as a pass of its own
    (nsc_stream_capture_id), so a captured step always contains its gather."""

    def __init__(self, store, arena):
        import weakref
        self.store, self.arena = weakref.ref(store), arena
        self.items, self.parts, self.total = {}, [], 0
        self.idx_all, self.n_all, self.total_all = None, 0, 0
        self.token, self.buf = None, None

    @staticmethod
    def _versions(refs):
        return tuple(-1 if t is None else t._version for t in (r() for r in refs))

    def get(self, key, make_index, tensors):
        import weakref
        lib, dev, base = _lib_(), self.arena.device, self.arena.data_ptr()
        it = self.items.get(key)
        if it is None:
            idx = np.ascontiguousarray(make_index(), np.int32)
            it = self.items[key] = [self.total, idx.size, torch.from_numpy(idx).to(dev), [weakref.ref(t) for t in tensors], None]
            self.parts.append(idx)
            self.total += (idx.size + 3) // 4 * 4
        token = (self.store().pass_id, _capture_id())
        # a new pass - or parameters of THIS item changed in place since the gather that served it (an optimizer step between two uses
        # without store.begin_pass(): every in-place update moves the tensor's version counter; assignments through .data do not)
        if token != self.token or (it[4] is not None and it[4] != self._versions(it[3])):
            if self.n_all != len(self.items) and not token[1]:
                flat = np.full(self.total, -1, np.int32)
                o = 0
                for part in self.parts:
                    flat[o:o + part.size] = part
                    o += (part.size + 3) // 4 * 4
                self.idx_all, self.n_all, self.total_all = torch.from_numpy(flat).to(dev), len(self.items), self.total
            self.buf = None
            if self.idx_all is not None:
                self.buf = torch.empty(self.total_all, dtype=torch.float32, device=dev)
                check(lib.nsc_gather(base, self.idx_all.data_ptr(), self.buf.data_ptr(), self.total_all, _st()), "gather (image set)")
            for other in self.items.values():
                other[4] = self._versions(other[3]) if (self.buf is not None and other[0] + other[1] <= self.buf.numel()) else None
            self.token = token
        start, n, dev_idx, refs, _ = it
        if self.buf is not None and start + n <= self.buf.numel():
            return self.buf[start:start + n]
        img = torch.empty(n, dtype=torch.float32, device=dev)
        check(lib.nsc_gather(base, dev_idx.data_ptr(), img.data_ptr(), n, _st()), "gather (image)")
        return img


def _image_set(tensors):
    """(the _ImageSet of the store arena that holds all of `tensors`, their word offsets in it), or None."""
    from .scope import find_arena
    ptrs = [t.data_ptr() for t in tensors]
    found = find_arena(min(ptrs), max(q + 4 * t.numel() for q, t in zip(ptrs, tensors)))
    if found is None:
        return None
    store, arena = found
    base = arena.data_ptr()
    if any((q - base) % 4 for q in ptrs):
        return None
    iset = store.image_sets.get(base)
    if iset is None:
        iset = store.image_sets[base] = _ImageSet(store, arena)
    return iset, tuple((q - base) // 4 for q in ptrs)


def _flipped_kernel(lib, w):
    """wt [K][Cout][Cin] = the flipped / transposed copy of a conv kernel (what tf.gradients' conv backprop reads): a slice of the
    store's image set when w is a store variable, else one nsc_weight_flip_transpose launch."""
    K, Cin, Cout = w.shape
    found = _image_set([w])
    if found is not None:
        iset, (off,) = found
        return iset.get(("flip", K, Cin, Cout, off), lambda: _flip_index(K, Cin, Cout, off), [w])
    wt = torch.empty((K, Cout, Cin), dtype=torch.float32, device=w.device)
    check(lib.nsc_weight_flip_transpose(w.data_ptr(), wt.data_ptr(), K, Cin, Cout, _st()), "flip")
    return wt


def _split_conv_image(lib, which, d, w):
    """Kernel-ready image of a stride-2 k9 100 -> 100 kernel for the split-operand conv kernels (csrc/conv_split.hip): a slice of the
    store's image set when w is a store variable, else one gather launch from the kernel tensor itself (index map cached per device).
    None: this shape is not served (or NSC_BLOCK_ARITH=exact)."""
    if not SPLIT_ARITH:
        return None
    n = int(lib.nsc_conv1d_simage_words(which, C.byref(d)))
    if n <= 0 or w.data_ptr() % 16:
        return None

    def index(off):
        idx = np.empty(n, np.int32)
        check(lib.nsc_conv1d_simage_index(which, C.byref(d), off, idx.ctypes.data_as(C.c_void_p)), "conv1d_simage_index")
        return idx
    found = _image_set([w])
    if found is not None:
        iset, (off,) = found
        return iset.get(("conv", which, tuple(w.shape), off), lambda: index(off), [w])
    key = ("cs_idx", which, str(w.device), tuple(w.shape))
    if key not in _CACHE:
        _CACHE[key] = torch.from_numpy(index(0)).to(w.device)
    img = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.nsc_gather(w.data_ptr(), _CACHE[key].data_ptr(), img.data_ptr(), n, _st()), "gather")
    return img


# Parameter gradients (gated blocks AND convs): deferred to the END of the backward pass and produced by a few batched launches
# (nsc_gated_block_wgrad_batch[_split], nsc_conv1d_wgrad_batch, nsc_sum_all_batch) instead of one launch + one slab reduction per
# op - the engine's arrangement (engine.py: "block weight gradients deferred to the end of the backward pass"; the reference's
# tf.gradients leaves the order of weight gradients open, cmrl.py:106-113).  An op's backward returns views of a ZEROED buffer as
# its parameter gradients and queues the job; a callback at the end of the autograd pass (the hook DistributedDataParallel
# finalises with) launches the batches, which ACCUMULATE into those views on the same stream.  autograd may keep a view as .grad
# (nothing to do), or have copied / summed the zeros into something else (the filled view is added to it afterwards).  Not deferred:
# parameters that are not leaves (their gradient is consumed inside the pass), and everything when DEFER_WGRAD is False.  Known
# limit: torch.autograd.grad() on a parameter SHARED by two deferring ops sees the sum formed before the batches ran - call
# backward() or set DEFER_WGRAD = False (NSC_SURFACE_DEFER_WGRAD=0).
DEFER_WGRAD = os.environ.get("NSC_SURFACE_DEFER_WGRAD", "1") == "1"
_PENDING = {}


def _deferrable(params):
    return DEFER_WGRAD and torch._C._current_graph_task_id() >= 0 and all(t.is_leaf for t in params)


def _defer(kind, group, job, keep, flat_g, params, sizes):
    """Queue one job of the batched launch `kind` ('block' | 'conv' | 'sum'); params / sizes: the parameters whose gradients are the
    consecutive slices of flat_g (no reference to the gradient views themselves is kept: autograd adopts a gradient as .grad without
    copying only when nobody else holds it)."""
    task = torch._C._current_graph_task_id()
    st = _PENDING.get(task)
    if st is None:
        if len(_PENDING) > 4:                     # passes that raised before their callback ran: their queues (and the tensors they hold) go
            for old in list(_PENDING)[:-4]:
                _PENDING.pop(old, None)
        st = _PENDING[task] = {}
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _flush_wgrads(task))
    offs = np.concatenate([[0], np.cumsum(sizes)]) if sizes else []
    info = [(prm, int(offs[i]), int(sizes[i]), prm.grad, prm.grad._version if prm.grad is not None else -1) for i, prm in enumerate(params)]
    st.setdefault((kind,) + tuple(group), []).append((job, keep, flat_g, info))


def _flush_wgrads(task):
    st = _PENDING.pop(task, None)
    if not st:
        return
    lib = _lib_()
    for key, grp in st.items():
        kind, dev = key[0], key[1]
        jobs = [g[0] for g in grp]
        if kind == "block":
            ws_ = _workspace(2 * int(lib.nsc_gated_block_wgrad_batch_workspace(112)), dev)
            fn = lib.nsc_gated_block_wgrad_batch_split if SPLIT_ARITH else lib.nsc_gated_block_wgrad_batch
            check(fn((_lib.BlockWgradJob * len(jobs))(*jobs), len(jobs), key[2], 20, 9, ws_.data_ptr(), ws_.numel(), _st()),
                  "gated_block_wgrad_batch")
        elif kind == "conv":
            arr = (_lib.ConvWgradJob * len(jobs))(*jobs)
            ws_ = _workspace(int(lib.nsc_conv1d_wgrad_batch_workspace(arr, len(jobs))), dev)
            check(lib.nsc_conv1d_wgrad_batch(arr, len(jobs), ws_.data_ptr(), ws_.numel(), _st()), "conv1d_wgrad_batch")
        else:
            for lo in range(0, len(jobs), 8):
                chunk = jobs[lo:lo + 8]
                check(lib.nsc_sum_all_batch((_lib.SumJob * len(chunk))(*chunk), len(chunk), _st()), "sum_all_batch")
    for grp in st.values():
        for _, _, flat_g, info in grp:
            for prm, off, n, g0, v0 in info:
                g = prm.grad
                if g is None or (g is g0 and g._version == v0):
                    continue                      # autograd left .grad alone (torch.autograd.grad): the caller holds the view itself
                if g.data_ptr() != flat_g.data_ptr() + 4 * off:
                    g.add_(flat_g[off:off + n].view(g.shape))   # autograd copied or summed the zeros elsewhere (existing .grad, shared parameter)


class Conv1dFn(torch.autograd.Function):
    """tf.compat.v1.layers.conv1d(padding='SAME', channels_last) + bias + activation.  The stride-2 down-sampling conv takes the
    split-operand kernels of the engine (forward, data gradient, weight gradient) like the gated blocks do."""

    @staticmethod
    def forward(ctx, x, w, b, dil, stride, act):
        lib = _lib_()
        params = (w, b)
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
        ctx.params = params
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
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]        # (a frozen conv: the data path only)
        dw = db = None
        dfw = _desc(B, Cin, Cout, T, Tout, K, dil, stride, padL)
        split = (stride == 2 and SPLIT_ARITH and int(lib.nsc_conv1d_simage_words(0, C.byref(dfw))) > 0 and
                 (xb.data_ptr() | dz.data_ptr() | w.data_ptr()) % 16 == 0)
        if need_w:
            dwb = _zeros(w.numel() + Cout, w.device)                                        # dw | db
            dw, db = dwb[:w.numel()].view(w.shape), dwb[w.numel():]
            # partial sums of the (b,t) splits go to private slabs + one reduce launch (nsc_conv1d_wgrad_ws) instead of same-address atomics
            if split:
                ws = _workspace(int(lib.nsc_conv1d_wgrad_split_workspace()), w.device)
                job = _lib.ConvWgradJob(dfw, xb.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0)
                check(lib.nsc_conv1d_wgrad_split((_lib.ConvWgradJob * 1)(job), 1, ws.data_ptr(), ws.numel(), _st()), "wgrad (split)")
            elif Cout == 1:      # roles swapped: "input" = dz, "gradient" = x, flipped taps; the bias gradient is the plain sum of dz
                d = _desc(B, 1, Cin, Tout, T, K, dil, 1, (K - 1) * dil - padL)
                if _deferrable(ctx.params):
                    _defer("conv", (w.device,), _lib.ConvWgradJob(d, dz.data_ptr(), xb.data_ptr(), dw.data_ptr(), None, 1), (xb, dz), dwb,
                           ctx.params, [w.numel(), Cout])
                    _defer("sum", (w.device,), _lib.SumJob(dz.data_ptr(), db.data_ptr(), dz.numel()), (dz, dwb), dwb, (), [])
                else:
                    ws = _workspace(int(lib.nsc_conv1d_wgrad_workspace(C.byref(d))), w.device)
                    check(lib.nsc_conv1d_wgrad_ws(C.byref(d), dz.data_ptr(), xb.data_ptr(), dw.data_ptr(), None, 1, ws.data_ptr(), ws.numel(),
                                                  _st()), "wgrad")
                    check(lib.nsc_sum_all(dz.data_ptr(), db.data_ptr(), dz.numel(), _st()), "bias grad")
            else:
                d = _desc(B, Cin, Cout, T, Tout, K, dil, stride, padL)
                if _deferrable(ctx.params):
                    _defer("conv", (w.device,), _lib.ConvWgradJob(d, xb.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0), (xb, dz),
                           dwb, ctx.params, [w.numel(), Cout])
                else:
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
            wt = _flipped_kernel(lib, w)
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
        dwd = _zeros(wd.numel(), wd.device).view(wd.shape)
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


class UpsampleFn(torch.autograd.Function):
    """_up_sampling_mod (neural_speech_coding_module.py:168-181: SeparableConv1D(C, 9) -> leaky-relu -> sub-pixel shuffle) as ONE
    autograd node on the engine's fused kernels: nsc_upsample_fwd (depthwise -> pointwise -> activation -> shuffle in one launch) and
    nsc_upsample_bwd (un-shuffle -> pointwise^T -> depthwise^T); the pointwise kernel's gradient joins the deferred conv batch.
    C in {100, 50}, 9 taps, stride 2 (nn_core_operator.conv1d_depth_shuffle decides)."""

    @staticmethod
    def forward(ctx, x, wd, wp, b, act):
        lib = _lib_()
        params = (wd, wp, b)
        xb = to_bct(_req(x, "inputs"))
        wd, wp, b = _req(wd).contiguous(), _req(wp).contiguous(), _req(b).contiguous()
        B, C_, T = xb.shape
        dwo = torch.empty_like(xb)
        y = torch.empty((B, C_ // 2, 2 * T), dtype=torch.float32, device=xb.device)
        check(lib.nsc_upsample_fwd(xb.data_ptr(), wd.data_ptr(), wp.data_ptr(), b.data_ptr(), dwo.data_ptr(), y.data_ptr(), B, C_, T, 9,
                                   ACT[act], _st()), "upsample_fwd")
        ctx.save_for_backward(xb, dwo, y, wd, wp)
        ctx.act, ctx.params = act, params
        return to_btc(y)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_()
        xb, dwo, y, wd, wp = ctx.saved_tensors
        B, C_, T = xb.shape
        dev = xb.device
        dz = to_bct(_req(dy, "grad"))
        if ACT[ctx.act]:
            dz = _act_bwd(lib, dz, y, ctx.act)
        dzp, ddw, dx = (torch.empty_like(xb) for _ in range(3))
        check(lib.nsc_upsample_bwd(dz.data_ptr(), wd.data_ptr(), wp.data_ptr(), dzp.data_ptr(), ddw.data_ptr(), dx.data_ptr(), B, C_, T, 9,
                                   _st()), "upsample_bwd")
        # the depthwise taps' gradient at once (its own small kernel); the pointwise kernel's with the deferred conv batch
        dwd = _zeros(wd.numel(), dev).view(ctx.params[0].shape)
        check(lib.nsc_depthwise_bwd(xb.data_ptr(), wd.data_ptr(), ddw.data_ptr(), None, dwd.data_ptr(), B, C_, T, 9, _st()), "depthwise wgrad")
        sizes = [wp.numel(), C_]
        flat_g = _zeros(sum(sizes), dev)
        dwp, db = (gr.view(t.shape) for gr, t in zip(flat_g.split(sizes), ctx.params[1:]))
        d = _desc(B, C_, C_, T, T, 1, 1, 1, 0)
        if _deferrable(ctx.params[1:]):
            _defer("conv", (dev,), _lib.ConvWgradJob(d, dwo.data_ptr(), dzp.data_ptr(), dwp.data_ptr(), db.data_ptr(), 0), (dwo, dzp), flat_g,
                   ctx.params[1:], sizes)
        else:
            ws = _workspace(int(lib.nsc_conv1d_wgrad_workspace(C.byref(d))), dev)
            check(lib.nsc_conv1d_wgrad_ws(C.byref(d), dwo.data_ptr(), dzp.data_ptr(), dwp.data_ptr(), db.data_ptr(), 0, ws.data_ptr(),
                                          ws.numel(), _st()), "wgrad")
        return to_btc(dx), dwd, dwp, db, None


def _block_image_meta(lib, C_, Cin, dil):
    """(words of the forward part incl. padding, forward kind 'split' | 'exact', words of the data-gradient part) of a gated block's
    image pair, or None when the shape has no image kernels."""
    key = ("blk_meta", C_, Cin, dil, SPLIT_ARITH)
    if key not in _CACHE:
        nb = int(lib.nsc_gated_block_image_floats(1, C_, Cin, dil))
        ns = int(lib.nsc_gated_block_simage_words(0, C_, Cin, dil)) if SPLIT_ARITH else 0
        ne = int(lib.nsc_gated_block_image_floats(0, C_, Cin, dil))
        _CACHE[key] = None if (nb <= 0 or (ns <= 0 and ne <= 0)) else (((ns if ns > 0 else ne) + 3) // 4 * 4, "split" if ns > 0 else "exact", nb)
    return _CACHE[key]


def _block_image_index(lib, C_, Cin, dil, offs):
    """nsc_gather index map of BOTH kernel-ready images of a gated block from memory that holds its eight parameters at word offsets
    `offs` (w1, b1, wl, bl, wr, br, w9, b9): the forward image (split operands where the shape is served and NSC_BLOCK_ARITH is not
    'exact', else the exact kernel's image) followed, 16-byte aligned, by the exact data-gradient image (defined on the flipped /
    transposed kernels: composed with the flip here)."""
    nf, kind, nb = _block_image_meta(lib, C_, Cin, dil)
    offs8 = (C.c_long * 8)(*offs)
    idx = np.full(nf + nb, -1, np.int32)
    if kind == "split":
        n = int(lib.nsc_gated_block_simage_words(0, C_, Cin, dil))
        f = np.empty(n, np.int32)
        check(lib.nsc_gated_block_simage_index(0, C_, Cin, dil, offs8, f.ctypes.data_as(C.c_void_p)), "gated_block_simage_index")
    else:
        n = int(lib.nsc_gated_block_image_floats(0, C_, Cin, dil))
        f = np.empty(n, np.int32)
        check(lib.nsc_gated_block_image_index(0, C_, Cin, dil, offs8, f.ctypes.data_as(C.c_void_p)), "gated_block_image_index")
    idx[:n] = f
    n1, n15 = Cin * 20, 15 * 20 * 20
    flip = np.concatenate([_flip_index(1, Cin, 20, offs[0]), _flip_index(15, 20, 20, offs[2]), _flip_index(15, 20, 20, offs[4]),
                           _flip_index(9, 20, C_, offs[6])])
    bw = np.empty(nb, np.int32)
    check(lib.nsc_gated_block_image_index(1, C_, Cin, dil, (C.c_long * 4)(0, n1, n1 + n15, n1 + 2 * n15),
                                          bw.ctypes.data_as(C.c_void_p)), "gated_block_image_index")
    idx[nf:] = np.where(bw >= 0, flip[np.maximum(bw, 0)], -1)
    return idx


def _block_images(lib, ws, C_, Cin, dil):
    """(image pair, words of the forward part, forward kind) of this block: a slice of the store's image set when its eight parameters
    are store variables (one gather launch per pass for ALL blocks), else ONE nsc_gather launch from wherever the eight tensors lie
    (any float32 tensors within 2^25 words of each other; index map cached per relative placement).  The forward hands the pair to
    the backward through its ctx.  None: not served this way (the pointer entry points take over)."""
    meta = _block_image_meta(lib, C_, Cin, dil)
    if meta is None:
        return None
    found = _image_set(ws)
    if found is not None:
        iset, offs = found
        if max(offs) < (1 << 25):
            return iset.get(("blk", meta[1], C_, Cin, dil, offs), lambda: _block_image_index(lib, C_, Cin, dil, offs), ws), meta[0], meta[1]
    ptrs = [t.data_ptr() for t in ws]
    base = min(ptrs)
    offs = tuple((q - base) // 4 for q in ptrs)
    if any((q - base) % 4 for q in ptrs) or max(offs) >= (1 << 25):
        return None
    dev = ws[0].device
    key = ("blk_idx", meta[1], str(dev), C_, Cin, dil, offs)
    if key not in _CACHE:
        _CACHE[key] = torch.from_numpy(_block_image_index(lib, C_, Cin, dil, offs)).to(dev)
    idx = _CACHE[key]
    img = torch.empty(idx.numel(), dtype=torch.float32, device=dev)
    check(lib.nsc_gather(base, idx.data_ptr(), img.data_ptr(), idx.numel(), _st()), "gather (block images)")
    return img, meta[0], meta[1]


def _block_forward(lib, xb, ws, dil, flat):
    """One gated block forward on [B,Cin,T] memory: (out, h, lin, th, g, data-gradient image | None)."""
    B, Cin, T = xb.shape
    C_ = ws[6].shape[2]
    out = torch.empty((B, C_, T), dtype=torch.float32, device=xb.device)
    h, lin, th, g = torch.empty((4, B, 20, T), dtype=torch.float32, device=xb.device).unbind(0)
    sv = [t.data_ptr() for t in (h, lin, th, g)]
    imgs = _block_images(lib, ws, C_, Cin, dil) if (C_ in (100, 50, 25) and Cin in (C_, 1)) else None
    if imgs is not None and imgs[2] == "split" and T % 4 == 0 and xb.data_ptr() % 16 == 0:
        check(lib.nsc_gated_block_fwd_simg(imgs[0].data_ptr(), xb.data_ptr(), out.data_ptr(), *sv, B, C_, Cin, T, dil, int(flat), _st()),
              "gated_block_fwd_simg")
    elif imgs is not None and imgs[2] == "exact":
        check(lib.nsc_gated_block_fwd_img(imgs[0].data_ptr(), xb.data_ptr(), out.data_ptr(), *sv, B, C_, Cin, T, dil, int(flat), _st()),
              "gated_block_fwd_img")
    else:
        fn = lib.nsc_gated_block_fwd_cin1 if Cin == 1 else lib.nsc_gated_block_fwd
        check(fn(xb.data_ptr(), *[t.data_ptr() for t in ws], out.data_ptr(), *sv, B, C_, T, 20, 9, dil, int(flat), _st()), "gated_block_fwd")
    # (the image's second part, as a view: it keeps the image alive until the backward has run)
    return out, h, lin, th, g, (None if imgs is None else imgs[0][imgs[1]:])


def _block_backward(lib, xb, h, lin, th, g, ws, img_bwd, dz, dil, in_act, params, need_w=True):
    """One gated block backward from dz = dL/d(pre-activation of its output): (dx [B,Cin,T], the eight parameter gradients).  in_act:
    the activation that PRODUCED the block's input (its derivative is applied to dx in the kernel's epilogue: dx is then dL/d of that
    pre-activation - how a chain of blocks skips the activation-backward launches); parameter gradients deferred to the end of the pass
    where possible (see DEFER_WGRAD)."""
    w1, b1, wl, bl, wr, br, w9, b9 = ws
    B, Cin, T = xb.shape
    C_ = w9.shape[2]
    dev = xb.device
    n1, n15, n9 = Cin * 20, 15 * 20 * 20, 9 * 20 * C_
    dx = torch.empty((B, Cin, T), dtype=torch.float32, device=dev)
    da = torch.empty((B, 40, T), dtype=torch.float32, device=dev)
    dz1 = torch.empty((B, 20, T), dtype=torch.float32, device=dev)
    if img_bwd is not None:
        check(lib.nsc_gated_block_dgrad_img(img_bwd.data_ptr(), None if Cin == 1 else xb.data_ptr(), h.data_ptr(), lin.data_ptr(),
                                            th.data_ptr(), dz.data_ptr(), dx.data_ptr(), da.data_ptr(), da.data_ptr() + 4 * 20 * T,
                                            dz1.data_ptr(), B, C_, Cin, T, dil, ACT[in_act], 40, _st()), "gated_block_dgrad_img")
    else:
        # flipped / transposed kernels of the four convs (what tf.gradients' conv backprop reads): one launch, one buffer
        wt = torch.empty(n1 + 2 * n15 + n9, dtype=torch.float32, device=dev)
        check(lib.nsc_gated_block_flip_weights(w1.data_ptr(), wl.data_ptr(), wr.data_ptr(), w9.data_ptr(), wt.data_ptr(), C_, Cin, 20,
                                               9, _st()), "flip")
        p0 = wt.data_ptr()
        wts = [p0, p0 + 4 * n1, p0 + 4 * (n1 + n15), p0 + 4 * (n1 + 2 * n15)]
        if Cin == 1:
            assert ACT[in_act] == 0
            check(lib.nsc_gated_block_dgrad_cin1(h.data_ptr(), lin.data_ptr(), th.data_ptr(), dz.data_ptr(), *wts,
                                                 dx.data_ptr(), da.data_ptr(), da.data_ptr() + 4 * 20 * T, dz1.data_ptr(), B, C_, T, 20,
                                                 9, dil, 40, _st()), "gated_block_dgrad_cin1")
        else:
            check(lib.nsc_gated_block_dgrad(xb.data_ptr(), h.data_ptr(), lin.data_ptr(), th.data_ptr(), dz.data_ptr(),
                                            *wts, dx.data_ptr(), da.data_ptr(), dz1.data_ptr(), B, C_, T, 20, 9,
                                            dil, ACT[in_act], _st()), "gated_block_dgrad")
    if not need_w:                                     # a frozen block (an earlier codec in a follower phase): data path only
        return dx, [None] * 8
    # parameter gradients: the batched launch (it serves both block forms); the eight gradients are one contiguous range in
    # creation order (a slice of the zero pool: the kernels accumulate), handed back as views of it
    sizes = [Cin * 20, 20, n15, 20, n15, 20, n9, C_]
    flat_g = _zeros(sum(sizes), dev)
    job = _lib.BlockWgradJob(xb.data_ptr(), h.data_ptr(), g.data_ptr(), dz.data_ptr(), da.data_ptr(), dz1.data_ptr(),
                             flat_g.data_ptr(), C_, T, dil, Cin)
    grads = [gr.view(t.shape) for gr, t in zip(flat_g.split(sizes), ws)]
    if _deferrable(params):
        _defer("block", (dev, B), job, (xb, h, g, dz, da, dz1), flat_g, params, sizes)
    else:
        ws_ = _workspace(int(lib.nsc_gated_block_wgrad_batch_workspace(112)), dev)
        fn = lib.nsc_gated_block_wgrad_batch_split if SPLIT_ARITH else lib.nsc_gated_block_wgrad_batch
        check(fn((_lib.BlockWgradJob * 1)(job), 1, B, 20, 9, ws_.data_ptr(), ws_.numel(), _st()), "gated_block_wgrad_batch")
    return dx, grads


def _act_bwd(lib, dz, out, act):
    dz2 = torch.empty_like(dz)
    check(lib.nsc_act_bwd(dz.data_ptr(), out.data_ptr(), dz2.data_ptr(), dz.numel(), ACT[act], _st()), "act_bwd")
    return dz2


class BlockFn(torch.autograd.Function):
    """gated_bottleneck (nn_core_operator.py:82-112) as ONE function call, like the reference's: the fused persistent kernels of the
    engine instead of four convs + multiply + add + activations - forward on the block's kernel-ready image (bf16 matrix cores on
    split operands by default, `nsc_gated_block_fwd_simg`; NSC_BLOCK_ARITH=exact: the fp32 instruction), data gradient
    `nsc_gated_block_dgrad_img`, parameter gradients batched at the end of the pass (see DEFER_WGRAD).  narrow 20, k9 9,
    dilation 1 | 2, wide <= 112 (one input channel: wide in {100, 50, 25}); shapes without an image kernel run the pointer entry
    points nsc_gated_block_fwd[_cin1] / nsc_gated_block_dgrad[_cin1]; other shapes keep the composed form
    (nn_core_operator.gated_bottleneck decides)."""

    @staticmethod
    def forward(ctx, x, w1, b1, wl, bl, wr, br, w9, b9, dil, flat):
        lib = _lib_()
        xb = to_bct(_req(x, "the_input"))
        ws = [_req(t).contiguous() for t in (w1, b1, wl, bl, wr, br, w9, b9)]
        out, h, lin, th, g, img_bwd = _block_forward(lib, xb, ws, int(dil), bool(flat))
        ctx.save_for_backward(xb, h, lin, th, g, out, *ws)
        ctx.cfg = (int(dil), bool(flat))
        ctx.img_bwd = img_bwd
        ctx.params = (w1, b1, wl, bl, wr, br, w9, b9)
        return to_btc(out)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_()
        xb, h, lin, th, g, out, *ws = ctx.saved_tensors
        dil, flat = ctx.cfg
        dz = to_bct(_req(dy, "grad"))
        if not flat:                                   # through the block's output leaky-relu
            dz = _act_bwd(lib, dz, out, "lrelu")
        dx, grads = _block_backward(lib, xb, h, lin, th, g, ws, ctx.img_bwd, dz, dil, None, ctx.params, any(ctx.needs_input_grad[1:9]))
        return (to_btc(dx), *grads, None, None)


class BlockStackFn(torch.autograd.Function):
    """_stack_bottleneck_blocks (neural_speech_coding_module.py:183-217: n gated blocks in a row, leaky-relu between them, the last one
    flat or not) as ONE autograd node: the same launches as n BlockFn calls, except that the backward of block i + 1 applies the
    derivative of the leaky-relu between the blocks in its own epilogue (the kernels' in_act: what the engine's chain does) - n - 1
    activation-backward launches and 2 (n - 1) autograd nodes less.  Arguments: x, the 8 n parameters in creation order, the n dilation
    rates, is_last_flat."""

    @staticmethod
    def forward(ctx, x, *args):
        lib = _lib_()
        dils, flat_last = tuple(int(d) for d in args[-2]), bool(args[-1])
        params = args[:-2]
        n = len(dils)
        assert len(params) == 8 * n
        cur = to_bct(_req(x, "the_input"))
        saved, imgs = [cur], []
        for i in range(n):
            ws = [_req(t).contiguous() for t in params[8 * i:8 * i + 8]]
            out, h, lin, th, g, img_bwd = _block_forward(lib, cur, ws, dils[i], flat_last if i == n - 1 else False)
            saved += [h, lin, th, g, out, *ws]
            imgs.append(img_bwd)
            cur = out
        ctx.save_for_backward(*saved)
        ctx.cfg, ctx.imgs, ctx.params = (dils, flat_last), imgs, params
        return to_btc(cur)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib_()
        dils, flat_last = ctx.cfg
        n = len(dils)
        sv = ctx.saved_tensors
        blk = [sv[1 + 13 * i:1 + 13 * (i + 1)] for i in range(n)]          # h, lin, th, g, out, 8 parameters
        dz = to_bct(_req(dy, "grad"))
        if not flat_last:
            dz = _act_bwd(lib, dz, blk[n - 1][4], "lrelu")
        grads = [None] * n
        for i in reversed(range(n)):
            h, lin, th, g, _, *ws = blk[i]
            xin = sv[0] if i == 0 else blk[i - 1][4]                        # the block's input: x, or the previous block's output
            dz, grads[i] = _block_backward(lib, xin, h, lin, th, g, ws, ctx.imgs[i], dz, dils[i], "lrelu" if i > 0 else None,
                                           ctx.params[8 * i:8 * i + 8], any(ctx.needs_input_grad[1 + 8 * i:9 + 8 * i]))
        return (to_btc(dz), *[t for gs in grads for t in gs], None, None)


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
        dab = _zeros(4 + bins.numel(), code.device)
        dalpha, dbins = dab[:1], dab[4:].view(bins.shape)
        dp = _req(dp).contiguous() if dp is not None else None
        dout = _req(dout).contiguous() if dout is not None else None
        check(_lib_().nsc_quantize_bwd(code.data_ptr(), alpha.data_ptr(), bins.data_ptr(), on, soft, B, L, nb,
                                       _lib.ptr(dout), _lib.ptr(dp), 0.0, None, 0.0, 0, dcode.data_ptr(), dalpha.data_ptr(),
                                       dbins.data_ptr(), _st()), "quantize_bwd")
        return dcode, dalpha.reshape(()), dbins, None, None


class ReconLossFn(torch.autograd.Function):
    """(mse_loss, mfcc_loss) of decoded vs original, both [B,512] -> two [B] vectors; gradient wrt decoded only.  When a gradient
    will be asked for, the ONE forward launch also leaves d mfcc_loss / d decoded behind (nsc_recon_loss_banded with unit
    coefficients); the backward is then the elementwise nsc_recon_loss_combine instead of a second pass through the FFTs and the
    mel banks."""

    @staticmethod
    def forward(ctx, decoded, original):
        from .loss_terms_and_measures import mel_matrix_cat, mel_band_ranges
        decoded, original = _req(decoded).contiguous(), _req(original).contiguous()
        B = decoded.shape[0]
        dev = decoded.device
        key = ("mel", str(dev))
        if key not in _CACHE:
            m = mel_matrix_cat()
            _CACHE[key] = (torch.from_numpy(m).to(dev), torch.from_numpy(np.ascontiguousarray(m.T)).to(dev),
                           torch.from_numpy(mel_band_ranges(m)).to(dev))
        mel, melT, ranges = _CACHE[key]
        t, f = torch.empty(B, device=dev), torch.empty(B, device=dev)
        gfreq = torch.empty_like(decoded) if ctx.needs_input_grad[0] else None
        check(_lib_().nsc_recon_loss_banded(decoded.data_ptr(), original.data_ptr(), B, 0.0, 1.0, None, None, mel.data_ptr(),
                                            melT.data_ptr(), ranges.data_ptr(), t.data_ptr(), f.data_ptr(), _lib.ptr(gfreq), _st()),
              "recon_loss")
        ctx.save_for_backward(decoded, original, t, gfreq)
        return t, f

    @staticmethod
    def backward(ctx, gt, gf):
        _LAST.pop("recon", None)                  # (a node that has run its backward is not handed out again: see _shared)
        decoded, original, t, gfreq = ctx.saved_tensors
        B = decoded.shape[0]
        gt = _req(gt).contiguous() if gt is not None else None
        gf = _req(gf).contiguous() if gf is not None else None
        g = torch.empty_like(decoded)
        check(_lib_().nsc_recon_loss_combine(decoded.data_ptr(), original.data_ptr(), t.data_ptr(), _lib.ptr(gt), _lib.ptr(gf),
                                             gfreq.data_ptr(), B, g.data_ptr(), _st()), "recon_loss_combine")
        return g, None


_CACHE = {}


class PStatsFn(torch.autograd.Function):
    """quan_loss AND entropy_coding_loss of one soft assignment p [B,L,nb] (loss_terms_and_measures.py:257-267): the per-frame
    sum_k sqrt(p) term and the batch histogram come out of ONE pass over p (nsc_p_stats), the entropy of the histogram follows
    (nsc_entropy_from_hist), and ONE backward pass forms dL/dp from both upstream gradients (nsc_p_stats_bwd) - the two losses share
    this node when they are asked of the same tensor (see _shared)."""

    @staticmethod
    def forward(ctx, p):
        p = _req(p).contiguous()
        B, L, nb = p.shape
        lib = _lib_()
        q = torch.empty(B, device=p.device)
        z = _zeros(nb, p.device)
        hist = z[:nb]
        ent = torch.empty(1, device=p.device)
        gh = torch.empty(nb, device=p.device)
        check(lib.nsc_p_stats(p.data_ptr(), B, L, nb, q.data_ptr(), hist.data_ptr(), _st()), "p_stats")
        check(lib.nsc_entropy_from_hist(hist.data_ptr(), nb, ent.data_ptr(), gh.data_ptr(), _st()), "entropy")
        ctx.save_for_backward(p, gh)
        return q, ent.reshape(())

    @staticmethod
    def backward(ctx, gq, ge):
        _LAST.pop("p_stats", None)                # (a node that has run its backward is not handed out again)
        p, gh = ctx.saved_tensors
        B, L, nb = p.shape
        lib = _lib_()
        ghs = None
        if ge is not None:
            ghs = torch.empty_like(gh)
            check(lib.nsc_mul(gh.data_ptr(), _req(ge.reshape(1)).expand(nb).contiguous().data_ptr(), ghs.data_ptr(), nb, _st()), "mul")
        dp = torch.empty_like(p)
        check(lib.nsc_p_stats_bwd(p.data_ptr(), _lib.ptr(None if gq is None else _req(gq).contiguous()), _lib.ptr(ghs), dp.data_ptr(),
                                  B, L, nb, _st()), "p_stats_bwd")
        return dp


_LAST = {}


def _shared(name, args, make):
    """make(*args), or the result of the previous call when it was made for the SAME tensor objects at the same versions under the
    same grad mode: two losses of one tensor (mse_loss + mfcc_loss of `decoded`, quan_loss + entropy_coding_loss of `p`) then share one
    autograd node - one forward launch, one backward launch, no gradient sum in between - as the engine's fused loss kernels do.  Object
    identity, not addresses: a new tensor in recycled memory is a new key."""
    import weakref
    ent = _LAST.get(name)
    key = tuple((id(a), a._version) for a in args) + (torch.is_grad_enabled(), _capture_id())
    if ent is not None and ent[0] == key and all(r() is a for r, a in zip(ent[1], args)):
        return ent[2]
    out = make(*args)
    _LAST[name] = (key, [weakref.ref(a) for a in args], out)
    return out


# ---- functional wrappers used by nn_core_operator / loss_terms_and_measures ----
def recon_losses(decoded, original):
    return _shared("recon", (decoded, original), lambda d, o: ReconLossFn.apply(d.reshape(-1, 512), o.reshape(-1, 512)))


def rfft512(sig):
    sig = _req(sig).contiguous().reshape(-1, 512)
    B = sig.shape[0]
    re, im, mag = (torch.empty((B, 257), device=sig.device) for _ in range(3))
    check(_lib_().nsc_rfft512(sig.data_ptr(), B, re.data_ptr(), im.data_ptr(), mag.data_ptr(), _st()), "rfft512")
    return torch.complex(re, im), mag


def quan_loss(p):
    return _shared("p_stats", (p,), PStatsFn.apply)[0]


def entropy_coding_loss(p):
    return _shared("p_stats", (p,), PStatsFn.apply)[1]
