"""GPU versions of the reference's lpc_utilities.py functions that run inside tf.py_func.

Same names and argument order as the reference (file:line cited); arrays are float32 CUDA tensors instead of NumPy
arrays, every frame of the batch is processed by one kernel launch instead of a Python loop.  LPC *analysis* of raw
audio (`lpc_analysis_at_train/_at_test`: audiolazy.lpc + spectrum.poly2lsf) stays out of scope: the training data
carries precomputed LSFs (nsc_module:45-55), exactly as in the reference's training loop.
"""
from __future__ import annotations

import torch

from . import _lib
from .constants import frame_length


def _st():
    return torch.cuda.current_stream().cuda_stream


def lsf2poly_after_quan(lpc_in_lsf, order):
    """lpc_utilities.py:28-33: [B, order] quantised LSFs -> [B, order+1] prediction polynomial (float32)."""
    lib = _lib.load()
    x = lpc_in_lsf.reshape(-1, int(order)).contiguous().float()
    out = torch.empty((x.shape[0], int(order) + 1), dtype=torch.float32, device=x.device)
    _lib.check(lib.nsc_lsf2poly(x.data_ptr(), out.data_ptr(), x.shape[0], int(order), _st()), "lsf2poly")
    return out


def lpc_analysis_get_residual(raw_data_one_batch, quan_lpc_coeff):
    """lpc_utilities.py:37-77: [B,512,1] (or [B,512]) frames + [B,17] polynomial -> [B,512] residual (float32)."""
    lib = _lib.load()
    x = raw_data_one_batch.reshape(-1, frame_length).contiguous().float()
    a = quan_lpc_coeff.contiguous().float()
    out = torch.empty_like(x)
    _lib.check(lib.nsc_lpc_residual(x.data_ptr(), a.data_ptr(), out.data_ptr(), x.shape[0], a.shape[1] - 1, _st()),
               "lpc_residual")
    return out


def lpc_synthesizer_tr(lpc_coeff, lpc_res):
    """lpc_utilities.py:137-156: [B,17] polynomial + [B,512] residual -> [B,512] synthesised frames (float32).
    The reference wraps it in tf.custom_gradient with an identity gradient; it only feeds evaluation."""
    lib = _lib.load()
    a = lpc_coeff.contiguous().float()
    r = lpc_res.reshape(-1, frame_length).contiguous().float()
    out = torch.empty_like(r)
    _lib.check(lib.nsc_lpc_synthesis(a.data_ptr(), r.data_ptr(), out.data_ptr(), r.shape[0], a.shape[1] - 1, _st()),
               "lpc_synthesis")
    return out


def quantize_lsf_hard(lpc_x, alpha, bins):
    """The LSF quantizer at evaluation / refresh time (nsc_module:1088-1098: is_quan_on = 1.0, the_share False):
    [B,16,1] -> [B,16] hard-quantised LSFs."""
    lib = _lib.load()
    x = lpc_x.reshape(lpc_x.shape[0], -1, 1).contiguous().float()
    B, L = x.shape[0], x.shape[1]
    out = torch.empty_like(x)
    _lib.check(lib.nsc_quantize_fwd(x.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, 0, B, L, int(bins.numel()), None,
                                    out.data_ptr(), None, None, _st()), "lsf quantize")
    return out[:, :, 0]


def residual_from_lsf(frames, lsf, alpha, bins, chunk=50000):
    """The periodic residual refresh of the reference (`_update_lpc_residual`, nsc_module:1075-1121): hard-quantise the
    stored LSFs with the CURRENT (sorted) LSF codebook, rebuild A(z), re-filter the raw frames - in chunks of 50000
    frames like the reference.  frames [N,512], lsf [N,16] (CUDA or CPU tensors) -> residual [N,512] on the GPU."""
    dev = alpha.device
    sb = torch.sort(bins.reshape(-1))[0].contiguous()
    outs = []
    for lo in range(0, frames.shape[0], chunk):
        f = frames[lo:lo + chunk].to(dev)
        w = lsf[lo:lo + chunk].to(dev)
        q = quantize_lsf_hard(w.reshape(w.shape[0], -1, 1), alpha, sb)
        outs.append(lpc_analysis_get_residual(f, lsf2poly_after_quan(q, w.shape[1])))
    return torch.cat(outs, 0)
