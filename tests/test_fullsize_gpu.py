"""Size-independent properties at BASELINE.json's full size (config 3: 2-codec CMRL on the fed LPC residual + LSF quantizer,
B = 128 per GPU), where the float64 oracle is too slow to be the checker: determinism of the forward, independence of a
frame's result from the batch it is computed in, linearity of the backward in the loss coefficients, idempotence of the
hard quantizer, and additivity of the gradient over a split of the batch (what the data-parallel SUM all-reduce relies on)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 128


def _setup(b=B):
    import bench
    from nsc_amd.engine import CascadeEngine
    dev = torch.device("cuda", 0)
    eng = CascadeEngine(b, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True, device=dev)
    x, lpc, _, _ = bench.synth_batch(B, 0, dev)
    return bench, eng, x, lpc


def test_forward_is_deterministic_and_batch_independent():
    bench, eng, x, lpc = _setup()
    d1 = eng.forward(x, 1.0, True, lpc_x=lpc).clone()
    d2 = eng.forward(x, 1.0, True, lpc_x=lpc).clone()
    assert torch.equal(d1, d2)                                    # bit-identical run to run
    assert bool(torch.isfinite(d1).all())
    # the same frames through a 64-frame engine: every frame's arithmetic is its own (no cross-frame term in the forward)
    from nsc_amd.engine import CascadeEngine
    e64 = CascadeEngine(64, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True,
                        device=x.device)
    e64.params.copy_(eng.params)
    dh = e64.forward(x[:64].contiguous(), 1.0, True, lpc_x=lpc[:64].contiguous())
    assert torch.equal(dh, d1[:64])


def test_backward_is_linear_in_the_loss_coefficients():
    bench, eng, x, lpc = _setup()
    eng.refresh_wt()

    def grads(scale):
        eng.grads.zero_()
        eng.forward(x, 1.0, True, lpc_x=lpc)
        eng.loss_backward(x, scale * 60.0, scale * 10.0, [scale * 10.0] * 2, [0.0, 0.0], [True, True], c_quan_lpc=scale * 10.0)
        torch.cuda.synchronize()
        return eng.grads.clone()

    g1, g2 = grads(1.0), grads(2.0)
    scale = float(g1.abs().max())
    assert scale > 0 and bool(torch.isfinite(g1).all())
    assert float((g2 - 2.0 * g1).abs().max()) <= 2e-5 * scale     # fp32 reduction order only (slabs / atomics)


def test_gradient_is_additive_over_a_split_of_the_batch():
    """grad(B = 128) == grad(frames 0..63) + grad(frames 64..127) when no batch-global (entropy) term is on: the identity
    behind the data-parallel SUM all-reduce, at the full per-GPU batch."""
    bench, eng, x, lpc = _setup()
    from nsc_amd.engine import CascadeEngine
    eng.refresh_wt()
    eng.grads.zero_()
    eng.forward(x, 1.0, True, lpc_x=lpc)
    eng.loss_backward(x, 60.0, 10.0, [10.0, 10.0], [0.0, 0.0], [True, True], c_quan_lpc=10.0)
    torch.cuda.synchronize()
    full = eng.grads.clone()
    e64 = CascadeEngine(64, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True,
                        device=x.device)
    e64.params.copy_(eng.params)
    e64.refresh_wt()
    acc = torch.zeros_like(full)
    for lo in (0, 64):
        xs, ls = x[lo:lo + 64].contiguous(), lpc[lo:lo + 64].contiguous()
        e64.grads.zero_()
        e64.forward(xs, 1.0, True, lpc_x=ls)
        e64.loss_backward(xs, 60.0, 10.0, [10.0, 10.0], [0.0, 0.0], [True, True], c_quan_lpc=10.0)
        torch.cuda.synchronize()
        acc += e64.grads
    scale = float(full.abs().max())
    assert float((acc - full).abs().max()) <= 5e-5 * scale


def test_hard_quantizer_is_idempotent_at_full_size():
    """Quantising already-quantised codes returns them unchanged (bit-exact), 128 x 256 codes on the 32 trained-shape bins."""
    import ctypes as C
    from nsc_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    L, nb = 256, 32
    code = torch.tensor(np.tanh(rng.standard_normal((B, L, 1))).astype(np.float32), device=dev)
    alpha = torch.tensor([-300.0], device=dev)
    bins = torch.linspace(-1, 1, nb, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    q1, q2 = torch.empty_like(code), torch.empty_like(code)
    quan = torch.empty(B, device=dev); hist = torch.zeros(nb, device=dev)
    for src, dst in ((code, q1), (q1, q2)):
        assert lib.nsc_quantize_fwd(src.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, 0, B, L, nb, None, dst.data_ptr(),
                                    quan.data_ptr(), hist.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(q1, q2)
    assert bool(torch.isin(q1.flatten(), bins).all())             # every output is exactly one of the bins
    # alpha = -300: the soft-to-hard quantizer IS nearest-bin rounding (no code is further than half a bin from its output)
    assert float((q1 - code).abs().max()) <= 0.5 * float(bins[1] - bins[0]) + 1e-6
