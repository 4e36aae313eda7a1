"""PROBES build: the k55 1 -> 100 input conv (CIN1 kernel, weights preloaded) with phases off (NSC_CONV_SKIP: 1 staging, 2 MFMAs)."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from nsc_amd import _lib
    from nsc_amd._lib import ConvDesc
    _lib.LIB_PATH = os.path.abspath(os.environ.get("NSC_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nsc_amd", "libnsc_hip_probes.so")))
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B, T = 128, 512
    x = torch.randn(B, 1, T, device="cuda"); w = torch.randn(55, 1, 100, device="cuda") * 0.1; bias = torch.randn(100, device="cuda")
    y = torch.empty(B, 100, T, device="cuda")
    d = ConvDesc(B=B, Cin=1, Cout=100, Tin=T, Tout=T, K=55, dil=1, stride=1, padL=27, act=2, res_mode=0, mul_mode=0, out_mode=0, in_up=0, accumulate=0)
    f = lambda: lib.nsc_conv1d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, None, y.data_ptr(), st)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print("skip=%s nopre=%s: %.1f us" % (os.environ.get("NSC_CONV_SKIP", "0"), os.environ.get("NSC_CONV_NOPRE", "-"), e0.elapsed_time(e1) * 20))
else:
    for env in ({}, {"NSC_CONV_SKIP": "1"}, {"NSC_CONV_SKIP": "2"}, {"NSC_CONV_SKIP": "3"}, {"NSC_CONV_NOPRE": "1"}):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, **env))
