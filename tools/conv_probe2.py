"""PROBES build: the stride-2 down-sampling conv (k9 100->100, T 512->256) and its polyphase data gradient, with the time
tile forced to 64 (NSC_CONV_NC=1) or 128 steps (=2)."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from nsc_amd import _lib
    from nsc_amd._lib import ConvDesc
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    def timeit(d, x, w, y, n=30):
        f = lambda: lib.nsc_conv1d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), None, None, None, y.data_ptr(), st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    B = 128
    x = torch.randn(B, 100, 512, device="cuda"); w = torch.randn(9, 100, 100, device="cuda") * 0.05; y = torch.empty(B, 100, 256, device="cuda")
    d = ConvDesc(B=B, Cin=100, Cout=100, Tin=512, Tout=256, K=9, dil=1, stride=2, padL=3, act=2, res_mode=0, mul_mode=0, out_mode=0, in_up=0, accumulate=0)
    print("NC=%s stride-2 fwd  %.1f us" % (os.environ.get("NSC_CONV_NC", "auto"), timeit(d, x, w, y)))
    dz = torch.randn(B, 100, 256, device="cuda"); wp = torch.randn(5, 100, 200, device="cuda") * 0.05; dx = torch.empty(B, 100, 512, device="cuda")
    d2 = ConvDesc(B=B, Cin=100, Cout=200, Tin=256, Tout=256, K=5, dil=1, stride=1, padL=2, act=0, res_mode=0, mul_mode=0, out_mode=1, in_up=0, accumulate=0)
    print("NC=%s polyphase dgrad %.1f us" % (os.environ.get("NSC_CONV_NC", "auto"), timeit(d2, dz, wp, dx)))
else:
    for nc in ("0", "1", "2"):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, NSC_CONV_NC=nc))
