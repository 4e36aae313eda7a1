"""PROBES build only: time one conv shape with NSC_CONV_SKIP phases switched off (1 staging, 2 MFMA loop, 4 epilogue).
usage: conv_probe.py  (spawns itself once per skip value)"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from nsc_amd import _lib
    from nsc_amd._lib import ConvDesc
    lib = _lib.load()
    B, Ci, Co, T = 128, 100, 100, 256
    x = torch.randn(B, Ci, T, device="cuda"); w = torch.randn(1, Ci, Co, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
    y = torch.empty(B, Co // 2, 2 * T, device="cuda")
    d = ConvDesc(B=B, Cin=Ci, Cout=Co, Tin=T, Tout=T, K=1, dil=1, stride=1, padL=0, act=2, res_mode=0, mul_mode=0, out_mode=1,
                 in_up=0, accumulate=0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        lib.nsc_conv1d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.nsc_conv1d_fwd(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), None, None, y.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    print(f"skip={os.environ.get('NSC_CONV_SKIP', '0')}: {e0.elapsed_time(e1) * 20:.1f} us per launch")
else:
    for skip in (0, 1, 2, 4, 3, 6, 7):
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, NSC_CONV_SKIP=str(skip)))
