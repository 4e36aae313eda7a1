"""Times nsc_conv1d_wgrad_batch on subsets of the headline step's per-conv weight-gradient jobs (class <1,7>: the k55 1 -> 100 input
convs and the pointwise 100 -> 100 convs of the up-sampling stage; class <4,7>: the stride-2 k9 convs).  Probes library: NSC_CW_BUDGET
sets the workgroup budget of a class.  usage: python tools/wgrad_batch_time.py"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
from nsc_amd._lib import ConvDesc, ConvWgradJob
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = 128
keep = []


def job(Cin, Cout, T, K, s, padL):
    Tout = -(-T // s)
    d = ConvDesc(B=B, Cin=Cin, Cout=Cout, Tin=T, Tout=Tout, K=K, dil=1, stride=s, padL=padL, act=0, res_mode=0, mul_mode=0, out_mode=0, in_up=0,
                 accumulate=0)
    x = torch.randn(B, Cin, T, device="cuda"); dz = torch.randn(B, Cout, Tout, device="cuda")
    dw = torch.zeros(K, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
    keep.extend([x, dz, dw, db])
    return ConvWgradJob(d, x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), 0), 2.0 * B * Tout * K * Cin * Cout


inc, pw, down = (1, 100, 512, 55, 1, 27), (100, 100, 256, 1, 1, 0), (100, 100, 512, 9, 2, 3)
for name, spec in (("2 x in_conv", [inc, inc]), ("2 x pointwise", [pw, pw]), ("class <1,7> of the step (2 + 2)", [inc, inc, pw, pw]), ("1 x in_conv", [inc]),
                   ("1 x pointwise", [pw]), ("2 x stride-2 k9 (class <4,7>)", [down, down])):
    js = [job(*sp) for sp in spec]
    jobs = (ConvWgradJob * len(js))(*[j for j, _ in js])
    fl = sum(f for _, f in js)
    need = int(lib.nsc_conv1d_wgrad_batch_workspace(jobs, len(js)))
    ws = torch.empty(need, device="cuda")
    run = lambda: _lib.check(lib.nsc_conv1d_wgrad_batch(jobs, len(js), ws.data_ptr(), need, st), "cw")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f"{name:36s} {us:7.1f} us (kernel + reduce)  {fl / us / 1e6:5.1f} TFLOP/s   workspace {need * 4 / 1e6:.1f} MB")
