"""PROBES build only: per-phase s_memtime stamps of workgroup 0 / wave 0 of gated_block_fwd2 (last tile)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nsc_amd", "libnsc_hip_probes.so")
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
lib = _lib.load()
lib.nsc_probe_read.argtypes = [C.c_void_p]
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
names = ["start", "prologue done", "tile start", "staged", "bar0", "ph1 done", "bar1", "ph2 done", "bar2", "ph3 done", "bar3", "end"]
for (B, C_, T, dil) in [(128, 100, 512, 1), (128, 50, 512, 2), (1024, 100, 512, 2)]:
    x = torch.randn(B, C_, T, device=dev)
    w1 = torch.randn(1, C_, 20, device=dev) * 0.1; b1 = torch.randn(20, device=dev) * 0.1
    wl = torch.randn(15, 20, 20, device=dev) * 0.05; bl = torch.randn(20, device=dev) * 0.1
    wr = torch.randn(15, 20, 20, device=dev) * 0.05; br = torch.randn(20, device=dev) * 0.1
    w9 = torch.randn(9, 20, C_, device=dev) * 0.05; b9 = torch.randn(C_, device=dev) * 0.1
    out = torch.empty_like(x)
    h, lin, th, g = (torch.empty(B, 20, T, device=dev) for _ in range(4))
    p = lambda t: t.data_ptr()
    for _ in range(5):
        _lib.check(lib.nsc_gated_block_fwd(p(x), p(w1), p(b1), p(wl), p(bl), p(wr), p(br), p(w9), p(b9), p(out), p(h), p(lin), p(th), p(g),
                                           B, C_, T, 20, 9, dil, 0, st), "fwd")
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert lib.nsc_probe_read(buf) == 0
    v, w = list(buf)[32:44], list(buf)[96:108]
    print(f"B={B} C={C_} T={T} dil={dil}: total {v[11]-v[0]} cycles   (wave 0 | wave 4)")
    for i in range(1, 12):
        print(f"  {names[i]:>14}: +{v[i]-v[i-1]:6d} | +{w[i]-w[i-1]:6d}")
