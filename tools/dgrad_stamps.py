"""Probes library (`make -C nsc_amd/csrc probes`): per-phase s_memtime stamps of workgroup 0, waves 0 and 4, of gated_block_dgrad2 (last tile)."""
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
names = ["start", "prologue done", "tile start", "staged", "bar0", "D9 done", "bar1", "GLU done", "bar2", "D15 done", "bar3",
         "dz1 done", "bar4", "D1 done", "bar5", "copyout done", "bar6", "end"]
for (B, C_, T, dil) in [(128, 100, 512, 1), (128, 100, 256, 2), (1024, 100, 512, 2)]:
    x = torch.randn(B, C_, T, device=dev); dy = torch.randn(B, C_, T, device=dev)
    h, lin, th = (torch.randn(B, 20, T, device=dev) for _ in range(3)); th = torch.tanh(th)
    wt1 = torch.randn(1, 20, C_, device=dev) * 0.1
    wtl = torch.randn(15, 20, 20, device=dev) * 0.05; wtr = torch.randn(15, 20, 20, device=dev) * 0.05
    wt9 = torch.randn(9, C_, 20, device=dev) * 0.05
    dx = torch.empty_like(x); da = torch.empty(B, 40, T, device=dev); dz1 = torch.empty(B, 20, T, device=dev)
    p = lambda t: t.data_ptr()
    for _ in range(5):
        _lib.check(lib.nsc_gated_block_dgrad(p(x), p(h), p(lin), p(th), p(dy), p(wt1), p(wtl), p(wtr), p(wt9), p(dx), p(da), p(dz1),
                                             B, C_, T, 20, 9, dil, 2, st), "dgrad")
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert lib.nsc_probe_read(buf) == 0
    v, w = list(buf)[:18], list(buf)[64:82]
    print(f"B={B} C={C_} T={T} dil={dil}: total {v[17]-v[0]} ticks   (wave 0 | wave 4)")
    for i in range(1, 18):
        print(f"  {names[i]:>14}: +{v[i]-v[i-1]:6d} | +{w[i]-w[i-1]:6d}")
    a, c = list(buf)[:20], list(buf)[64:84]
    print(f"  inside 'dz1 done': next tile's lin/tanh/h requested +{a[18]-a[10]} | +{c[18]-c[10]}, halo carried +{a[19]-a[18]} | +{c[19]-c[18]}, "
          f"dz1 +{a[11]-a[19]} | +{c[11]-c[19]}")
