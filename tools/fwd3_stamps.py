"""Phase stamps of gated_block_fwd3_kernel (probes library): per round and role, cycles of
   [a-work | wait beta1 | stage | b-MFMA | b-epilogue | wait beta2] for workgroup 0, wave 0 (P) and wave 4 (C)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "nsc_amd", "libnsc_hip_probes.so")
lib = _lib.load()
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr() if t is not None else None
for (B, Cc, T, dil) in [(128, 100, 512, 2), (4096, 100, 256, 2), (128, 100, 256, 1)]:
    x = torch.randn(B, Cc, T, device=dev)
    w1 = torch.randn(1, Cc, 20, device=dev) * 0.1; b1 = torch.zeros(20, device=dev)
    wl = torch.randn(15, 20, 20, device=dev) * 0.05; wr = torch.randn(15, 20, 20, device=dev) * 0.05
    bl = torch.zeros(20, device=dev); br = torch.zeros(20, device=dev)
    w9 = torch.randn(9, 20, Cc, device=dev) * 0.05; b9 = torch.zeros(Cc, device=dev)
    out = torch.empty_like(x)
    sv = [torch.empty(B, 20, T, device=dev) for _ in range(4)] if B <= 128 else [None] * 4
    def run():
        _lib.check(lib.nsc_gated_block_fwd(p(x), p(w1), p(b1), p(wl), p(bl), p(wr), p(br), p(w9), p(b9), p(out), p(sv[0]), p(sv[1]),
                                           p(sv[2]), p(sv[3]), B, Cc, T, 20, 9, dil, 0, st), "blk")
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    lib.nsc_probe_read3.argtypes = [C.c_void_p]
    assert lib.nsc_probe_read3(buf) == 0
    s = list(buf)
    print(f"--- B={B} C={Cc} T={T} dil={dil}: cycles  P [a-work, wait b1, stage, b-mfma, b-epi, wait b2] | C [copy-out, k9 a, wait b1, -, k9 b, ost]")
    nr = min(16, B * ((T + 63) // 64) // 256 + 1)
    for role, name in ((0, "P"), (1, "C")):
        for r in range(min(nr, 7)):
            v = s[128 * role + 8 * r:128 * role + 8 * r + 7]
            d = [v[i + 1] - v[i] for i in range(6)]
            extra = f"  (stage-only {s[128 * role + 8 * r + 7] - v[2]})" if role == 0 else ""
            print(f"  {name} round {r}: {d}  total {v[6] - v[0]}{extra}")
