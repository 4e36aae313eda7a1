"""Times nsc_quantize_fwd in the op-surface form (p materialised, 32 bins) at the config-5 batch; with the probes library the
environment selects the workgroup-per-frame kernel (NSC_QUANT_WG=1) or the grid of the wave-per-frame one (NSC_QWGRID)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsc_amd import _lib
if os.environ.get("NSC_PROBES_LIB", "1") == "1":
    _lib.LIB_PATH = os.path.join(ROOT, "nsc_amd", "libnsc_hip_probes.so")
lib = _lib.load()
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
Bq, L, nb = 4096, 256, 32
torch.manual_seed(0)
code = torch.tanh(torch.randn(Bq, L, 1, device=dev))
alpha = torch.tensor([-20.0], device=dev); bins = torch.linspace(-1, 1, nb, device=dev)
def run(soft, p, outq, qv, hist):
    _lib.check(lib.nsc_quantize_fwd(code.data_ptr(), alpha.data_ptr(), bins.data_ptr(), 1.0, soft, Bq, L, nb, p.data_ptr(), outq.data_ptr(),
                                    qv.data_ptr(), hist.data_ptr(), st), "q")
res = {}
ONLY = os.environ.get("NSC_QT_ONLY")      # run one variant only (for rocprofv3 --pmc passes: one kernel shape per process)
for variant, env in (("workgroup per frame", {"NSC_QUANT_WG": "1"}), ("wave per frame, grid 512 (shipped)", {}), ("wave per frame, grid 1024", {"NSC_QWGRID": "1024"}), ("wave per frame, 4 lanes/code, grid 512", {"NSC_QLPC": "4"}),
                     ("wave per frame, p stores dropped (compute only)", {"NSC_QUANT_NO_STORE": "1"}),
                     ("wave per frame, 4 lanes/code, stores dropped", {"NSC_QLPC": "4", "NSC_QUANT_NO_STORE": "1"}),
                     ("write-only probe, same geometry", {"NSC_QUANT_WRITE_ONLY": "1"}), ("write-only probe, nontemporal", {"NSC_QUANT_WRITE_ONLY": "2"}),
                     ("wave per frame, grid 512 (again)", {})):
    if ONLY and ONLY not in variant:
        continue
    for k in ("NSC_QUANT_WG", "NSC_QWGRID", "NSC_QUANT_WRITE_ONLY", "NSC_QLPC", "NSC_QUANT_NO_STORE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    outs = []
    for soft in (1, 0):
        p = torch.empty(Bq, L, nb, device=dev); outq = torch.empty_like(code); qv = torch.empty(Bq, device=dev); hist = torch.zeros(nb, device=dev)
        run(soft, p, outq, qv, hist); torch.cuda.synchronize()
        outs.append((p.clone(), outq.clone(), qv.clone(), hist.clone()))
    res[variant] = outs
    p = torch.empty(Bq, L, nb, device=dev); outq = torch.empty_like(code); qv = torch.empty(Bq, device=dev); hist = torch.zeros(nb, device=dev)
    for _ in range(5):
        run(1, p, outq, qv, hist)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(1, p, outq, qv, hist)
        e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / 20)
    us = sorted(ts)[len(ts) // 2]
    byts = Bq * L * (4 + 4 * nb + 4)
    print(f"{variant:44s}: {us:6.2f} us (min {min(ts):6.2f})  {byts / us / 1e6:5.2f} TB/s  = {byts / us / 1e6 / 8:.3f} of 8 TB/s", flush=True)
if ONLY:
    sys.exit(0)
base = res["workgroup per frame"]
for k, v in res.items():
    if "probe" in k or "dropped" in k:
        continue
    d = [max(float((a - b).abs().max()) for a, b in zip(v[s][:3], base[s][:3])) for s in (0, 1)]
    dh = [float(((v[s][3] - base[s][3]).abs() / base[s][3].abs().clamp_min(1)).max()) for s in (0, 1)]
    print(f"  {k:30s} max |diff| vs workgroup kernel: soft {d[0]:.2e}, hard {d[1]:.2e}; histogram rel {dh[0]:.1e} / {dh[1]:.1e}")
