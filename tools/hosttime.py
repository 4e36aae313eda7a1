import sys, time, torch
sys.path.insert(0, '.')
import bench
from nsc_amd.engine import CascadeEngine
dev = torch.device('cuda', 0)
eng = CascadeEngine(128, 2, bench.BKD, [[2],[2]], [32,32], res_scalar=1.0, scale_first=True, lpc=True, device=dev)
xd, lpcd, _, _ = bench.synth_batch(128, 0, dev)
cfg = bench.step_cfg()
for _ in range(3): eng.train_step(xd, xd, cfg, lpc_x=lpcd)
torch.cuda.synchronize()
# host-only cost: enqueue while the GPU is blocked behind a long sleep-like kernel is hard; instead time enqueue of 5 steps (GPU runs behind)
t0 = time.perf_counter()
for _ in range(10): eng.train_step(xd, xd, cfg, lpc_x=lpcd)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step (host), total {1e3*(t2-t0)/10:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): eng.train_step(xd, xd, cfg, lpc_x=lpcd)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
