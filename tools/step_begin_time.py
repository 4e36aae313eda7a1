"""nsc_step_begin alone on the headline engine's index map (the launch that opens a step: gather of flipped kernels + parameter images,
zeroing, step counter).   NSC_LIB_PATH=<build> python tools/step_begin_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsc_amd.engine import CascadeEngine
from nsc_amd._lib import check
dev = torch.device("cuda", 0)
eng = CascadeEngine(128, 2, bench.BKD, [[2], [2]], [32, 32], res_scalar=bench.RES_SCALAR, scale_first=True, lpc=True, device=dev)
st = torch.cuda.current_stream().cuda_stream
run = lambda: check(eng.lib.nsc_step_begin(eng.p_ptr, eng.wt_idx.data_ptr(), eng.wt_ptr, eng.wt.numel(), eng.g_ptr, eng._gh_floats,
                                           eng.adam[1]["t_dev"].data_ptr(), st), "step_begin")
for _ in range(10):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record()
torch.cuda.synchronize()
print(f"step_begin: {eng.wt.numel()} gathered words, {eng._gh_floats} zeroed floats: {1e3 * e0.elapsed_time(e1) / 200:.1f} us per launch (back to back)")
