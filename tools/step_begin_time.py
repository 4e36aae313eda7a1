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
# ... and the chunked form over the regions a joint step reads (round 6): every region the default arm touches, from the engine's registry
import numpy as np
x, lpc, _, _ = bench.synth_batch(128, 0, dev)
cfg = dict(is_quan_on=1.0, c_time=60.0, c_freq=10.0, c_quan=[10.0, 10.0], c_ent=[0.0, 0.0], trainable=[True, True], lr=0.0, slot=1,
           c_quan_lpc=10.0, train_lpc=True, quan_op=True)
eng.train_step(x, x, cfg, lpc_x=lpc)
torch.cuda.synchronize()
(tab, n), = eng._live_tables.values()
words = int(tab[:, 1].sum().item())
run2 = lambda: check(eng.lib.nsc_step_begin_chunks(eng.p_ptr, eng.wt_idx.data_ptr(), eng.wt_ptr, tab.data_ptr(), n, eng.g_ptr, eng._gh_floats,
                                                   eng.adam[1]["t_dev"].data_ptr(), st), "step_begin_chunks")
for _ in range(10):
    run2()
e0.record()
for _ in range(200):
    run2()
e1.record()
torch.cuda.synchronize()
print(f"step_begin_chunks: {words} gathered words in {n} chunks: {1e3 * e0.elapsed_time(e1) / 200:.1f} us per launch (back to back)")
# cold caches between launches (as in a step, where 2 ms of other kernels run in between): a big copy in between
junk = torch.empty(64 << 20, device=dev)
for name, fn in (("step_begin", run), ("step_begin_chunks", run2)):
    tot = 0.0
    for _ in range(20):
        junk.add_(1.0)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"{name} after a 256 MB sweep: {1e3 * tot / 20:.1f} us")
