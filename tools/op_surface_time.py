"""The op-surface leg of bench.py alone (a codec step built from nn_core_operator ops beside the engine's config-2 step), with the
gated blocks fused (ops.BlockFn) and composed op by op.   python tools/op_surface_time.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nsc_amd import nn_core_operator as nn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
_, _, x_np, _ = bench.synth_batch(B, 0, dev)
for fused in (True, False):
    nn.FUSED_BLOCKS = fused
    print("fused blocks" if fused else "composed blocks", json.dumps(bench.op_surface_leg(B, x_np, dev)))
