"""PMC calibration for 4-byte-per-lane coalesced access (the width every kernel of this library uses): nsc_axpby over
128 Mi floats reads 2 x 512 MiB and writes 512 MiB per launch (buffers far larger than the 256 MiB Infinity Cache).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/pmc_traffic.py divides the
counters by the known byte counts to get the correction factors it applies to the training-step kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
lib = _lib.load()
n = 128 * 1024 * 1024
x = torch.ones(n, device="cuda"); y = torch.ones(n, device="cuda"); out = torch.empty(n, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    _lib.check(lib.nsc_axpby(x.data_ptr(), y.data_ptr(), out.data_ptr(), 1.0, 1.0, n, st), "axpby")
torch.cuda.synchronize()
print("axpby", n, "floats: read", 2 * 4 * n, "B, write", 4 * n, "B per launch")
