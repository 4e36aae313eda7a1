"""One short bench run with the Cin=1 batched weight gradients on/off (argv[1] = 1|0), for a kernel-trace timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsc_amd.engine as E
E.CascadeEngine.batch_cin1_wgrad = bool(int(sys.argv[1]))
import bench
sys.argv = ["bench.py", "--steps", "10", "--warmup", "4", "--no-cpu-baseline", "--no-infer", "--prof-steps", "1"]
bench.main()
