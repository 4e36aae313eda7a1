"""One short eager bench run for a kernel-trace timeline (tools/timeline.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = ["bench.py", "--steps", "10", "--warmup", "4", "--no-cpu-baseline", "--no-infer", "--prof-steps", "1", "--no-graph"] + sys.argv[1:]
bench.main()
