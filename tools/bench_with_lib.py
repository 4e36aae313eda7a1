"""bench.py on another build of the library: NSC_LIB=<path to .so> python tools/bench_with_lib.py [bench flags]  (experiment builds: make exp EXP=n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsc_amd import _lib
if os.environ.get("NSC_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NSC_LIB"])
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
