"""PROBES build: forward up-sampling kernel with phases switched off (NSC_UP_SKIP bits: 1 x loads, 2 depthwise, 4 MFMAs, 8 y stores, 16 weight loads)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for skip in (0, 1, 2, 4, 8, 16, 17, 6, 31):
    print("skip", skip, flush=True)
    subprocess.run([sys.executable, os.path.join(here, "up_probe.py")], env=dict(os.environ, NSC_UP_SKIP=str(skip)))
