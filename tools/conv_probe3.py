"""PROBES build: stride-2 down-sampling conv and its polyphase data gradient with phases off (NSC_CONV_SKIP: 1 staging, 2 MFMA loop)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for skip in ("0", "1", "2", "3"):
    subprocess.run([sys.executable, os.path.join(here, "conv_probe2.py"), "run"], env=dict(os.environ, NSC_CONV_SKIP=skip, NSC_CONV_NC="0"))
