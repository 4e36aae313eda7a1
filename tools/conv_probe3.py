"""PROBES build: stride-2 down-sampling conv and its polyphase data gradient, 32x32x2 kernel vs the 16x16x4 one (NSC_CONV_NO_M32)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for env in ({}, {"NSC_CONV_NO_M32": "1"}):
    print(env, flush=True)
    subprocess.run([sys.executable, os.path.join(here, "conv_probe2.py"), "run"], env=dict(os.environ, NSC_CONV_NC="0", **env))
