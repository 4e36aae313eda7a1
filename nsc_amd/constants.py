"""Constants of the hot path (values of the reference's constants.py; cited by line)."""
import numpy as np

VERBOSE = False
init_alpha = -300            # constants.py:5
beta_boundary = 1            # constants.py:8
sample_rate = 16000          # constants.py:9
is_pure_time_domain = True   # constants.py:12 (edit-the-source switch in the reference; a flag here)
resnet_type = 'gln'          # constants.py:14
max_amp_tr = 33.461480140686035 if is_pure_time_domain else 22.307652973859113   # constants.py:15-18
mu_law_transform = False     # constants.py:21
frame_length = 512           # constants.py:25
overlap_each_side = 32       # constants.py:26
training_data_size = 500000  # constants.py:27
selected_ind = [8.0, 16.0, 32.0, 128.0]   # constants.py:28

# constants.py:66-119: the 256-value LSF quantisation table (16 LSF dimensions x 16 uniformly spaced levels).
# Kept as a data file (nsc_amd/lsf_bins_256.json); the pinned copy lives in tests/golden/reference_kats.json.


def _load_lsf_table():
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "lsf_bins_256.json")) as f:
        return [float(v) for v in json.load(f)]


lpc_coeff_lsf_bins = _load_lsf_table()
