"""ctypes binding of libnsc_hip.so (the C ABI declared in include/nsc_hip.h).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.  The oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NSC_LIB_PATH: another build of the library (A/B runs of tools/ and tests; the shipped path is the in-tree libnsc_hip.so)
LIB_PATH = os.environ.get("NSC_LIB_PATH") or os.path.join(_HERE, "libnsc_hip.so")

ACT_NONE, ACT_TANH, ACT_LRELU = 0, 1, 2


class ConvDesc(C.Structure):
    """Mirror of ``nsc_conv_desc`` (include/nsc_hip.h)."""
    _fields_ = [(n, C.c_int) for n in
                ("B", "Cin", "Cout", "Tin", "Tout", "K", "dil", "stride", "padL", "act", "res_mode", "mul_mode",
                 "out_mode", "in_up", "accumulate")]


_P = C.c_void_p
_F = C.c_float
_I = C.c_int
_L = C.c_long

# name -> argtypes (all return int except the two noted)
PROTOTYPES = {
    "nsc_conv1d_fwd": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "nsc_conv1d_cout1_fwd": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P],
    "nsc_conv1d_wgrad": [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _P],
    "nsc_frame_entropy": [_P, _I, _I, _I, _P, _P],
    "nsc_conv1d_wgrad_ws": [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _P, _L, _P],
    "nsc_weight_flip_transpose": [_P, _P, _I, _I, _I, _P],
    "nsc_conv1d_simage_index": [_I, C.POINTER(ConvDesc), _L, _P],
    "nsc_conv1d_fwd_simg": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P],
    "nsc_conv1d_dgrad_simg": [C.POINTER(ConvDesc), _P, _P, _P, _P],
    "nsc_gated_block_flip_weights": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "nsc_gated_block_fwd": [_P] * 14 + [_I] * 7 + [_P],
    "nsc_gated_block_fwd_cin1": [_P] * 14 + [_I] * 7 + [_P],
    "nsc_gated_block_wgrad": [_P] * 16 + [_I] * 9 + [_P, _P],
    "nsc_glu_bwd_cat": [_P, _P, _P, _P, _I, _I, _I, _P],
    "nsc_gated_block_dgrad": [_P] * 12 + [_I] * 7 + [_P],
    "nsc_gated_block_dgrad_cin1": [_P] * 12 + [_I] * 7 + [_P],
    "nsc_gated_block_fwd_img": [_P] * 7 + [_I] * 6 + [_P],
    "nsc_gated_block_dgrad_img": [_P] * 10 + [_I] * 7 + [_P],
    "nsc_gated_block_image_index": [_I, _I, _I, _I, _P, _P],
    "nsc_gated_block_simage_index": [_I, _I, _I, _I, _P, _P],
    "nsc_gated_block_fwd_simg": [_P] * 7 + [_I] * 6 + [_P],
    "nsc_gated_block_dgrad_simg2": [_P] * 10 + [_I] * 6 + [_P],
    "nsc_gated_block_pair_fwd_simg": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "nsc_depthwise_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "nsc_depthwise_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "nsc_gate_fwd": [_P, _P, _I, _I, _I, _P],
    "nsc_gate_bwd": [_P, _P, _P, _I, _I, _I, _P],
    "nsc_mul": [_P, _P, _P, _L, _P],
    "nsc_glu_bwd": [_P, _P, _P, _P, _P, _L, _P],
    "nsc_gather": [_P, _P, _P, _L, _P],
    "nsc_act_fwd": [_P, _P, _L, _I, _P],
    "nsc_act_bwd": [_P, _P, _P, _L, _I, _P],
    "nsc_p_stats": [_P, _I, _I, _I, _P, _P, _P],
    "nsc_p_stats_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "nsc_axpby": [_P, _P, _P, _F, _F, _L, _P],
    "nsc_channel_sum": [_P, _P, _I, _I, _I, _I, _P],
    "nsc_unshuffle2": [_P, _P, _I, _I, _I, _P],
    "nsc_shuffle2": [_P, _P, _I, _I, _I, _P],
    "nsc_transpose_last2": [_P, _P, _I, _I, _I, _P],
    "nsc_sum_all": [_P, _P, _L, _P],
    "nsc_cascade_step": [_P, _P, _I, _P, _P, _F, _F, _L, _P],
    "nsc_upsample_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nsc_upsample_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "nsc_quantize_fwd": [_P, _P, _P, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nsc_entropy_from_hist": [_P, _I, _P, _P, _P],
    "nsc_quantize_bwd": [_P, _P, _P, _F, _I, _I, _I, _I, _P, _P, _F, _P, _F, _I, _P, _P, _P, _P],
    "nsc_recon_loss": [_P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsc_recon_loss_banded": [_P, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nsc_recon_loss_combine": [_P, _P, _P, _P, _P, _P, _I, _P, _P],
    "nsc_rfft512": [_P, _I, _P, _P, _P, _P],
    "nsc_adam_tf1_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _P, _P],
    "nsc_increment": [_P, _P],
    "nsc_spin": [_P, _I, _P],
    "nsc_gated_block_pair_fwd_img": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "nsc_gated_block_pair_dgrad_img": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "nsc_step_begin": [_P, _P, _P, _L, _P, _L, _P, _P],
    "nsc_step_begin_chunks": [_P, _P, _P, _P, _I, _P, _L, _P, _P],
    "nsc_frame_utterance": [_P, _L, _P, _P, _I, _P],
    "nsc_overlap_add": [_P, _I, _P, _P, _P],
    "nsc_lsf2poly": [_P, _P, _I, _I, _P],
    "nsc_lpc_residual": [_P, _P, _P, _I, _I, _P],
    "nsc_lpc_synthesis": [_P, _P, _P, _I, _I, _P],
    "nsc_zero": [_P, _L, _P],
    "nsc_stream_capture_id": [_P, _P],
}
class BlockWgradJob(C.Structure):
    """include/nsc_hip.h: struct nsc_block_wgrad_job"""
    _fields_ = [(n, C.c_void_p) for n in ("x", "h", "g", "dy", "da", "dz1", "grads")] + [(n, C.c_int) for n in ("C", "T", "dil", "Cin")]


class ConvWgradJob(C.Structure):
    """include/nsc_hip.h: struct nsc_conv_wgrad_job"""
    _fields_ = [("d", ConvDesc), ("x", C.c_void_p), ("dz", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p),
                ("flip_taps", C.c_int)]


class Cout1Chain(C.Structure):
    """include/nsc_hip.h: struct nsc_cout1_chain"""
    _fields_ = [("p_in", C.c_void_p), ("out2", C.c_void_p), ("pa", C.c_float), ("pb", C.c_float),
                ("q_in", C.c_void_p), ("out3", C.c_void_p), ("qa", C.c_float), ("qb", C.c_float)]


class Cout1Quant(C.Structure):
    """include/nsc_hip.h: struct nsc_cout1_quant"""
    _fields_ = [("alpha", C.c_void_p), ("bins", C.c_void_p), ("is_quan_on", C.c_float), ("soft", C.c_int), ("nb", C.c_int),
                ("qcode", C.c_void_p), ("quan", C.c_void_p), ("hist", C.c_void_p)]


class SumJob(C.Structure):
    """include/nsc_hip.h: struct nsc_sum_job"""
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("n", C.c_long)]


class EntropyJob(C.Structure):
    """include/nsc_hip.h: struct nsc_entropy_job"""
    _fields_ = [("hist", C.c_void_p), ("ent", C.c_void_p), ("ghist", C.c_void_p), ("nb", C.c_int)]


PROTOTYPES["nsc_conv1d_cout1_fwd_chain"] = [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.POINTER(Cout1Chain), _P]
PROTOTYPES["nsc_conv1d_cout1_fwd_quant"] = [C.POINTER(ConvDesc), _P, _P, _P, _P, C.POINTER(Cout1Quant), _P]
PROTOTYPES["nsc_sum_all_batch"] = [C.POINTER(SumJob), _I, _P]
PROTOTYPES["nsc_entropy_from_hist_batch"] = [C.POINTER(EntropyJob), _I, _P]
PROTOTYPES["nsc_gated_block_wgrad_batch"] = [C.POINTER(BlockWgradJob), _I, _I, _I, _I, _P, _L, _P]
PROTOTYPES["nsc_gated_block_wgrad_batch_split"] = [C.POINTER(BlockWgradJob), _I, _I, _I, _I, _P, _L, _P]
PROTOTYPES["nsc_conv1d_wgrad_batch"] = [C.POINTER(ConvWgradJob), _I, _P, _L, _P]
PROTOTYPES["nsc_conv1d_wgrad_split"] = [C.POINTER(ConvWgradJob), _I, _P, _L, _P]
EXPORTS = sorted(list(PROTOTYPES) + ["nsc_version", "nsc_last_error", "nsc_gated_block_wgrad_workspace", "nsc_gated_block_image_floats",
                  "nsc_gated_block_simage_words", "nsc_conv1d_simage_words", "nsc_conv1d_wgrad_split_workspace",
                  "nsc_gated_block_pair_flag_ints",
                  "nsc_conv1d_wgrad_workspace", "nsc_gated_block_wgrad_batch_workspace",
                  "nsc_conv1d_wgrad_batch_workspace"])


class NscError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NscError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C nsc_amd/csrc`. nsc_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argt in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = C.c_int
    lib.nsc_gated_block_wgrad_workspace.argtypes = [C.c_int]
    lib.nsc_gated_block_wgrad_workspace.restype = C.c_long
    lib.nsc_conv1d_wgrad_workspace.argtypes = [C.POINTER(ConvDesc)]
    lib.nsc_conv1d_wgrad_workspace.restype = C.c_long
    lib.nsc_gated_block_wgrad_batch_workspace.argtypes = [C.c_int]
    lib.nsc_gated_block_wgrad_batch_workspace.restype = C.c_long
    lib.nsc_conv1d_wgrad_batch_workspace.argtypes = [C.POINTER(ConvWgradJob), C.c_int]
    lib.nsc_conv1d_wgrad_batch_workspace.restype = C.c_long
    lib.nsc_gated_block_image_floats.argtypes = [C.c_int] * 4
    lib.nsc_gated_block_image_floats.restype = C.c_long
    lib.nsc_conv1d_wgrad_split_workspace.argtypes = []
    lib.nsc_conv1d_wgrad_split_workspace.restype = C.c_long
    lib.nsc_conv1d_simage_words.argtypes = [C.c_int, C.POINTER(ConvDesc)]
    lib.nsc_conv1d_simage_words.restype = C.c_long
    lib.nsc_gated_block_simage_words.argtypes = [C.c_int] * 4
    lib.nsc_gated_block_simage_words.restype = C.c_long
    lib.nsc_version.restype = C.c_int
    lib.nsc_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def last_error() -> str:
    return load().nsc_last_error().decode("utf-8", "replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = last_error()
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise NscError(f"{what}: rc={rc}: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
