"""Loss surface of the hot path - same names as the reference's loss_terms_and_measures.py.

Live functions (reference file:line):  mse_loss :77-79, mse_loss_v1 :82-84, mfcc_loss :151-175, tf_stft :178-183,
quan_loss :257-259, entropy_coding_loss :262-267, entropy_to_bitrate :63-67, bitrate_to_entropy :70-74,
snr :270-277, si_snr :36-49.  Tensor functions take/return torch CUDA tensors (channels_last like the
reference) and run on the HIP kernels of libnsc_hip.so through autograd Functions; there is no CPU path.
Dead/eval-only reference code (SMR, MNR, psd_loss, stft_loss, tp_*, mdct, pesq) is out of scope.
"""
from __future__ import annotations

import numpy as np

from .constants import frame_length, overlap_each_side, sample_rate

MEL_BANKS = (8, 16, 32, 128)  # loss_terms_and_measures.py:133


# ------------------------------------------------------------------------------------------------
# host-side constants
# ------------------------------------------------------------------------------------------------
def _hertz_to_mel(f):
    return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)


def linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins=257, sr=16000, lower_edge_hertz=0.0,
                                upper_edge_hertz=8000.0):
    """What tf.signal.linear_to_mel_weight_matrix returns for the call at loss_terms_and_measures.py:138
    (HTK mel scale, DC bin dropped then zero-padded back, triangles in the mel domain, float64 internally)."""
    lin = np.linspace(0.0, sr / 2.0, num_spectrogram_bins)[1:]
    spec_mel = _hertz_to_mel(lin)[:, None]
    edges = np.linspace(_hertz_to_mel(lower_edge_hertz), _hertz_to_mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    w = np.maximum(0.0, np.minimum((spec_mel - lower) / (center - lower), (upper - spec_mel) / (upper - center)))
    return np.pad(w, [[1, 0], [0, 0]])


_MEL = None


def mel_matrix_cat():
    """[257, 184] float32: the four banks side by side (column ranges 0:8, 8:24, 24:56, 56:184)."""
    global _MEL
    if _MEL is None:
        _MEL = np.concatenate([linear_to_mel_weight_matrix(n) for n in MEL_BANKS], axis=1).astype(np.float32)
    return _MEL


def mel_band_ranges(mel=None):
    """int32 ranges for nsc_recon_loss_banded (include/nsc_hip.h): [184][2] = [k_lo, k_hi) of every column of the [257,184]
    mel matrix, then [257][4][2] = [j_lo, j_hi) of every bin inside each bank.  Every non-zero lies inside its range;
    zeros inside a range are simply multiplied like in the dense form."""
    m = mel_matrix_cat() if mel is None else np.asarray(mel)
    nz = m != 0
    cols = np.zeros((m.shape[1], 2), np.int32)
    for j in range(m.shape[1]):
        k = np.flatnonzero(nz[:, j])
        if k.size:
            cols[j] = (k[0], k[-1] + 1)
    edges = np.concatenate([[0], np.cumsum(MEL_BANKS)])
    rows = np.zeros((m.shape[0], len(MEL_BANKS), 2), np.int32)
    for k in range(m.shape[0]):
        for b in range(len(MEL_BANKS)):
            j = np.flatnonzero(nz[k, edges[b]:edges[b + 1]])
            if j.size:
                rows[k, b] = (edges[b] + j[0], edges[b] + j[-1] + 1)
    return np.concatenate([cols.reshape(-1), rows.reshape(-1)]).astype(np.int32)


# ------------------------------------------------------------------------------------------------
# scalar helpers (pure host arithmetic, as in the reference)
# ------------------------------------------------------------------------------------------------
def entropy_to_bitrate(total_entropy, the_strides):
    code_len_val = 128 if the_strides == 4 else 256
    return ((sample_rate / 1024.0) / (frame_length - overlap_each_side)) * code_len_val * total_entropy


def bitrate_to_entropy(bitrate, the_strides):
    pre = (frame_length / the_strides) * (float(frame_length / the_strides) / frame_length)
    entropy = (bitrate / pre * sample_rate)
    entropy *= (frame_length - overlap_each_side / float(frame_length))
    return entropy


def snr(ori_sig, dec_sig):
    min_len = min(len(ori_sig), len(dec_sig))
    ori_sig, dec_sig = ori_sig[:min_len], dec_sig[:min_len]
    nom = np.sum(np.power(ori_sig, 2))
    denom = np.sum(np.power(np.subtract(ori_sig, dec_sig), 2))
    eps = 1e-20
    return min_len, 10 * np.log10(nom / (denom + eps) + eps), ori_sig, dec_sig


def si_snr(x, s):
    x_zm = x - np.mean(x)
    s_zm = s - np.mean(s)
    t = (np.inner(x_zm, s_zm) / np.linalg.norm(s_zm, 2) ** 2) * s_zm
    n = x_zm - t
    return 20 * np.log10(np.linalg.norm(t, 2) / np.linalg.norm(n, 2))


# ------------------------------------------------------------------------------------------------
# tensor losses (HIP kernels; autograd glue only)
# ------------------------------------------------------------------------------------------------
def _ops():
    from . import ops
    return ops


def mse_loss(decoded_sig, original_sig, kai_re_mat=1):
    """sqrt(mean_t (d-o)^2 + 1e-7) per frame -> [B]."""
    return _ops().recon_losses(decoded_sig, original_sig)[0]


mse_loss_v1 = mse_loss


def mfcc_loss(decoded_sig, original_sig, is_finetuning=False):
    """rFFT-512 -> PSD -> 4 mel banks -> log -> per-bank RMSE -> mean -> [B]."""
    return _ops().recon_losses(decoded_sig, original_sig)[1]


def tf_stft(sig, the_frame_length=frame_length):
    """Returns (complex64 stft [B,257], magnitude [B,257]) like the reference (window_fn=None => bare rFFT)."""
    return _ops().rfft512(sig)


def quan_loss(softmax_assignment):
    """mean_l sum_k sqrt(p + 1e-20) -> [B]."""
    return _ops().quan_loss(softmax_assignment)


def entropy_coding_loss(soft_assignment):
    """Entropy (bits) of the batch-global soft histogram -> scalar."""
    return _ops().entropy_coding_loss(soft_assignment)
