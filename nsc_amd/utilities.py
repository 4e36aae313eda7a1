"""Framing helpers of the reference's utilities.py:7-39 (hop 480, 512-sample frames, three Hann variants).

`utterance_to_segment` / `hann_process` keep the reference signatures on NumPy arrays (host arithmetic, like the
reference); `frames_on_gpu` / `overlap_add_on_gpu` are the batched HIP versions used by the inference path
(SURVEY 8f N1): whole utterances go wav -> frames -> codec -> overlap-add without host round trips.
"""
from __future__ import annotations

import numpy as np

from .constants import frame_length, overlap_each_side


def training_window():
    """utilities.py:28-30: hanning(63)[:32] | ones(448) | hanning(63)[31:]."""
    o = overlap_each_side
    return np.append(np.append(np.hanning(o * 2 - 1)[:o], np.array([1] * (frame_length - o * 2))),
                     np.hanning(o * 2 - 1)[o - 1:])


def hann_windows3():
    """[3,512] float32: first / middle / last windows of utilities.py:10-15."""
    o = overlap_each_side
    mid = np.array([1] * (frame_length - o * 2))
    first = np.append(np.append(np.array([1] * o), mid), np.hanning(o * 2)[o:])
    last = np.append(np.append(np.hanning(o * 2)[:o], mid), np.array([1] * o))
    return np.stack([first, training_window(), last]).astype(np.float32)


def hann_process(utterance_seg, seg_ind, seg_amount):
    """utilities.py:7-22."""
    w = hann_windows3().astype(np.float64) if False else None
    o = overlap_each_side
    mid = np.array([1] * (frame_length - o * 2))
    if seg_ind == 0:
        return utterance_seg * np.append(np.append(np.array([1] * o), mid), np.hanning(o * 2)[o:])
    if seg_ind == seg_amount - 1:
        return utterance_seg * np.append(np.append(np.hanning(o * 2)[:o], mid), np.array([1] * o))
    return utterance_seg * training_window()


def num_frames(n):
    """len(range(0, n - 512, 480)) - the reference's frame count (utilities.py:26)."""
    return len(range(0, n - frame_length, frame_length - overlap_each_side))


def utterance_to_segment(utterance, post_window=False):
    """utilities.py:25-39."""
    hop = frame_length - overlap_each_side
    ret = np.empty((num_frames(len(utterance)), frame_length))
    win = 1 if post_window else training_window()
    for ind, i in enumerate(range(0, len(utterance) - frame_length, hop)):
        ret[ind, :] = utterance[i:(i + frame_length)] * win
    return ret


def frames_on_gpu(utt, post_window=False):
    """utt: 1-D float32 CUDA tensor -> [nframes, 512] CUDA tensor (bit-exact frame indexing; HIP kernel)."""
    import torch
    from . import _lib
    lib = _lib.load()
    n = int(utt.numel())
    nf = num_frames(n)
    frames = torch.empty((nf, frame_length), dtype=torch.float32, device=utt.device)
    win = None if post_window else torch.from_numpy(training_window().astype(np.float32)).to(utt.device)
    _lib.check(lib.nsc_frame_utterance(utt.data_ptr(), n, _lib.ptr(win), frames.data_ptr(), nf,
                                       torch.cuda.current_stream().cuda_stream), "frame_utterance")
    return frames


def overlap_add_on_gpu(frames):
    """[nframes,512] decoded frames -> Hann-windowed overlap-add signal (cmrl.py:595-597), HIP kernel."""
    import torch
    from . import _lib
    lib = _lib.load()
    nf = int(frames.shape[0])
    out = torch.empty(480 * (nf - 1) + frame_length, dtype=torch.float32, device=frames.device)
    win3 = torch.from_numpy(hann_windows3()).to(frames.device)
    _lib.check(lib.nsc_overlap_add(frames.contiguous().data_ptr(), nf, win3.data_ptr(), out.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "overlap_add")
    return out
