"""Host-side mirror of the reference's cmrl.py (Cross-Module Residual Learning) for the hot path.

Phases (reference file:line): `_greedy_followers[_lpc]` cmrl.py:22-135 / 137-293 (train the newest codec on the
residual of the frozen earlier ones), `_finetuning[_lpc]` :295-390 / 392-511 (all codecs jointly),
`_feedforward` :876-907 (inference), `model` :909-958 (mode dispatch).  Adam state is re-initialised at every
phase and earlier scopes are restored from the previous phase's checkpoint, like the reference.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .neural_speech_coding_module import neuralSpeechCodingModule, _split


def encode_decode_utterance(eng, utt, soft=True, lpc_x=None, want_entropy=False):
    """Utterance-level inference of cmrl_eval (cmrl.py:566-597) without host round trips: a 1-D float32 CUDA signal is
    cut into hop-480 frames by the framing kernel (`utterance_to_segment(per_sig, True)`, :566), the frames go through
    the cascade in CHUNKS of the engine's batch (the reference feeds them one per sess.run, :585-593 - frames are
    independent, so a batch is the same arithmetic; the last chunk is zero-padded and the padding discarded), and the
    decoded frames are Hann-windowed (first / middle / last variants) and overlap-added by the OLA kernel (:595-597).
    `soft=True` is what cmrl_eval actually feeds (`the_share: 1.0`, :592); end2end_eval feeds hard codes (nsc_module:697).
    One fixed-batch engine serves every utterance length.
    want_entropy: also return the per-frame entropy (bits, summed over the codecs) of each frame's OWN soft histogram -
    cmrl_eval's all_entropy[j] (:593), whose mean it prints as 'Entropy' (:621)."""
    from . import _lib
    from .utilities import frames_on_gpu, num_frames, overlap_add_on_gpu
    nf = num_frames(int(utt.numel()))
    B = eng.B
    frames = frames_on_gpu(utt, post_window=True).view(nf, 1, -1)
    out = torch.empty((nf, frames.shape[-1]), dtype=torch.float32, device=frames.device)
    ent = torch.zeros((nf,), dtype=torch.float32, device=frames.device) if want_entropy else None
    for lo in range(0, nf, B):
        n = min(B, nf - lo)
        if n == B:
            xin = frames[lo:lo + B]
            lx = lpc_x[lo:lo + B] if lpc_x is not None else None
        else:
            xin = eng.buf("infer.in", (B, 1, frames.shape[-1]))
            xin.zero_()
            xin[:n].copy_(frames[lo:lo + n])
            lx = None
            if lpc_x is not None:
                lx = eng.buf("infer.lpc", (B,) + tuple(lpc_x.shape[1:]))
                lx.zero_()
                lx[:n].copy_(lpc_x[lo:lo + n])
        dec = eng.forward(xin.contiguous(), 1.0, soft, lpc_x=lx, want_p=want_entropy)
        out[lo:lo + n].copy_(dec.view(B, -1)[:n])
        if want_entropy:
            fe = eng.buf("infer.frame_ent", (B,))
            for c in eng.codecs:
                _lib.check(eng.lib.nsc_frame_entropy(c.p.data_ptr(), B, c.L, c.nb, fe.data_ptr(), _lib.stream_ptr()), "frame_entropy")
                ent[lo:lo + n] += fe[:n]          # padded frames never enter a histogram: each frame has its own
    sig = overlap_add_on_gpu(out)
    return (sig, ent) if want_entropy else sig


class CMRL(neuralSpeechCodingModule):
    def __init__(self, arg):
        super(CMRL, self).__init__(arg)
        self._num_resnets = arg.num_resnets
        self._from_where_step = int(arg.from_where_step)
        self._learning_rate_greedy_followers = _split(arg.learning_rate_greedy_followers, float)

    def _greedy_followers(self, num_res):
        """cmrl.py:22-135: codec num_res+1 trains on res_scalar*(x - sum of earlier outputs); earlier scopes are
        restored and frozen; fresh Adam slots; lr/epochs = [-2] entries (:131-132)."""
        n = num_res + 1
        eng = self._make_engine(n, per_codec_list_semantics=False)
        prev = '' if num_res == 1 else 'follower_' + str(num_res - 1) + self._suffix
        self.restore(eng, prev, scopes=[f"scope_{i + 1}" for i in range(num_res)] + ["lpc_quan"])
        eng.reset_adam()
        no_quan, quan, tau_map = self._loss_cfgs(n, "follower")
        self.model_training(eng, no_quan, quan, self._learning_rate_greedy_followers[-2], self._epoch_greedy_followers[-2],
                            'the_follower', save_id='follower_' + str(num_res) + self._suffix,
                            the_tau_val=self._coeff_term[3], tau_map=tau_map)
        self._engine = eng
        return eng

    _greedy_followers_lpc = _greedy_followers

    def _finetuning(self, num_res):
        """cmrl.py:295-390 (time domain) / :392-511 (LPC): all scopes trainable, quan = sum_i, tau_i * ent_i
        (no entropy term in the LPC variant), lr/epochs = [-1] entries."""
        eng = self._make_engine(num_res, per_codec_list_semantics=self._is_pure_time_domain)
        if num_res == 1:
            self.restore(eng, '')
        elif self._from_where_step == 3:
            self.restore(eng, 'finetune_' + str(num_res) + self._suffix)
        else:
            self.restore(eng, 'follower_' + str(num_res - 1) + 'end2endcascade')
        eng.reset_adam()
        no_quan, quan, tau_map = self._loss_cfgs(num_res, "finetune")
        self.model_training(eng, no_quan, quan, self._learning_rate_greedy_followers[-1], self._epoch_greedy_followers[-1],
                            'finetune', save_id='finetune_' + str(num_res) + self._suffix + self._save_unique_mark,
                            the_tau_val=self._coeff_term[3], tau_map=tau_map)
        self._engine = eng
        return eng

    _finetuning_lpc = _finetuning

    def _restore_for_inference(self, eng, num_res):
        """Checkpoint choice of _feedforward (cmrl.py:887-902)."""
        try:
            if num_res == 1:
                self.restore(eng, '')
            elif self._from_where_step == 1:
                self.restore(eng, 'follower_' + str(num_res - 1) + 'end2endcascade')
            else:
                self.restore(eng, 'finetune_' + str(num_res) + 'end2endcascade')
        except FileNotFoundError as e:
            print('no checkpoint found (%s): running with freshly initialised weights' % e)

    def _feedforward(self, num_res, utterances=None):
        """Mode '0' (cmrl.py:876-907 -> cmrl_eval :545-644), the in-scope part: restore, then per utterance
        frames -> cascade -> Hann overlap-add on the GPU (`encode_decode_utterance`), print wall time / real-time factor
        (:608-611), entropy and bit rate (:621-625), and the journal line (:626-628; PESQ needs the external binary: nan).
        `utterances`: list of 1-D float arrays (already normalised like _load_sig does); default: synthetic ones.
        Returns the list of decoded signals (NumPy)."""
        from .loss_terms_and_measures import entropy_to_bitrate, snr
        if utterances is None:
            rng = np.random.default_rng(99)
            utterances = [(0.03 * rng.standard_normal(n)).astype(np.float32) for n in (16000, 24000)]
        outs = []
        # ONE fixed-batch engine for every utterance (buffers are sized by the batch; real test sets have nearly unique
        # utterance lengths, so an engine per frame count would grow without bound): frames run in chunks of this batch
        from .engine import CascadeEngine
        strides = [list(self._the_strides)] * num_res
        bins = (self._num_bins_for_follower + [self._num_bins_for_follower[-1]] * num_res)[:num_res]
        eng = CascadeEngine(int(getattr(self, "_infer_batch", 64)), num_res, self._bottleneck_kernel_and_dilation, strides, bins,
                            res_scalar=self._res_scalar, device=self._device, seed=self._seed)
        eng.keep_activations = False     # inference: nothing is kept for a backward pass
        self._restore_for_inference(eng, num_res)
        warm = False
        for i, sig in enumerate(utterances):
            sig = np.asarray(sig, np.float32)
            nf = len(range(0, len(sig) - 512, 480))
            if nf == 0:
                outs.append(np.zeros(0, np.float32))
                continue
            u = torch.from_numpy(sig).to(eng.device)
            if not warm:
                encode_decode_utterance(eng, u, soft=True, want_entropy=True)      # first-use allocations
                warm = True
            torch.cuda.synchronize()
            t0 = time.time()
            dec, frame_ent = encode_decode_utterance(eng, u, soft=True, want_entropy=True)
            # interested_var = [reduce_sum(ent_loss_arr)] (:882) evaluated on ONE frame per sess.run (:585-593); the printed
            # value is the mean over the utterance's frames (:621) - not the entropy of the pooled histogram (>= the mean)
            ent = float(frame_ent.mean().item())
            torch.cuda.synchronize()
            exec_time = time.time() - t0
            sig_duration = len(sig) / 16000.0
            print('Execute time for the neural codec:', exec_time, sig_duration, exec_time / sig_duration)
            d = dec.cpu().numpy()
            the_snr = float(snr(sig.astype(np.float64), d.astype(np.float64))[1])
            print('Test Utterance %1d: SNR: %7.5f dB  PESQ-WB: %6.5f  Entropy: %6.5f  Bit rate: %6.5f  ID: %s' % (
                i, the_snr, float('nan'), ent, entropy_to_bitrate(ent, self._the_strides[0]), self._rand_model_id))
            self._write_to_file_and_update_to_display(
                'Test Utterance %1d: SNR: %7.5f dB  PESQ-WB: %6.5f   Entropy: %6.5f \n' % (i, the_snr, float('nan'), ent))
            outs.append(d)
            self._engine = eng
        return outs

    _feedforward_lpc = _feedforward

    def model(self, training_mode, arg):
        """cmrl.py:909-958."""
        if training_mode == 'one_ae':
            print('one_ae')
            self.one_ae()
        elif training_mode == 'cascaded':
            self.one_ae()
            for i in range(1, self._num_resnets):
                self._greedy_followers(i)
        elif training_mode == 'retrain_from_somewhere':
            self._rand_model_id = arg.base_model_id
            for i in range(1, self._num_resnets):
                self._greedy_followers(i)
        elif training_mode == 'finetune':
            self._rand_model_id = arg.base_model_id
            self._finetuning(self._num_resnets)
        elif training_mode == 'feedforward':
            self._rand_model_id = arg.base_model_id
            self._feedforward(self._num_resnets)
        else:
            pass
