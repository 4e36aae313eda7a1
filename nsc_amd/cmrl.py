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
        no_quan, quan, tau_slots = self._loss_cfgs(n, "follower")
        self.model_training(eng, no_quan, quan, self._learning_rate_greedy_followers[-2], self._epoch_greedy_followers[-2],
                            'the_follower', save_id='follower_' + str(num_res) + self._suffix,
                            the_tau_val=self._coeff_term[3], tau_slots=tau_slots)
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
        no_quan, quan, tau_slots = self._loss_cfgs(num_res, "finetune")
        self.model_training(eng, no_quan, quan, self._learning_rate_greedy_followers[-1], self._epoch_greedy_followers[-1],
                            'finetune', save_id='finetune_' + str(num_res) + self._suffix + self._save_unique_mark,
                            the_tau_val=self._coeff_term[3], tau_slots=tau_slots)
        self._engine = eng
        return eng

    _finetuning_lpc = _finetuning

    def _feedforward(self, num_res, frames=None):
        """cmrl.py:876-907 + cmrl_eval :545-644 restricted to the in-scope part: restore, run encode+quantise+decode
        over frames and report wall time per frame (the reference prints wall time / real-time factor :608-611).
        NOTE the reference feeds one frame per sess.run; here frames are batched."""
        eng = self._make_engine(num_res, per_codec_list_semantics=True)
        try:
            if num_res == 1:
                self.restore(eng, '')
            elif self._from_where_step == 1:
                self.restore(eng, 'follower_' + str(num_res - 1) + 'end2endcascade')
            else:
                self.restore(eng, 'finetune_' + str(num_res) + 'end2endcascade')
        except FileNotFoundError as e:
            print('no checkpoint found (%s): running with freshly initialised weights' % e)
        B = self._batch_size
        if frames is None:
            frames = self._tr_data[:B, :512]
        x = torch.from_numpy(np.ascontiguousarray(frames[:B].reshape(B, 1, 512).astype(np.float32))).to(eng.device)
        eng.keep_activations = False         # inference: nothing is kept for a backward pass
        eng.forward(x, 1.0, True)            # cmrl_eval feeds the_share: 1.0 => soft codes (cmrl.py:592)
        torch.cuda.synchronize()
        t0 = time.time()
        dec = eng.forward(x, 1.0, True)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print('feedforward: %d frames in %.3f ms -> %.2f us/frame, real-time factor %.1f' %
              (B, 1e3 * dt, 1e6 * dt / B, (B * 480 / 16000.0) / dt))
        self._engine = eng
        return dec

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
