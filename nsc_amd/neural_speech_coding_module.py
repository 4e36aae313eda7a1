"""Host-side mirror of the reference's neural_speech_coding_module.py for the hot path (file:line cited).

`neuralSpeechCodingModule` keeps the reference's constructor contract (an argparse namespace with the 21 flags of
main.py:6-27), its graph-builder method names (they now build the op-surface graph on HIP kernels) and its training
loop (`model_training`, nsc_module:424-549) - the inner `sess.run(trainop)` is `CascadeEngine.train_step`.

Out of scope here: wav loading, PESQ/STOI evaluation, LPC analysis/synthesis.  Consequences:
  * data comes from `--data_root` (.npy frames, the reference's format) or a synthetic generator;
  * the tau controller (nsc_module:494-517) is driven by the entropy of the last training batches of the epoch
    (soft assignment, global batch) instead of the out-of-scope validation utterances ("next" row N2).
"""
from __future__ import annotations

import os
import random
import time

import numpy as np
import torch

from . import constants as K
from . import nn_core_operator as nn
from .engine import CascadeEngine
from .loss_terms_and_measures import entropy_to_bitrate
from .scope import current_store, variable_scope
from .utilities import training_window


def _split(s, typ):
    return [typ(v) for v in str(s).split()]


class neuralSpeechCodingModule(object):
    def __init__(self, arg):
        """nsc_module:26-71.  Space-separated list flags are parsed exactly like the reference."""
        self._learning_rate_tanh = arg.learning_rate_tanh
        self._coeff_term = _split(arg.coeff_term, float)
        self._pretrain_step = arg.pretrain_step
        self._target_entropy = arg.target_entropy
        self._the_strides = _split(arg.the_strides, int)
        self._res_scalar = arg.res_scalar
        self._save_unique_mark = arg.save_unique_mark
        self._num_bins_for_follower = _split(arg.num_bins_for_follower, int)
        self._epoch_tanh = arg.epoch_tanh
        self._epoch_greedy_followers = _split(arg.epoch_greedy_followers, int)
        self._batch_size = arg.batch_size
        self._training_mode = int(arg.training_mode)
        self._base_model_id = arg.base_model_id
        self._suffix = arg.suffix
        self._bottleneck_kernel_and_dilation = _split(arg.bottleneck_kernel_and_dilation, int)
        self._window_size = arg.window_size          # parsed but unused, like nsc_module:70
        self._is_cq = int(getattr(arg, "is_cq", 0) or 0)
        self._lpc_order = 16
        # extras (defaults reproduce the reference's edit-the-source globals)
        self._is_pure_time_domain = not bool(getattr(arg, "lpc_domain", not K.is_pure_time_domain))
        self._max_batches = getattr(arg, "max_batches_per_epoch", None) or 2500     # nsc_module:118
        self._data_root = getattr(arg, "data_root", None)
        self._val_data_root = getattr(arg, "val_data_root", None)
        self._tau_from_validation = bool(getattr(arg, "tau_from_validation", 1))   # reference behaviour
        self._val_data = None
        self._out_root = getattr(arg, "out_root", ".") or "."
        self._seed = int(getattr(arg, "seed", 20200504) or 20200504)
        self._comm = getattr(arg, "comm", None)
        self._device = getattr(arg, "device", "cuda")
        seed_id = getattr(arg, "model_id", None)
        self._rand_model_id = str(seed_id) if seed_id else str(np.random.randint(1000000, 2000000))   # nsc_module:65
        self._load_training_data()
        for d in ("check", "doc"):
            os.makedirs(os.path.join(self._out_root, d), exist_ok=True)
        self._write_to_file_and_update_to_display(str(arg) + '\n\n')
        self._engine = None

    # ------------------------------------------------------------------ data
    def _load_training_data(self):
        """nsc_module:40-55: [N,512] frames (time domain) or [N, 512+16+512] (frame | LSF | residual)."""
        n = int(getattr(self, "_tr_data_size", K.training_data_size))
        if self._data_root and os.path.exists(self._data_root):
            self._tr_data = np.load(self._data_root)[:n].astype(np.float32)
        else:
            nb = max(self._batch_size * (min(self._max_batches, 16) + 1), 2 * self._batch_size)
            rng = np.random.default_rng(1234 + (self._comm.rank if self._comm else 0))
            frames = (np.clip(0.03 * rng.standard_normal((nb, K.frame_length)), -1, 1) * training_window()).astype(np.float32)
            if self._is_pure_time_domain:
                self._tr_data = frames
            else:
                lsf = np.sort(rng.uniform(0.03, 3.1, (nb, self._lpc_order)), axis=1).astype(np.float32)
                res = (np.clip(0.03 * rng.standard_normal((nb, K.frame_length)), -1, 1) * training_window()).astype(np.float32)
                self._tr_data = np.concatenate([frames, lsf, res], 1)

    def _load_validation_data(self):
        """Frames the tau controller's entropy is measured on (the reference reads it off its validation utterances,
        nsc_module:462-470 -> end2end_eval :656-740).  A file next to the training data, or synthetic frames from another seed."""
        root = self._val_data_root
        n = 4 * self._batch_size
        if root and os.path.exists(root):
            self._val_data = np.load(root)[:n].astype(np.float32)
            return
        rng = np.random.default_rng(4321)
        frames = (np.clip(0.03 * rng.standard_normal((n, K.frame_length)), -1, 1) * training_window()).astype(np.float32)
        if self._is_pure_time_domain:
            self._val_data = frames
        else:
            lsf = np.sort(rng.uniform(0.03, 3.1, (n, self._lpc_order)), axis=1).astype(np.float32)
            self._val_data = np.concatenate([frames, lsf, frames], 1)

    def validation_entropy(self, eng):
        """Mean per-frame entropy of each codec over the validation frames (one forward per batch; the per-frame
        histograms are separated on the GPU by nsc_frame_entropy).  Returns a list, one value per codec."""
        if getattr(self, "_val_data", None) is None:
            self._load_validation_data()
        B, dev = self._batch_size, eng.device
        tot, cnt = None, 0
        for k in range(0, self._val_data.shape[0] - B + 1, B):
            rows = self._val_data[k:k + B]
            if self._is_pure_time_domain:
                x = torch.from_numpy(np.ascontiguousarray(rows[:, :K.frame_length].reshape(B, 1, -1))).to(dev)
                lpc_x = None
            else:
                o = K.frame_length + self._lpc_order
                x = torch.from_numpy(np.ascontiguousarray(rows[:, o:o + K.frame_length].reshape(B, 1, -1))).to(dev)
                lpc_x = torch.from_numpy(np.ascontiguousarray(rows[:, K.frame_length:o].reshape(B, self._lpc_order, 1))).to(dev)
            e = eng.frame_entropies(x, lpc_x=lpc_x).sum(dim=1)
            tot = e if tot is None else tot + e
            cnt += B
        return [float(v) / cnt for v in tot.cpu().numpy()] if cnt else None

    def _write_to_file_and_update_to_display(self, the_string):
        """nsc_module:73-79."""
        path = os.path.join(self._out_root, 'doc', self._rand_model_id + self._suffix + self._save_unique_mark + '_journal.txt')
        with open(path, 'a') as f:
            f.write(the_string)

    def _generate_one_epoch_end2end(self, x, y, batchsize):
        """nsc_module:115-121: shuffled batch starts, contiguous rows, first 2500 batches."""
        the_list = list(range(0, x.shape[0] - self._batch_size, self._batch_size))
        random.shuffle(the_list)
        for i in the_list[:self._max_batches]:
            ret = np.reshape(self._tr_data[i:(i + batchsize), :K.frame_length], (batchsize, K.frame_length, 1))
            yield ret, ret

    def _generate_one_epoch_end2end_lpc_fast(self, x, y, batchsize):
        """nsc_module:132-142: (frame, frame, LSF, precomputed residual)."""
        fl, lo = K.frame_length, self._lpc_order
        the_list = list(range(0, self._tr_data.shape[0] - self._batch_size, self._batch_size))
        random.shuffle(the_list)
        for i in the_list[:self._max_batches]:
            blk = self._tr_data[i:i + batchsize]
            ret = blk[:, :fl].reshape(batchsize, fl, 1)
            yield ret, ret, blk[:, fl:fl + lo].reshape(batchsize, lo, 1), blk[:, fl + lo:].reshape(batchsize, fl, 1)

    # ------------------------------------------------------------------ op-surface graph builders
    # (these run on the autograd op surface; the trainers below use the explicit engine for speed)
    def _down_sampling_mod(self, the_input, the_stride=2):
        """nsc_module:152-156."""
        out = nn.conv1d(the_input, self._bottleneck_kernel_and_dilation[2], filter_size=9, padding='SAME', dilation_rate=1,
                        strides=the_stride, activation=None)
        return nn.activation_func(out)

    def _up_sampling_mod_helper(self, the_input, the_stride=2):
        """nsc_module:158-167 (sub-pixel shuffle)."""
        from .ops import ShuffleFn
        assert the_stride == 2
        return ShuffleFn.apply(the_input)

    def _up_sampling_mod(self, the_input, the_stride=2):
        """nsc_module:169-181 (resnet_type 'gln' -> separable conv)."""
        out = nn.conv1d_depth(the_input, int(the_input.shape[-1]), filter_size=9, padding='SAME', dilation_rate=1, strides=1,
                              activation=None)
        return self._up_sampling_mod_helper(nn.activation_func(out), the_stride=the_stride)

    def _stack_bottleneck_blocks(self, compressed_bit, strides=1, is_post_up_samling=True, the_share=False, is_enc=True):
        """nsc_module:183-217."""
        bkd = self._bottleneck_kernel_and_dilation
        assert bkd[2] % strides == 0
        if compressed_bit.shape[-1] == 1:
            wide_layer = bkd[2]
        else:
            wide_layer = int(compressed_bit.shape[-1] / strides) if is_post_up_samling else int(compressed_bit.shape[-1])
        for i in range(len(bkd) - 4):
            flag = i == (len(bkd) - 5)
            compressed_bit = nn.gated_bottleneck(compressed_bit, non_dilated_neck_kernel_size=bkd[1],
                                                 dilated_neck_kernel_size=bkd[0], wide_layer=wide_layer, narrow_layer=bkd[3],
                                                 dilation_rate=bkd[i + 4], is_last_flat=flag, the_share=the_share)
        return compressed_bit

    def _the_encoder_in_each_module(self, the_input, the_stride, the_share):
        """nsc_module:219-237."""
        c = nn.change_channel(the_input, the_channel=self._bottleneck_kernel_and_dilation[2], kernel_size=55, activation=None)
        c = nn.activation_func(c)
        for i in the_stride:
            c = self._stack_bottleneck_blocks(c, is_post_up_samling=False, the_share=the_share)
            c = self._down_sampling_mod(c, the_stride=i)
        post_down_sampling_hidden = c
        c = self._stack_bottleneck_blocks(c, is_post_up_samling=False, the_share=the_share)
        c = nn.change_channel(c, the_channel=1, kernel_size=55, activation=nn.tanh)
        return post_down_sampling_hidden, c

    def _the_decoder_in_each_module(self, the_code, the_stride, the_share):
        """nsc_module:239-260."""
        c = the_code
        pre_up_sampling_hidden = None
        for i in the_stride:
            c = self._stack_bottleneck_blocks(c, is_post_up_samling=False, the_share=the_share, is_enc=False)
            pre_up_sampling_hidden = c
            c = self._up_sampling_mod(c, the_stride=i)
        c = self._stack_bottleneck_blocks(c, is_post_up_samling=False, the_share=the_share, is_enc=False)
        expand_back = nn.change_channel(c, the_channel=1, kernel_size=55, activation=None)
        return pre_up_sampling_hidden, expand_back

    def computational_graph_end2end_quan_on(self, encoded, the_share, is_quan_on, number_bins, the_scope, the_strides):
        """nsc_module:262-295.  Returns the reference's 8-tuple."""
        st = current_store()
        with variable_scope(the_scope):
            alpha = st.get(the_scope + "/alpha", (), lambda s: np.float32(K.init_alpha))
            bins = st.get(the_scope + "/bins", (number_bins,), lambda s: np.linspace(-1, 1, number_bins))
            hidden_1, floating_code = self._the_encoder_in_each_module(encoded, the_strides, the_share)
            soft_assignment_3d, the_final_code = nn.scalar_softmax_quantization(
                floating_code, alpha, bins, is_quan_on, the_share, K.frame_length // (2 ** len(the_strides)), number_bins)
            hidden_2, expand_back = self._the_decoder_in_each_module(the_final_code, the_strides, the_share)
            return soft_assignment_3d, -1, -1, the_final_code[0, :, 0], expand_back[:, :, 0], alpha, bins, soft_assignment_3d

    # ------------------------------------------------------------------ engine-backed training
    def _strides_for(self, code):
        """`--the_strides` read as one code per codec, 4 => [2,2] else [2] (cmrl.py:32; SURVEY §5 ambiguity)."""
        return [2, 2] if code == 4 else [2]

    def _make_engine(self, num_codecs, per_codec_list_semantics):
        if per_codec_list_semantics:        # one_ae / _finetuning: the list IS the down-sampler list of every codec
            strides = [list(self._the_strides)] * num_codecs
        else:                               # followers / all LPC paths: one code per codec
            codes = self._the_strides + [self._the_strides[-1]] * num_codecs
            strides = [self._strides_for(codes[i]) for i in range(num_codecs)]
        bins = (self._num_bins_for_follower + [self._num_bins_for_follower[-1]] * num_codecs)[:num_codecs]
        lpc = not self._is_pure_time_domain
        eng = CascadeEngine(self._batch_size, num_codecs, self._bottleneck_kernel_and_dilation, strides, bins,
                            res_scalar=self._res_scalar, scale_first=lpc, lpc=lpc, device=self._device, seed=self._seed)
        return eng

    def ckpt_path(self, save_id):
        """./check/model_bnn_ac_<id>_<save_id>.ckpt (nsc_module:548) - an .npz of named arrays in TF variable names."""
        return os.path.join(self._out_root, "check", "model_bnn_ac_" + self._rand_model_id + '_' + save_id + ".ckpt.npz")

    def save(self, eng, save_id):
        np.savez(self.ckpt_path(save_id), **{k.replace("/", "|"): v for k, v in eng.named().items()})
        print('Model saved!')

    def restore(self, eng, save_id, scopes=None):
        with np.load(self.ckpt_path(save_id)) as z:
            named = {k.replace("|", "/"): z[k] for k in z.files}
        if scopes is not None:
            named = {k: v for k, v in named.items() if any(k.startswith(s + "/") for s in scopes)}
        eng.load_named(named)
        print('model ' + self.ckpt_path(save_id) + ' is restored!')

    def model_training(self, eng, cfg_no_quan, cfg_quan, the_learning_rate, epoch, flag, save_id='', the_tau_val=1.0,
                       tau_slots=None):
        """nsc_module:424-549 (time domain) / :551-655 (LPC).  Epoch loop, op switch at pretrain_step, tau controller.
        cfg_*: engine step configs for trainop_no_quan / trainop_quan.  tau_slots: indices into cfg_quan['c_ent'] that
        are driven by tau (one per controlled codec)."""
        dev = eng.device
        init_tau = the_tau_val
        taus = [the_tau_val] * max(1, len(tau_slots or [0]))
        lpc = not self._is_pure_time_domain
        frames = 0
        for i in range(epoch):
            if flag == 'pretrain' and i < self._pretrain_step:
                cfg = dict(cfg_no_quan)
                print('no quan op is used')
            else:
                cfg = dict(cfg_quan)
                c_ent = list(cfg["c_ent"])
                for j, slot in enumerate(tau_slots or []):
                    c_ent[slot] = taus[j] if flag == 'finetune' else init_tau
                cfg["c_ent"] = c_ent
                print((flag if flag != 'pretrain' else 'quan op'), init_tau)
            cfg["lr"] = the_learning_rate
            print('Epoch ----------------------- ', i)
            start = time.perf_counter()
            gen = (self._generate_one_epoch_end2end_lpc_fast if lpc else self._generate_one_epoch_end2end)(
                self._tr_data, self._tr_data, self._batch_size)
            terms = None
            for batch in gen:
                if lpc:
                    _, _, b_lpc, b_res = batch
                    x = torch.from_numpy(np.ascontiguousarray(b_res.reshape(self._batch_size, 1, -1))).to(dev)
                    lpc_x = torch.from_numpy(np.ascontiguousarray(b_lpc)).to(dev)
                else:
                    x = torch.from_numpy(np.ascontiguousarray(batch[0].reshape(self._batch_size, 1, -1))).to(dev)
                    lpc_x = None
                terms = eng.train_step(x, x, cfg, lpc_x=lpc_x, comm=self._comm)
                frames += self._batch_size
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - start
            np.random.shuffle(self._tr_data)                                       # nsc_module:460
            ents = [float(e.item()) for e in terms["ent"]] if terms else [0.0]
            if self._tau_from_validation and cfg.get("is_quan_on", 1.0) == 1.0:
                # the control signal of the reference: per-frame entropy on held-out frames (nsc_module:462-470, 715-733)
                vents = self.validation_entropy(eng)
                if vents:
                    ents = vents
                # artefact of the reference's validation pass (nsc_module:740): 'bins<id><epoch>.npy'
                if not (self._comm and self._comm.rank != 0):
                    np.save(os.path.join(self._out_root, 'bins' + self._rand_model_id + str(i) + '.npy'),
                            eng.view(f"scope_{len(eng.codecs)}/bins").detach().cpu().numpy())
            fully_entropy = ents[-1] if flag != 'finetune' else float(sum(ents))
            tl, fl_ = float(terms["time"].mean().item()), float(terms["freq"].mean().item())
            ql = float(terms["quan"][-1].mean().item())
            print('Epoch %3d: time_loss: %7.5f freq_loss: %7.5f modelid: %s _quan_loss: %6.5f,  fully_entropy: %6.5f , '
                  'time: %.3f, tau: %.3f, frames/s: %.1f' % (i, tl, fl_, self._rand_model_id, ql, fully_entropy, elapsed,
                                                             init_tau, frames / max(elapsed, 1e-9) if i == 0 else
                                                             self._batch_size * min(self._max_batches, (self._tr_data.shape[0] - 1) // self._batch_size) / elapsed))
            # ---- tau controller (nsc_module:494-517; LPC variant :630-639) ----
            ent_change = 0.015
            if cfg.get("is_quan_on", 1.0) == 1.0 and cfg is not cfg_no_quan:
                if flag == 'finetune' and len(taus) >= 2 and len(ents) >= 2:
                    for j, target in enumerate((1.5, 2.5)):                        # hard-coded targets :495-512
                        if ents[j] > target:
                            taus[j] += ent_change
                        if ents[j] < target:
                            taus[j] -= ent_change
                elif lpc:
                    if fully_entropy > self._target_entropy + 0.05:
                        init_tau += ent_change
                    elif fully_entropy < self._target_entropy:
                        init_tau -= ent_change * 3
                else:
                    if fully_entropy > self._target_entropy:
                        init_tau += ent_change
                    elif fully_entropy < self._target_entropy:
                        init_tau -= ent_change
            print('Tau: %7.5f' % init_tau, taus)
            self._write_to_file_and_update_to_display(
                'Epoch %3d: time_loss: %7.5f freq_loss: %7.5f _quan_loss: %6.5ftau: %6.5f   fully_entropy: %6.5f  '
                'bitrate_kbps: %6.3f \n' % (i, tl, fl_, ql, init_tau, fully_entropy,
                                            entropy_to_bitrate(fully_entropy, 4 if eng.codecs[-1].L == 128 else 2)))
        if not (self._comm and self._comm.rank != 0):
            self.save(eng, save_id)
        return init_tau

    def _loss_cfgs(self, num_codecs, mode):
        """Step configs for the two optimizers (nsc_module:914-926; cmrl.py:101-113, 355-372, 483-485)."""
        c = self._coeff_term
        lpc = not self._is_pure_time_domain
        zeros = [0.0] * num_codecs
        if mode == "single":            # one_ae / one_ae_lpc
            train = [True] + [False] * (num_codecs - 1)
        elif mode == "follower":        # newest codec only
            train = [False] * (num_codecs - 1) + [True]
        else:                           # finetune: all
            train = [True] * num_codecs
        no_quan = dict(is_quan_on=0.0 if mode == "single" else 1.0, c_time=c[0], c_freq=c[1], c_quan=zeros, c_ent=zeros,
                       trainable=train, slot=0)
        cq = list(zeros)
        ce = list(zeros)
        tau_slots = []
        extra = {}
        if mode in ("single", "follower"):
            cq[-1 if mode == "follower" else 0] = c[2]
            tau_slots = [num_codecs - 1 if mode == "follower" else 0]
            if lpc and mode == "single" and self._is_cq:
                # nsc_module:1036-1046: quan/entropy terms blended by code lengths 16 : L
                L = 128.0 if self._the_strides[0] == 4 else 256.0
                cq[0] = c[2] * L / (16.0 + L)
                extra = dict(c_quan_lpc=c[2] * 16.0 / (16.0 + L))
        else:
            cq = [c[2]] * num_codecs
            if lpc:
                extra = dict(c_quan_lpc=c[2] if self._is_cq else 0.0)   # no entropy term (cmrl.py:483-485)
            else:
                tau_slots = list(range(num_codecs))
        quan = dict(is_quan_on=1.0, c_time=c[0], c_freq=c[1], c_quan=cq, c_ent=ce, trainable=train, slot=1, **extra)
        return no_quan, quan, tau_slots

    def one_ae(self):
        """nsc_module:891-939 (time domain) / :989-1073 (LPC): train codec 1, pretrain then quantised."""
        eng = self._make_engine(1, per_codec_list_semantics=self._is_pure_time_domain)
        self._engine = eng
        no_quan, quan, tau_slots = self._loss_cfgs(1, "single")
        self.model_training(eng, no_quan, quan, self._learning_rate_tanh, self._epoch_tanh, 'pretrain', save_id='',
                            the_tau_val=self._coeff_term[3], tau_slots=tau_slots)
        return eng

    one_ae_lpc = one_ae
