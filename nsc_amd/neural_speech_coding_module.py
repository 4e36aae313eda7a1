"""Host-side mirror of the reference's neural_speech_coding_module.py for the hot path (file:line cited).

`neuralSpeechCodingModule` keeps the reference's constructor contract (an argparse namespace with the 21 flags of
main.py:6-27), its graph-builder method names (they now build the op-surface graph on HIP kernels) and its training
loop (`model_training`, nsc_module:424-549) - the inner `sess.run(trainop)` is `CascadeEngine.train_step`.

Out of scope here: wav loading, PESQ/STOI evaluation, LPC analysis of raw audio.  Consequences:
  * data comes from `--data_root` (.npy frames, the reference's format) or a synthetic generator;
  * the tau controller (nsc_module:494-517, 630-639) is driven, like the reference's, by the per-frame (batch-of-1,
    hard assignment) entropy of held-out frames (`validation_entropy`); `--tau_from_validation 0`
    falls back to the entropy of the epoch's last training batch;
  * journal lines keep the reference's format strings (nsc_module:520-530, 623-628); STOI / PESQ, which need the
    out-of-scope tools, are written as nan.

Two behaviours of the shipped reference that this module does NOT reproduce (both found by executing it, see
tests/golden/make_reference_exec.py): `model_training` feeds `tau` as a (1,2) array built from tau_1/tau_2 even when
the graph's tau is a scalar (one_ae / followers), which only broadcasts against the [B] loss for B in {1,2} and never
feeds the value its controller updates; and `end2end_eval` raises for every flag but 'finetune'.  Here the controller's
tau is the tau that is fed (what nsc_module:494-517 plainly intends and what the finetune / LPC loops do).
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


ENT_CHANGE = 0.015     # nsc_module:494, 630


def tau_controller(flag, lpc, is_quan_on_val, taus, fully_entropy, ents, target_entropy):
    """One epoch of the reference's entropy controller.  taus = (tau, tau_1, tau_2) -> new (tau, tau_1, tau_2).
    time domain (nsc_module:494-517): flag 'finetune' moves tau_1 / tau_2 by 0.015 toward the hard-coded per-codec
    targets 1.5 / 2.5; any other flag moves tau by 0.015 toward --target_entropy.
    LPC (nsc_module:630-639): only while is_quan_on == 1: +0.015 above target + 0.05, -0.045 below target."""
    tau, t1, t2 = taus
    if lpc:
        if is_quan_on_val == 1.0:
            if fully_entropy > target_entropy + 0.05:
                tau += ENT_CHANGE
            elif fully_entropy < target_entropy:
                tau -= ENT_CHANGE * 3
        return tau, t1, t2
    if flag == 'finetune':
        target_1, target_2 = 1.5, 2.5
        e1, e2 = ents[0], (ents[1] if len(ents) > 1 else ents[0])
        if e1 > target_1:
            t1 += ENT_CHANGE
        if e1 < target_1:
            t1 -= ENT_CHANGE
        if e2 > target_2:
            t2 += ENT_CHANGE
        if e2 < target_2:
            t2 -= ENT_CHANGE
    else:
        if fully_entropy > target_entropy:
            tau += ENT_CHANGE
        elif fully_entropy < target_entropy:
            tau -= ENT_CHANGE
    return tau, t1, t2


def journal_line(i, ave_snr, ave_si_snr, ave_stoi, ave_pesq, _quan_loss, init_tau, fully_entropy):
    """The per-epoch journal line of model_training (nsc_module:520-530), format string kept verbatim."""
    return ('Epoch %3d: SNR: %7.5f dB Si-SNR: %7.5f dB STOI: '
            '%6.5f PESQ: %6.5f _quan_loss: %6.5f'
            'tau: %6.5f   '
            'fully_entropy: %6.5f \n' % (i, ave_snr, ave_si_snr, ave_stoi, ave_pesq, _quan_loss, init_tau, fully_entropy))


def journal_line_lpc(i, ave_snr, ave_stoi, ave_pesq, _quan_loss, fully_snr, fully_pesq, fully_entropy):
    """The per-epoch journal line of model_training_lpc (nsc_module:623-628), format string kept verbatim."""
    return ('Epoch %3d: SNR: %7.5f dB    STOI: %6.5f   PESQ: %6.5f   _quan_loss: %6.5f  fully_snr: %6.5f   fully_pesq: %6.5f  '
            'fully_entropy: %6.5f \n' % (i, ave_snr, ave_stoi, ave_pesq, _quan_loss, fully_snr, fully_pesq, fully_entropy))


def epoch_permutation(n, seed, epoch, perm=None):
    """Row order of one epoch.  The reference shuffles the training matrix in place after every epoch (nsc_module:460) with
    NumPy's unseeded global generator; here the same operation is a seeded permutation of row INDICES, identical on every
    rank: epoch e's order is epoch e-1's order composed with a shuffle drawn from (seed, e)."""
    perm = np.arange(n, dtype=np.int64) if perm is None else perm
    if epoch > 0:
        np.random.default_rng([int(seed), int(epoch)]).shuffle(perm)
    return perm


def epoch_batches(perm, batch, world, rank, seed, epoch, max_batches):
    """Rows of every step of one epoch for ONE rank.  nsc_module:115-121: batch starts range(0, N - batch, batch), shuffled,
    contiguous rows i:i+batch of the (shuffled) matrix, first max_batches starts.  Data parallel: the GLOBAL batch is
    batch * world rows from one start, rank r owns rows [start + r batch, start + (r + 1) batch) - so an N-rank run sees
    exactly the steps of a 1-process run at batch N * batch, and no two ranks ever see the same row in a step."""
    gb = batch * world
    starts = list(range(0, perm.shape[0] - gb, gb))
    random.Random(f"{int(seed)}/{int(epoch)}").shuffle(starts)
    return [perm[i + rank * batch:i + (rank + 1) * batch] for i in starts[:max_batches]]


class _Feeder:
    """Host -> HBM feed of the training loop: rows are gathered into one of two PINNED host buffers and copied on a side
    stream into one of two device buffers while the previous step computes (the reference hands sess.run a pageable
    NumPy slice per step, nsc_module:455-458).  next() returns device tensors valid until the call after next."""

    def __init__(self, batch, width, device):
        self.dev = torch.device(device)
        self.pin = [torch.empty((batch, width), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.buf = [torch.empty((batch, width), dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.copied = [torch.cuda.Event() for _ in range(2)]      # H2D of slot k finished (pinned buffer reusable)
        self.used = [None, None]                                   # main-stream event: the step that read slot k has run
        self.k = 0

    def put(self, rows_np):
        """Start moving one batch (a [batch, width] float32 array view / gather) to the device; returns the slot."""
        k = self.k
        self.k ^= 1
        self.copied[k].synchronize()                               # the copy that last used this pinned buffer is done
        np.copyto(self.pin[k].numpy(), rows_np)
        if self.used[k] is not None:
            self.copy_stream.wait_event(self.used[k])              # the step two back no longer reads this device buffer
        with torch.cuda.stream(self.copy_stream):
            self.buf[k].copy_(self.pin[k], non_blocking=True)
            self.copied[k].record(self.copy_stream)
        return k

    def get(self, k):
        """Device batch of slot k, ordered after its copy on the current stream."""
        torch.cuda.current_stream().wait_event(self.copied[k])
        return self.buf[k]

    def done(self, k):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.used[k] = ev


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
        self._local_entropy = bool(getattr(arg, "local_entropy", 0))   # skip the histogram all-reduce
        self._tf_checkpoint = bool(getattr(arg, "tf_checkpoint", 0))   # also write TF V2-format checkpoints (nsc_amd/tf_checkpoint.py)
        self._dump_rows = bool(getattr(arg, "dump_rows", 0))
        self._rows_seen = []
        self._comm = getattr(arg, "comm", None)
        self._device = getattr(arg, "device", "cuda")
        seed_id = getattr(arg, "model_id", None)
        self._rand_model_id = str(seed_id) if seed_id else str(np.random.randint(1000000, 2000000))   # nsc_module:65
        if self._comm is not None and self._comm.world > 1:
            # one id for the whole job: checkpoints / journal are written by rank 0 and read back by every rank
            self._rand_model_id = self._comm.broadcast_object(self._rand_model_id)
        self._epochs_done = 0           # epochs over all phases: seeds the row order (identical on every rank)
        self._load_training_data()
        for d in ("check", "doc"):
            os.makedirs(os.path.join(self._out_root, d), exist_ok=True)
        shown = {k: v for k, v in vars(arg).items() if k != "comm"} if hasattr(arg, "__dict__") else arg
        self._write_to_file_and_update_to_display(str(shown) + '\n\n')
        self._engine = None

    # ------------------------------------------------------------------ data
    def _world_rank(self):
        return (self._comm.world, self._comm.rank) if self._comm is not None else (1, 0)

    def _load_training_data(self):
        """nsc_module:40-55: [N,512] frames (time domain) or [N, 512+16+512] (frame | LSF | residual).  A --data_root file is
        memory-mapped: a rank only ever touches the rows of its own share of each global batch (epoch_batches), so N ranks
        read disjoint rows and no rank holds the matrix.  Row order lives in self._perm (epoch_permutation)."""
        n = int(getattr(self, "_tr_data_size", K.training_data_size))
        world, _ = self._world_rank()
        if self._data_root and os.path.exists(self._data_root):
            self._tr_data = np.load(self._data_root, mmap_mode="r")[:n]
        else:
            # synthetic frames: the SAME matrix on every rank, sized for the global batch, sharded like a file
            nb = max(self._batch_size * world * (min(self._max_batches, 16) + 1), 2 * self._batch_size * world)
            rng = np.random.default_rng(1234)
            frames = (np.clip(0.03 * rng.standard_normal((nb, K.frame_length)), -1, 1) * training_window()).astype(np.float32)
            if self._is_pure_time_domain:
                self._tr_data = frames
            else:
                lsf = np.sort(rng.uniform(0.03, 3.1, (nb, self._lpc_order)), axis=1).astype(np.float32)
                res = (np.clip(0.03 * rng.standard_normal((nb, K.frame_length)), -1, 1) * training_window()).astype(np.float32)
                self._tr_data = np.concatenate([frames, lsf, res], 1)
        self._perm = epoch_permutation(self._tr_data.shape[0], self._seed, 0)
        self._res_override = None       # residual columns recomputed by _update_lpc_residual (the file itself is read-only)

    def _load_validation_data(self):
        """Frames the tau controller's entropy is measured on (the reference reads it off its validation utterances,
        nsc_module:462-470 -> end2end_eval :656-740).  A file next to the training data, or synthetic frames from another seed."""
        root = self._val_data_root
        n = 4 * self._batch_size * self._world_rank()[0]     # the same frames whatever the number of ranks
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

    def validation_pass(self, eng):
        """The in-scope part of the reference's validation (end2end_eval / end2end_eval_lpc, nsc_module:657-758,
        760-889): every held-out frame goes through the cascade with the HARD assignment and is_quan_on = 1, and
        entropy_coding_loss is evaluated on each frame's OWN histogram (the reference feeds one frame per sess.run).
        Here a whole batch goes through one forward and nsc_frame_entropy separates the per-frame histograms.
        Returns dict(ent=[per codec], ent_lpc=float|None, snr, si_snr, quan=[per codec])."""
        from .loss_terms_and_measures import si_snr, snr
        if getattr(self, "_val_data", None) is None:
            self._load_validation_data()
        B, dev = self._batch_size, eng.device
        tot, tot_lpc, cnt = None, 0.0, 0
        ori, dec = [], []
        fl, lo = K.frame_length, self._lpc_order
        for k in range(0, self._val_data.shape[0] - B + 1, B):
            rows = self._val_data[k:k + B]
            if self._is_pure_time_domain:
                x = torch.from_numpy(np.ascontiguousarray(rows[:, :fl].reshape(B, 1, -1))).to(dev)
                lpc_x = None
            else:
                x = torch.from_numpy(np.ascontiguousarray(rows[:, fl + lo:fl + lo + fl].reshape(B, 1, -1))).to(dev)
                lpc_x = torch.from_numpy(np.ascontiguousarray(rows[:, fl:fl + lo].reshape(B, lo, 1))).to(dev)
            e = eng.frame_entropies(x, lpc_x=lpc_x)
            tot = e.sum(dim=1) if tot is None else tot + e.sum(dim=1)
            if lpc_x is not None:
                tot_lpc += float(eng.lpc_frame_ent.sum().item())
                # end2end_eval_lpc scores the SYNTHESISED signal against the (filtered) utterance (nsc_module:826-860), not the
                # residual against its reconstruction: hard-quantised LSFs -> A(z) -> all-pole synthesis of the decoded residual,
                # compared with the raw-frame columns of the validation matrix (frame by frame: the matrix holds frames, not utterances)
                from .lpc_utilities import lpc_synthesizer_tr, lsf2poly_after_quan, quantize_lsf_hard
                q = quantize_lsf_hard(lpc_x, eng.view("lpc_quan/alpha"), eng.view("lpc_quan/bins"))
                syn = lpc_synthesizer_tr(lsf2poly_after_quan(q, lo), eng.decoded.reshape(B, fl))
                ori.append(rows[:, :fl].reshape(-1).astype(np.float32))
                dec.append(syn.reshape(-1).cpu().numpy())
            else:
                ori.append(x.reshape(-1).cpu().numpy())
                dec.append(eng.decoded.reshape(-1).cpu().numpy())
            cnt += B
        if not cnt:
            return None
        o, d = np.concatenate(ori).astype(np.float64), np.concatenate(dec).astype(np.float64)
        return dict(ent=[float(v) / cnt for v in tot.cpu().numpy()],
                    ent_lpc=(tot_lpc / cnt if not self._is_pure_time_domain else None),
                    snr=float(snr(o, d)[1]), si_snr=float(si_snr(d, o)))

    def validation_entropy(self, eng):
        """Mean per-frame entropy of each codec over the validation frames.  Returns a list, one value per codec."""
        r = self.validation_pass(eng)
        return r["ent"] if r else None

    def _write_to_file_and_update_to_display(self, the_string):
        """nsc_module:73-79."""
        if self._comm is not None and self._comm.rank != 0:
            return                      # one writer per job
        path = os.path.join(self._out_root, 'doc', self._rand_model_id + self._suffix + self._save_unique_mark + '_journal.txt')
        with open(path, 'a') as f:
            f.write(the_string)

    def _epoch_rows(self, epoch):
        """This rank's row indices for every step of the epoch (see epoch_batches)."""
        world, rank = self._world_rank()
        return epoch_batches(self._perm, self._batch_size, world, rank, self._seed, epoch, self._max_batches)

    def _generate_one_epoch_end2end(self, x, y, batchsize, epoch=0):
        """nsc_module:115-121: shuffled batch starts, contiguous rows of the shuffled matrix, first 2500 batches."""
        for rows in self._epoch_rows(epoch):
            blk = np.asarray(self._tr_data[rows])
            ret = blk[:, :K.frame_length].reshape(batchsize, K.frame_length, 1)
            yield ret, ret

    def _generate_one_epoch_end2end_lpc_fast(self, x, y, batchsize, epoch=0):
        """nsc_module:132-142: (frame, frame, LSF, precomputed residual)."""
        fl, lo = K.frame_length, self._lpc_order
        for rows in self._epoch_rows(epoch):
            blk = np.asarray(self._tr_data[rows])
            res = blk[:, fl + lo:] if self._res_override is None else self._res_override[rows]
            ret = blk[:, :fl].reshape(batchsize, fl, 1)
            yield ret, ret, blk[:, fl:fl + lo].reshape(batchsize, lo, 1), res.reshape(batchsize, fl, 1)

    # ------------------------------------------------------------------ op-surface graph builders
    # (these run on the autograd op surface; the trainers below use the explicit engine for speed)
    def _down_sampling_mod(self, the_input, the_stride=2):
        """nsc_module:152-156 (conv1d(activation=None) then activation_func there: the leaky-relu rides the conv's epilogue here)."""
        return nn.conv1d(the_input, self._bottleneck_kernel_and_dilation[2], filter_size=9, padding='SAME', dilation_rate=1,
                         strides=the_stride, activation='lrelu')

    def _up_sampling_mod_helper(self, the_input, the_stride=2):
        """nsc_module:158-167 (sub-pixel shuffle)."""
        from .ops import ShuffleFn
        assert the_stride == 2
        return ShuffleFn.apply(the_input)

    def _up_sampling_mod(self, the_input, the_stride=2):
        """nsc_module:169-181 (resnet_type 'gln' -> separable conv)."""
        # (conv1d_depth(activation=None) -> activation_func -> _up_sampling_mod_helper there: one call, one fused kernel per direction)
        return nn.conv1d_depth_shuffle(the_input, int(the_input.shape[-1]), filter_size=9, activation='lrelu', stride=the_stride)

    def _stack_bottleneck_blocks(self, compressed_bit, strides=1, is_post_up_samling=True, the_share=False, is_enc=True):
        """nsc_module:183-217."""
        bkd = self._bottleneck_kernel_and_dilation
        assert bkd[2] % strides == 0
        if compressed_bit.shape[-1] == 1:
            wide_layer = bkd[2]
        else:
            wide_layer = int(compressed_bit.shape[-1] / strides) if is_post_up_samling else int(compressed_bit.shape[-1])
        # (the reference's loop over bkd[4:], one gated_bottleneck per dilation rate, the last one flat: one call here)
        compressed_bit = nn.gated_bottleneck_stack(compressed_bit, wide_layer=wide_layer, narrow_layer=bkd[3],
                                                   non_dilated_neck_kernel_size=bkd[1], dilation_rates=bkd[4:], is_last_flat=True,
                                                   the_share=the_share)
        return compressed_bit

    def _the_encoder_in_each_module(self, the_input, the_stride, the_share):
        """nsc_module:219-237."""
        c = nn.change_channel(the_input, the_channel=self._bottleneck_kernel_and_dilation[2], kernel_size=55, activation='lrelu')
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

    def computational_graph_end2end_quan_on_lpc(self, encoded, quan_lpc_coeff, the_share, is_quan_on, number_bins,
                                                the_scope, the_strides):
        """nsc_module:297-335.  Same graph as the time-domain builder; bins span +-beta_boundary (:308), the quantised LPC
        polynomial argument is unused by the graph (its only use is commented out, :329-331).  Returns the 7-tuple."""
        st = current_store()
        with variable_scope(the_scope):
            alpha = st.get(the_scope + "/alpha", (), lambda s: np.float32(K.init_alpha))
            bins = st.get(the_scope + "/bins", (number_bins,),
                          lambda s: np.linspace(-K.beta_boundary, K.beta_boundary, number_bins))
            _, floating_code = self._the_encoder_in_each_module(encoded, the_strides, the_share)
            soft_assignment_3d, the_final_code = nn.scalar_softmax_quantization(
                floating_code, alpha, bins, is_quan_on, the_share, K.frame_length // (2 ** len(the_strides)), number_bins)
            _, expand_back = self._the_decoder_in_each_module(the_final_code, the_strides, the_share)
            return soft_assignment_3d, -1, -1, the_final_code[0, :, 0], expand_back[:, :, 0], alpha, bins

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
        named = eng.named()
        np.savez(self.ckpt_path(save_id), **{k.replace("/", "|"): v for k, v in named.items()})
        if getattr(self, "_tf_checkpoint", False):
            # what saver.save(sess, './check/model_bnn_ac_<id>_<save_id>.ckpt') leaves (nsc_module:548): same prefix, TF variable names
            from .tf_checkpoint import write_checkpoint
            write_checkpoint(self.ckpt_path(save_id)[:-len(".npz")], named)
        print('Model saved!')

    def restore(self, eng, save_id, scopes=None):
        path = self.ckpt_path(save_id)
        if not os.path.exists(path) and os.path.exists(path[:-len(".npz")] + ".index"):
            # a checkpoint written by the reference's tf.compat.v1.train.Saver (V2 format): same variable names (optimizer slots and
            # counters, if any, are not parameters of the engine and are ignored by load_named)
            from .tf_checkpoint import read_checkpoint
            named = read_checkpoint(path[:-len(".npz")])
        else:
            with np.load(path) as z:
                named = {k.replace("|", "/"): z[k] for k in z.files}
        if scopes is not None:
            named = {k: v for k, v in named.items() if any(k.startswith(s + "/") for s in scopes)}
        eng.load_named(named)
        print('model ' + self.ckpt_path(save_id) + ' is restored!')

    def _update_lpc_residual(self, eng):
        """nsc_module:1075-1121: hard-quantise every stored LSF vector with the current (sorted) LSF codebook, rebuild
        A(z) and re-filter the raw frames; the residual columns of the training matrix are replaced.  The three
        py_func bodies run as HIP kernels (nsc_amd/lpc_utilities.py)."""
        from .lpc_utilities import residual_from_lsf
        fl, lo = K.frame_length, self._lpc_order
        frames = torch.from_numpy(np.ascontiguousarray(self._tr_data[:, :fl], dtype=np.float32))
        lsf = torch.from_numpy(np.ascontiguousarray(self._tr_data[:, fl:fl + lo], dtype=np.float32))
        alpha = torch.full((1,), float(K.init_alpha), dtype=torch.float32, device=eng.device)   # a fresh alpha, :1084
        res = residual_from_lsf(frames, lsf, alpha, eng.view("lpc_quan/bins"))
        # every rank recomputes the whole column block (device kernels, once per 30 epochs); the file stays read-only
        self._res_override = res.cpu().numpy().astype(np.float32)

    def _barrier(self):
        if self._comm is not None:
            self._comm.barrier()

    def _is_writer(self):
        return self._comm is None or self._comm.rank == 0

    def model_training(self, eng, cfg_no_quan, cfg_quan, the_learning_rate, epoch, flag, save_id='', the_tau_val=1.0,
                       tau_map=None):
        """nsc_module:424-549 (time domain) / :551-655 (LPC).  Epoch loop, op switch at pretrain_step, tau controller,
        journal line, checkpoint.  cfg_*: engine step configs of trainop_no_quan / trainop_quan.
        tau_map: [(cfg key, index or None, scale, which tau)] - where the controlled tau(s) enter the quan op's loss:
        cfg[key][index] = scale * tau (``which`` = 0: the single tau; j >= 1: tau_j of the finetune phase)."""
        dev = eng.device
        init_tau = the_tau_val
        init_tau_1 = the_tau_val
        init_tau_2 = the_tau_val
        lpc = not self._is_pure_time_domain
        tau_map = tau_map or []
        nan = float('nan')
        for i in range(epoch):
            print('-----------------------')
            if flag == 'pretrain' and i < self._pretrain_step:
                cfg = dict(cfg_no_quan)
                is_quan_on_val = 0.0
                print('no quan op is used')
            else:
                cfg = dict(cfg_quan)
                is_quan_on_val = 1.0
                for key, idx, scale, which in tau_map:
                    t = (init_tau, init_tau_1, init_tau_2)[which]
                    if idx is None:
                        cfg[key] = scale * t
                    else:
                        lst = list(cfg[key])
                        lst[idx] = scale * t
                        cfg[key] = lst
                print(('quan op' if flag == 'pretrain' else flag + ' is working.'), init_tau)
            cfg["lr"] = the_learning_rate
            print('Epoch ----------------------- ', i)
            start = time.perf_counter()
            refresh = lpc and self._is_cq and ((i % 30 == 0 and i != 0) or i == epoch - 3)    # nsc_module:578
            if refresh:
                # the reference spends this epoch recomputing the residuals with the current LSF codebook and saves
                print('=============Recalculating the residuals...=============')
                self._update_lpc_residual(eng)
                if self._is_writer():
                    self.save(eng, save_id)
                self._barrier()
            gen = () if refresh else (self._generate_one_epoch_end2end_lpc_fast if lpc else self._generate_one_epoch_end2end)(
                self._tr_data, self._tr_data, self._batch_size, epoch=self._epochs_done)
            terms, nsteps = None, 0
            fl, lo, Bsz = K.frame_length, self._lpc_order, self._batch_size
            if getattr(self, "_feeder", None) is None:
                self._feeder = _Feeder(Bsz, fl + (lo if lpc else 0), dev)
            feeder = self._feeder

            def put(batch):
                # one [B, 512 (+16)] host row block per step: residual | LSF on the LPC path (the feed dict of :586-595)
                if lpc:
                    return feeder.put(np.concatenate([batch[3].reshape(Bsz, fl), batch[2].reshape(Bsz, lo)], 1))
                return feeder.put(batch[0].reshape(Bsz, fl))
            if self._dump_rows and not refresh:
                self._rows_seen += [r.copy() for r in self._epoch_rows(self._epochs_done)]
            it = iter(gen)
            nxt = next(it, None)
            slot = put(nxt) if nxt is not None else None
            while nxt is not None:
                cur_slot = slot
                nxt = next(it, None)
                d = feeder.get(cur_slot)
                x = d[:, :fl].reshape(Bsz, 1, fl)
                if lpc:
                    x = x.contiguous()
                    lpc_x = d[:, fl:].reshape(Bsz, lo, 1).contiguous()
                else:
                    lpc_x = None
                terms = eng.train_step(x, x, cfg, lpc_x=lpc_x, comm=self._comm)
                feeder.done(cur_slot)
                if nxt is not None:
                    slot = put(nxt)              # gathered and copied while the step above runs
                nsteps += 1
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - start
            if nsteps and eng.pair_timeouts():
                raise RuntimeError("a neighbour wait of a pair launch timed out (gated-block stack kernels): results invalid; "
                                   "set CascadeEngine.fused_pairs = False")
            self._epochs_done += 1
            self._perm = epoch_permutation(self._tr_data.shape[0], self._seed, self._epochs_done, self._perm)   # nsc_module:460
            # ---- validation signal (nsc_module:462-470; the loops themselves are out of scope) ----
            ents = [float(e.item()) for e in terms["ent"]] if terms else [0.0] * len(eng.codecs)
            ent_lpc = float(terms["ent_lpc"].item()) if (terms and terms.get("ent_lpc") is not None) else 0.0
            ave_snr = ave_si_snr = nan
            if self._tau_from_validation:
                v = self.validation_pass(eng)
                if v:
                    ents, ave_snr, ave_si_snr = v["ent"], v["snr"], v["si_snr"]
                    ent_lpc = v["ent_lpc"] if v["ent_lpc"] is not None else 0.0
                if self._is_writer():   # artefacts of the reference's validation pass (nsc_module:740, 839)
                    np.save(os.path.join(self._out_root, 'bins' + self._rand_model_id + str(i) + '.npy'),
                            eng.view(f"scope_{len(eng.codecs)}/bins").detach().cpu().numpy())
                    if lpc:
                        np.save(os.path.join(self._out_root, 'lpc_coeff_lsf_bins_updated_' + self._rand_model_id + '.npy'),
                                eng.view("lpc_quan/bins").detach().cpu().numpy())
            _quan_loss = float(terms["quan"][-1].mean().item()) if terms else nan
            if lpc:
                # end2end_eval_lpc :826-833: code-length weighted mean over [LSF, codec...] (16 : L : L)
                w = np.array([16.0] + [float(c.L) for c in eng.codecs])
                fully_entropy = float(np.sum(w / w.sum() * np.array([ent_lpc] + list(ents))))
            elif flag == 'finetune':
                fully_entropy = float(sum(ents))            # interested_var[5] = reduce_sum(ent_loss_arr), cmrl.py:357
            else:
                fully_entropy = ents[-1]                    # the newest codec's entropy (nsc_module:912; cmrl.py:98)
            fps = self._batch_size * (self._comm.world if self._comm else 1) * nsteps / max(elapsed, 1e-9)
            if lpc:
                print('Epoch %3d: SNR: %7.5f dB    STOI: %6.5f    PESQ: %6.5f   linearity: %6.5f  modelid: %s  _quan_loss: '
                      '%6.5f,  fully_entropy: %6.5f , time: %.3f, tau: %.3f' % (
                          i, ave_snr, nan, nan, nan, self._rand_model_id, _quan_loss, fully_entropy, elapsed, init_tau))
                self._write_to_file_and_update_to_display(
                    journal_line_lpc(i, ave_snr, nan, nan, _quan_loss, 0.0, 0.0, fully_entropy))
            else:
                print('Epoch %3d: SNR: %7.5f dB  Si-SNR: %7.5f dB  STOI: %6.5f    PESQ: %6.5f   linearity: %6.5f  modelid: %s  '
                      '_quan_loss: %6.5f,  fully_entropy: %6.5f , time: %.3f, tau: %.3f, tau_1: %.3f, tau_2: %.3f' % (
                          i, ave_snr, ave_si_snr, nan, nan, nan, self._rand_model_id, _quan_loss, fully_entropy, elapsed,
                          init_tau, init_tau_1, init_tau_2))
            print('frames/s: %.1f' % fps)
            # ---- tau controller (nsc_module:494-517; LPC variant :630-639) ----
            init_tau, init_tau_1, init_tau_2 = tau_controller(flag, lpc, is_quan_on_val, (init_tau, init_tau_1, init_tau_2),
                                                              fully_entropy, ents, self._target_entropy)
            if lpc:
                if is_quan_on_val == 1.0:
                    print('tau:', init_tau)
            else:
                if flag == 'finetune':
                    self._write_to_file_and_update_to_display('codec-1: ' + str(np.array(ents[:1])))
                    self._write_to_file_and_update_to_display('codec-2: ' + str(np.array(ents[1:2] or ents[:1])))
                print('Tau: %7.5f, Tau_1: %7.5f, Tau_2: %7.5f' % (init_tau, init_tau_1, init_tau_2))
                self._write_to_file_and_update_to_display(
                    journal_line(i, ave_snr, ave_si_snr, nan, nan, _quan_loss, init_tau, fully_entropy))
            if lpc and flag == 'finetuning_follower_all' and fully_entropy < self._target_entropy:   # nsc_module:641
                break
        if self._is_writer():
            self.save(eng, save_id)
        if self._dump_rows:
            _, rank = self._world_rank()
            np.save(os.path.join(self._out_root, f"rows_rank{rank}.npy"), np.array(self._rows_seen, dtype=np.int64))
        self._barrier()       # no rank restores the checkpoint before rank 0 has finished writing it
        self._last_taus = (init_tau, init_tau_1, init_tau_2)
        return init_tau

    def _loss_cfgs(self, num_codecs, mode):
        """Step configs of the two optimizers and where tau enters (nsc_module:914-926, 1032-1050; cmrl.py:95-113,
        348-372, 463-485) - checked against the reference's own loss vectors in tests/test_reference_exec.py.
        The engine minimises  sum_b [c_time*time + c_freq*freq + sum_i c_quan[i]*quan_i[b] + c_quan_lpc*quan_lpc[b]]
                              + Bglobal * (sum_i c_ent[i]*ent_i + c_ent_lpc*ent_lpc)."""
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
                       trainable=train, slot=0, quan_op=False)
        cq = list(zeros)
        tau_map = []
        extra = {}
        if mode == "single" and lpc:
            # nsc_module:1032-1050: quan and entropy terms of the LSF quantizer and the codec are ALWAYS blended by
            # their code lengths 16 : L; is_cq only decides whether the LSF quantizer's variables train (:993-995)
            L = 128.0 if self._the_strides[0] == 4 else 256.0
            a, b = 16.0 / (16.0 + L), L / (16.0 + L)
            cq[0] = c[2] * b
            tau_map = [("c_ent", 0, b, 0)]
            if self._is_cq:
                extra = dict(c_quan_lpc=c[2] * a, c_ent_lpc=0.0, train_lpc=True)
                tau_map.append(("c_ent_lpc", None, a, 0))
        elif mode in ("single", "follower"):
            k = num_codecs - 1 if mode == "follower" else 0
            cq[k] = c[2]
            tau_map = [("c_ent", k, 1.0, 0)]
        elif lpc:
            # cmrl.py:463-485: quan_loss(LSF) + sum_i quan_loss(codec i), every one times coeff[2]; NO entropy term;
            # lpc_quan/alpha, bins are created without `trainable=` (:398-401) -> always trained here
            cq = [c[2]] * num_codecs
            # no entropy term in this phase by construction (not a tau value): data-parallel runs skip the histogram all-reduce
            # in the middle of the step (the entropies this phase journals are then those of the local batch)
            extra = dict(c_quan_lpc=c[2], train_lpc=True, global_entropy=False)
        else:
            # cmrl.py:355: quantization_loss = tf.reduce_sum([quan_0, quan_1]) is a SCALAR (summed over the batch too)
            # that :361-365 broadcasts back into the [B] loss vector => its weight is coeff[2] * (global batch)
            gb = self._batch_size * (self._comm.world if self._comm else 1)
            cq = [c[2] * gb] * num_codecs
            tau_map = [("c_ent", i, 1.0, min(i + 1, 2)) for i in range(num_codecs)]
        quan = dict(is_quan_on=1.0, c_time=c[0], c_freq=c[1], c_quan=cq, c_ent=list(zeros), trainable=train, slot=1,
                    quan_op=True, **extra)
        if getattr(self, "_local_entropy", False):
            # --local_entropy: entropy_coding_loss sees each rank's own batch histogram - the standard
            # data-parallel behaviour, one all-reduce less per step; an N-rank run then no longer equals a 1-process run at
            # batch N B, and the entropies rank 0 journals are those of ITS frames
            quan["global_entropy"] = False
        return no_quan, quan, tau_map

    def one_ae(self):
        """nsc_module:891-939 (time domain) / :989-1073 (LPC): train codec 1, pretrain then quantised."""
        eng = self._make_engine(1, per_codec_list_semantics=self._is_pure_time_domain)
        self._engine = eng
        no_quan, quan, tau_map = self._loss_cfgs(1, "single")
        self.model_training(eng, no_quan, quan, self._learning_rate_tanh, self._epoch_tanh, 'pretrain', save_id='',
                            the_tau_val=self._coeff_term[3], tau_map=tau_map)
        return eng

    one_ae_lpc = one_ae
