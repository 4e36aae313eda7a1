#!/usr/bin/env python3
"""CLI of the reference (main.py:5-47) on the MI355X hot path: the same 21 flags and modes '0'..'5'.

Dropped: the three blocking input() calls at exit (main.py:49-51).  Added (all optional, defaults reproduce the
reference's edit-the-source globals): --lpc_domain (constants.is_pure_time_domain=False), --data_root,
--max_batches_per_epoch, --out_root, --model_id, --seed, --local_entropy, --tf_checkpoint.  Multi-GPU: launch with torch.distributed.run; frames are
sharded over ranks and gradients all-reduced (sum) over RCCL.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def build_parser():
    parser = argparse.ArgumentParser(description='NSC / CMRL neural speech codec on MI355X (reference CLI).')
    parser.add_argument('--learning_rate_tanh', type=float, help='learning_rate for training tanh NN')
    parser.add_argument('--learning_rate_greedy_followers', type=str, help='learning_rate for training greedy followers NN')
    parser.add_argument('--epoch_tanh', type=int, help='epoch to train tanh NN.')
    parser.add_argument('--epoch_greedy_followers', type=str, help='epoch to fine tuning NN.')
    parser.add_argument('--from_where_step', type=int, help='0: from beginning; 1 from the second resnet or the first follower...')
    parser.add_argument('--batch_size', type=int, help='batch size.')
    parser.add_argument('--num_resnets', type=int, help='num_resnets.')
    parser.add_argument('--training_mode', type=str, help='How to train the NN.')
    parser.add_argument('--base_model_id', type=str, help='which model to re-train?')
    parser.add_argument('--suffix', type=str, help='save model name suffix..')
    parser.add_argument('--window_size', type=int, help='window_size')
    parser.add_argument('--bottleneck_kernel_and_dilation', type=str, help='bottleneck_kernel_and_dilation')
    parser.add_argument('--is_cq', type=int, help='is_cq')
    parser.add_argument('--the_strides', type=str, help='the_strides')
    parser.add_argument('--save_unique_mark', type=str, help='save_unique_mark')
    parser.add_argument('--coeff_term', type=str, help='coeff_term')
    parser.add_argument('--res_scalar', type=float, help='res_scalar')
    parser.add_argument('--pretrain_step', type=int, help='pretrain_step')
    parser.add_argument('--target_entropy', type=float, help='target_entropy')
    parser.add_argument('--num_bins_for_follower', type=str, help='num_bins_for_follower')
    # ---- additions ----
    parser.add_argument('--lpc_domain', action='store_true', help='LPC-residual (collaborative quantisation) path')
    parser.add_argument('--data_root', type=str, default=None, help='.npy of training frames; synthetic frames if absent')
    parser.add_argument('--max_batches_per_epoch', type=int, default=None, help='default 2500 like the reference')
    parser.add_argument('--val_data_root', type=str, default=None, help='.npy of validation frames (tau controller); synthetic if absent')
    parser.add_argument('--tau_from_validation', type=int, default=1,
                        help='1: tau follows the per-frame entropy of validation frames like the reference; 0: the last training batch')
    parser.add_argument('--out_root', type=str, default='.', help="where ./check and ./doc live")
    parser.add_argument('--model_id', type=str, default=None, help='fix the random model id')
    parser.add_argument('--seed', type=int, default=20200504, help='weights, and the per-epoch row order (identical on every rank)')
    parser.add_argument('--local_entropy', type=int, default=0,
                        help='data parallel: 1 = entropy term from each rank\'s own batch histogram (no histogram all-reduce)')
    parser.add_argument('--tf_checkpoint', type=int, default=0,
                        help='1: ALSO write every checkpoint in TensorFlow\'s V2 format (<name>.ckpt.index + .data-00000-of-00001), readable by the reference\'s Saver')
    parser.add_argument('--dump_rows', type=int, default=0, help='debug: write the training rows each rank fed to <out_root>/rows_rank<r>.npy')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args)
    from nsc_amd.cmrl import CMRL
    from nsc_amd.dist import Comm
    comm = Comm()
    args.comm = comm if comm.world > 1 else None
    import torch
    ldev = comm.local_rank % max(torch.cuda.device_count(), 1)      # one GPU per rank; wraps only in single-GPU tests
    args.device = "cuda:%d" % ldev
    torch.cuda.set_device(ldev)
    audio_coding_ae = CMRL(args)
    modes = {'1': 'one_ae', '2': 'retrain_from_somewhere', '3': 'cascaded', '4': 'cascaded', '5': 'finetune',
             '0': 'feedforward'}
    if args.training_mode in modes:
        audio_coding_ae.model(training_mode=modes[args.training_mode], arg=args)
    else:
        print('WRONG INPUT...')
    comm.close()


if __name__ == "__main__":
    main()
