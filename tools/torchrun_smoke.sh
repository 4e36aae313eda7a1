# exercises bench.py's N>1 control flow (barriers, max-over-ranks timing, gradient all-reduce) with 2 ranks sharing ONE GPU
# over gloo (NCCL/RCCL needs one GPU per rank; the driver's multi-GPU runs use nccl)
NSC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-infer 2>&1 | tail -3 | cut -c1-400
