"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The path shards over frames (independent units).  Collectives per step:
  * gradient all-reduce, SUM, over the trainable range of the flat fp32 gradient buffer: by default ONE message at the
    tail of the step (0.7 - 2.8 MB; frozen scopes in front of the first trainable one are not sent); with
    engine.dp_overlap one message per trainable scope, issued as soon as that codec's backward pass and weight gradients
    are done (measured slower on the compute side: DESIGN section 6);
  * optional all-reduce (SUM) of the tiny soft-assignment histograms so entropy_coding_loss sees the global batch.
On CPU (tests) the same code runs over the gloo backend.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend = backend or os.environ.get("NSC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            self.backend = backend
            if torch.cuda.is_available():
                ndev = torch.cuda.device_count()
                lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
                if backend == "nccl" and ndev < lws:
                    # RCCL needs one GPU per rank: two ranks on one device fail deep inside the first collective
                    raise RuntimeError(f"nsc_amd.dist: backend nccl (RCCL) needs one GPU per local rank, but {lws} local ranks see "
                                       f"{ndev} device(s) (LOCAL_RANK {self.local_rank}); start fewer ranks, or set "
                                       f"NSC_DIST_BACKEND=gloo to share a GPU between ranks (tests only)")
                torch.cuda.set_device(self.local_rank % ndev)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def preflight(self, device):
        """First contact with the backend: one tiny all-reduce whose result is checked, with an error that names the backend,
        the rank and the device instead of a stack trace from inside the first gradient exchange."""
        if self.world == 1:
            return
        try:
            t = torch.full((4,), float(self.rank + 1), dtype=torch.float32, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            got, want = float(t[0].item()), self.world * (self.world + 1) / 2.0
        except Exception as ex:
            raise RuntimeError(f"nsc_amd.dist: pre-flight all-reduce failed on rank {self.rank}/{self.world} (backend {self.backend}, "
                               f"device {device}, LOCAL_RANK {self.local_rank}): {type(ex).__name__}: {ex}") from ex
        if got != want:
            raise RuntimeError(f"nsc_amd.dist: pre-flight all-reduce returned {got}, expected {want} (rank {self.rank}/{self.world}, "
                               f"backend {self.backend}, device {device})")

    def allreduce(self, t, op=dist.ReduceOp.SUM):
        if self.world > 1:
            dist.all_reduce(t, op=op)
        return t

    def allreduce_async(self, t, op=dist.ReduceOp.SUM):
        """Starts the all-reduce and returns a handle whose .wait() orders the caller's stream after it (nccl/RCCL: the
        collective runs on the process group's own stream, so kernels launched meanwhile overlap with it)."""
        if self.world > 1:
            return dist.all_reduce(t, op=op, async_op=True)
        return None

    def allreduce_list(self, tensors):
        for t in tensors:
            self.allreduce(t)

    def broadcast_object(self, obj, src=0):
        if self.world > 1:
            box = [obj]
            dist.broadcast_object_list(box, src=src)
            return box[0]
        return obj

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max_float(self, v, device):
        t = torch.tensor([float(v)], dtype=torch.float64, device=device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shard(self, n_global):
        """Frame range [lo, hi) of this rank for a global batch of n_global frames (contiguous blocks)."""
        per = n_global // self.world
        return self.rank * per, (self.rank + 1) * per

    def close(self):
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()
