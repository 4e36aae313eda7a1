"""Replaying an op-surface step from a hipGraph.

A step written with nn_core_operator / loss_terms_and_measures under torch autograd is host-bound when launched op by op (2.4-2.9 ms
for the one-codec step whose kernels take 1.4 ms: bench.py key `op_surface`).  No op of the surface synchronises with the host, so the
whole step - forward, loss, backward, the batched weight-gradient launches at its end - can be captured once and replayed; what the
kernels derive from the parameters (images, flipped kernels) is gathered INSIDE the captured step, so a replay follows parameter
updates made between replays (optimizer steps on the same tensors).  The reference has no counterpart (a TF1 session runs its graph
itself); this is the torch.cuda.graph recipe with the warm-up the surface needs (index maps, workspaces and the image set are built
by eager runs, never under capture)."""
from __future__ import annotations

import torch


def capture_step(step_fn, warmup=2):
    """Returns replay() for `step_fn`, a callable without arguments that runs one whole step on static tensors (inputs it reads and
    `.grad`s / outputs it writes keep their addresses: copy new data INTO the input tensors before replay()).  step_fn must call
    store.begin_pass() itself if it re-traces a builder, and must not read values back to the host (.item(), float(), print)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, int(warmup))):
            step_fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step_fn()
    return _Replay(graph)           # (keeps the graph - and the memory pool its tensors live in - alive with the callable)


class _Replay:
    def __init__(self, graph):
        self.graph = graph

    def __call__(self):
        self.graph.replay()
