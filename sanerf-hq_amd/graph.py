"""A training step as ONE HIP graph.

The steps of the path (mask-field step: frozen-field render -> m_grid -> mask MLP -> NLL -> backward -> Adam; RGB step: three stages
under autograd + proposal loss) are 50-150 launches of small kernels each; at 2-3 ms per step the launches themselves are a visible part
(RGB step: 2.4 ms of kernels in 2.8 ms of wall clock).  Every operator of this package is capturable: no host synchronisation inside a
step (host copies of offsets / aabb are memoised, the fp16 range guard only re-evaluates when a parameter VERSION moved), workspaces are
cached tensors, every kernel goes to torch's current stream, the binned grid backward has static grids, and
`sanerf_hq_amd.optim.Adam(capturable=True)` keeps its step count on the device.  `GraphedStep` captures `fn()` once and replays it.

    opt = Adam(params, lr=..., eps=1e-15, capturable=True)
    def step():
        opt.zero_grad(set_to_none=True)
        out = model.render(rays_o, rays_d, ...)          # rays_o / rays_d / targets: STATIC tensors, refilled with .copy_() between replays
        loss = ...
        loss.backward(); opt.step()
        return loss
    g = GraphedStep(step, warmup=3)
    for batch in loader:
        rays_o.copy_(batch.rays_o); ...
        loss = g()                                        # one graph launch; `loss` is the same tensor object every time

Random numbers drawn inside the step (perturb=True: torch.rand) advance correctly under replay (torch registers the generator with the
graph).  Anything that changes the step's SHAPE (number of rays, which parameters require gradients, update_proposal on / off) needs its
own GraphedStep.  The reference has no counterpart (its trainer launches eagerly, nerf/trainer.py:360-430)."""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 3):
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up on a side stream: allocations, plan caches, lazy initialisations
            for _ in range(max(warmup, 1)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = fn()

    def __call__(self):
        self.graph.replay()
        return self.result
