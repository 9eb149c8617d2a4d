#!/usr/bin/env python3
"""Two identical forward + backward passes of the training MLP: do the gradients repeat?  native / BLAS forward x side stream on / off."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import ops, synth
gpu = torch.device("cuda:0")
N, din, n_out = 131072, 143, 2
ws = [torch.from_numpy(synth.linear_weight(o, i, 900 + k, 2.0)).to(gpu) for k, (o, i) in enumerate([(256, din), (256, 256), (n_out, 256)])]
x = torch.randn(N, din, device=gpu)
gy = torch.randn(N, n_out, device=gpu) * 10.0 ** torch.empty(N, 1, device=gpu).uniform_(-6, -1)
def run():
    xs = x.clone().requires_grad_(True)
    wl = [w.clone().requires_grad_(True) for w in ws]
    y = ops._wide_mlp_train.apply(xs, True, *wl)
    y.backward(gy)
    torch.cuda.synchronize()
    return [y.detach(), xs.grad] + [w.grad for w in wl]
for native in (True, False):
    for side in (False, True):
        ops.WIDE_MLP_FORWARD_NATIVE, ops.WGRAD_SIDE_STREAM = native, side
        a = run()
        worst = [0.0] * len(a)
        for _ in range(6):
            junk = torch.full((1 << 22,), float("nan"), device=gpu); del junk
            b = run()
            worst = [max(w, float((p - q).abs().max() / q.abs().max())) for w, p, q in zip(worst, a, b)]
        print(f"native={native} side_stream={side}: max rel diff over 6 repeats  y {worst[0]:.2e}  gx {worst[1]:.2e}  gw {[f'{v:.2e}' for v in worst[2:]]}")
