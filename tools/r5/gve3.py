#!/usr/bin/env python3
"""In-situ check of the native training forward: every call's (x, weights, y) recorded, y recomputed in fp64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import test_gpu_round4 as t4
from sanerf_hq_amd import ops
gpu = torch.device("cuda:0")
rec = []
orig = ops.wide_mlp_train
def wrapped(x, layers, leaky):
    y = orig(x, layers, leaky)
    rec.append((x.detach().clone(), [l.weight.detach().clone() for l in layers], y.detach().clone(), leaky))
    return y
ops.wide_mlp_train = wrapped
import sanerf_hq_amd.nerf.network as net
for mod in list(sys.modules.values()):
    if mod and getattr(mod, "wide_mlp_train", None) is orig: mod.wide_mlp_train = wrapped
m_e, s_e = t4._c5_like_step(gpu, 99, False)
m_g, s_g = t4._c5_like_step(gpu, 99, False)
for i in range(3):
    s_e(); s_g()
torch.cuda.synchronize()
print("calls recorded:", len(rec))
for k, (x, ws, y, leaky) in enumerate(rec):
    h = x.double().reshape(-1, x.shape[-1])
    for i, w in enumerate(ws):
        h = torch.nn.functional.linear(h, w.double())
        if i + 1 < len(ws): h = torch.nn.functional.leaky_relu(h) if leaky else torch.relu(h)
    d = (y.double().reshape(h.shape) - h).abs()
    print(f"call {k}: x {tuple(x.shape)} contiguous {x.is_contiguous()} | y vs fp64: max abs {float(d.max()):.3e} (scale {float(h.abs().mean()):.3e}) rows off by > 1e-4: {int((d.max(dim=1).values > 1e-4).sum())}")
for k in range(0, len(rec) - 1, 2):
    (xe, we, ye, _), (xg, wg, yg, _) = rec[k], rec[k + 1]
    print(f"step {k // 2 + 1}: x e-vs-g max {float((xe - xg).abs().max()):.3e} rows differing {int(((xe - xg).abs().reshape(-1, xe.shape[-1]).max(dim=1).values > 0).sum())} | y e-vs-g max {float((ye - yg).abs().max()):.3e} | weights differ: {[float((a - b).abs().max()) for a, b in zip(we, wg)]}")
