#!/usr/bin/env python3
"""Two identical models, one backward pass each, one after the other: are the table gradients equal, and which one matches the atomic path?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import test_gpu_round4 as t4
from sanerf_hq_amd import ops, raymarching as rm
gpu = torch.device("cuda:0")
def grads(model_step):
    model, step = model_step
    # one forward + backward without the optimiser: re-create what step() does up to loss.backward()
    return model
import types
def one_pass(seed=99):
    from helpers import make_opt
    model, step = t4._c5_like_step(gpu, seed, False)
    # monkeypatch: run the step but read gradients before they are consumed (Adam does not clear them)
    loss = step()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad and p.grad is not None}
mode = sys.argv[1] if len(sys.argv) > 1 else "binned"
if len(sys.argv) > 2 and sys.argv[2] == "blas":
    ops.WIDE_MLP_FORWARD_NATIVE = False
ops.GRID_BACKWARD_MODE = "atomic"; ref = one_pass()
ops.GRID_BACKWARD_MODE = mode
a = one_pass(); b = one_pass(); c = one_pass()
for n in a:
    r = ref[n]
    da, db, dc = (a[n] - r).abs(), (b[n] - r).abs(), (c[n] - r).abs()
    print(n, "| vs atomic: max", float(da.max()), float(db.max()), float(dc.max()), "| a vs b: max", float((a[n] - b[n]).abs().max()), "n differing", int((a[n] != b[n]).sum()),
          "| grad scale", float(r.abs().max()))
    if n.startswith("m_grid"):
        bad = (da > 1e-9 + 1e-4 * r.abs()).nonzero()
        print("   elements of a off by more than 1e-4 relative:", bad.shape[0], bad[:8].tolist())
