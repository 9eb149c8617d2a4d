#!/usr/bin/env python3
"""Two identical models stepped side by side (eager): parameter and gradient differences after every step."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import test_gpu_round4 as t4
from sanerf_hq_amd import ops
if os.environ.get("SN_FWD_BLAS"): ops.WIDE_MLP_FORWARD_NATIVE = False
gpu = torch.device("cuda:0")
m_e, s_e = t4._c5_like_step(gpu, 99, False)
m_g, s_g = t4._c5_like_step(gpu, 99, False)
for i in range(4):
    s_e(); s_g(); torch.cuda.synchronize()
    out = []
    for (n1, p1), (n2, p2) in zip(m_e.named_parameters(), m_g.named_parameters()):
        if p1.requires_grad:
            d = (p1.detach() - p2.detach()).abs(); dg = (p1.grad - p2.grad).abs()
            k = int(d.argmax())
            out.append(f"{n1.split('.')[0][:6]}{n1.split('.')[-2] if 'net' in n1 else ''}: dp {float(d.max()):.2e} (grad there {float(p1.grad.flatten()[k]):.2e} vs {float(p2.grad.flatten()[k]):.2e}) dgrad max {float(dg.max()):.2e}")
    print(f"step {i + 1}:", " | ".join(out), flush=True)
