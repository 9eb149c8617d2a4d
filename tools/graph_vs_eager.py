#!/usr/bin/env python3
"""Where do a graphed and an eager mask-field training run part?  Parameters after every step, side by side."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import test_gpu_round4 as t4
from sanerf_hq_amd.graph import GraphedStep
gpu = torch.device("cuda:0")
if os.environ.get("SN_GRID_ATOMIC"):
    from sanerf_hq_amd import ops as _o
    _o.GRID_BACKWARD_MODE = "atomic"
if os.environ.get("SN_NO_FUSED_BWD"):
    from sanerf_hq_amd import ops as _o2
    _o2.WIDE_MLP_BACKWARD_FUSED = False
if os.environ.get("SN_FWD_BLAS"):
    from sanerf_hq_amd import ops
    ops.WIDE_MLP_FORWARD_NATIVE = False          # A/B: the training MLP's forward through BLAS
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
m_e, s_e = t4._c5_like_step(gpu, 99, False)
m_g, s_g = t4._c5_like_step(gpu, 99, mode != "eager2")
def diff(tag):
    torch.cuda.synchronize()
    out = []
    for (n1, p1), (n2, p2) in zip(m_e.named_parameters(), m_g.named_parameters()):
        if p1.requires_grad:
            d = (p1 - p2).abs()
            out.append(f"{n1}: max {float(d.max()):.3e} n>1e-6 {int((d > 1e-6).sum())}")
    print(tag, " | ".join(out), flush=True)
diff("init")
for i in range(2):
    s_e()
if mode == "graph":
    g = GraphedStep(s_g, warmup=2)
else:
    for i in range(2): s_g()
    g = s_g
diff("after 2")
for i in range(4):
    le = s_e(); lg = g()
    diff(f"after {3 + i} loss {float(le):.7f} {float(lg):.7f}")
