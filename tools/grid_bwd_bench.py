#!/usr/bin/env python3
"""Embedding-gradient scatter of the grid encoder at the sizes of the training steps: binned kernels (grid_binned.hip) vs the
reference-style atomic kernel; positions are samples along random rays (coherent on coarse levels, like a training batch).
usage (GPU box): python tools/grid_bwd_bench.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as orc  # noqa: E402
from sanerf_hq_amd import ops  # noqa: E402
from sanerf_hq_amd.gridencoder import grid_encode  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("SN_BIN_PULL") is not None:          # experiments build: entries as products (0) / references (1) whatever C
    from sanerf_hq_amd import _lib as _l
    _l.check(_l.lib().sn_debug_set(b"bin_pull", int(os.environ["SN_BIN_PULL"])), "sn_debug_set")
CASES = [("mask / SAM grid  C=8 L=16 T=2^19, 4096 rays x 32", dict(L=16, C=8, log2T=19, desired=512), 4096, 32),
         ("main grid        C=2 L=16 T=2^19, 4096 rays x 32", dict(L=16, C=2, log2T=19, desired=4096), 4096, 32),
         ("proposal grid 0  C=2 L=5  T=2^17, 4096 rays x 128", dict(L=5, C=2, log2T=17, desired=128), 4096, 128),
         ("proposal grid 1  C=2 L=5  T=2^17, 4096 rays x 64", dict(L=5, C=2, log2T=17, desired=256), 4096, 64),
         ("mask grid, samples piled near a surface (a trained field's last stage)", dict(L=16, C=8, log2T=19, desired=512, surface=True), 4096, 32)]
for name, cfg, R, T in CASES:
    rng = np.random.default_rng(3)
    offs, pls = orc.grid_layout(3, cfg["L"], cfg["C"], 2, 16, cfg["log2T"], cfg["desired"])
    o = rng.uniform(0.1, 0.9, (R, 1, 3)); d = rng.normal(size=(R, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = np.sort(rng.uniform(-0.5, 0.5, (R, T, 1)), axis=1)
    if cfg.pop("surface", False):          # rays from one side onto a sphere of radius 0.2: 32 samples within +-0.01 of the hit
        o = np.array([0.5, 0.5, 0.02]) + rng.normal(0, 0.002, (R, 1, 3)); tgt = np.array([0.5, 0.5, 0.5]) + rng.uniform(-0.18, 0.18, (R, 1, 3)); tgt[..., 2] = 0.5
        d = tgt - o; d /= np.linalg.norm(d, axis=-1, keepdims=True)
        bq = np.sum((o - 0.5) * d, -1, keepdims=True); cq = np.sum((o - 0.5) ** 2, -1, keepdims=True) - 0.04
        th = -bq - np.sqrt(np.maximum(bq * bq - cq, 0.0))
        t = th + np.sort(rng.normal(0, 0.004, (R, T, 1)), axis=1)
    x = np.clip(o + d * t, 0.0, 1.0).reshape(-1, 3).astype(np.float32)
    B = x.shape[0]
    xt = torch.from_numpy(x).to(dev)
    emb = torch.zeros(int(offs[-1]), cfg["C"], device=dev).uniform_(-1e-4, 1e-4).requires_grad_(True)
    offt = torch.from_numpy(np.asarray(offs, dtype=np.int32)).to(dev)
    g = torch.randn(B, cfg["L"] * cfg["C"], device=dev)
    res = {}
    line = f"{name}: B={B}, pairs={B * cfg['L'] * 8 / 1e6:.1f} M |"
    for mode in ("binned", "atomic"):
        ops.GRID_BACKWARD_MODE = mode
        out = grid_encode(xt, emb, offt, pls, 16, False, 0, False, 0)

        def bwd():
            emb.grad = None
            out.backward(g, retain_graph=True)
        for _ in range(3):
            bwd()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            bwd()
        ev[1].record()
        torch.cuda.synchronize()
        res[mode] = emb.grad.clone()
        line += f" {mode} {ev[0].elapsed_time(ev[1]) / 10:.3f} ms (incl. the zero-fill of the gradient table)"
    ops.GRID_BACKWARD_MODE = "auto"
    rel = float((res["binned"] - res["atomic"]).double().norm() / res["atomic"].double().norm())
    same = bool(torch.equal(res["binned"].abs().sum(-1) > 0, res["atomic"].abs().sum(-1) > 0))
    print(line + f" | rel-L2 diff {rel:.2e}, same rows touched: {same}")
