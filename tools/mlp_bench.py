#!/usr/bin/env python3
"""Timing of the wide head MLP kernel (sn_mlp_wide_forward) against the torch module (rocBLAS/hipBLASLt + elementwise)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import raymarching as rm  # noqa: E402
from sanerf_hq_amd.nerf.network import SkipConnMLP  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = {"samvit 163-256x4-256 +LN, skip@2": (SkipConnMLP(163, 256, 256, 5, skip_layers=[2], bias=True), torch.nn.LayerNorm(256), 160000),
         "mask 143-256-256-2": (SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False), None, 4096 * 32)}
for name, (mlp, ln, N) in cases.items():
    mlp = mlp.to(dev); ln = ln.to(dev) if ln is not None else None
    x = torch.randn(N, mlp.dim_in, device=dev)
    def fused(): return rm.mlp_forward(x, mlp, ln)
    def eager():
        with torch.no_grad():
            y = mlp(x)
            return ln(y) if ln is not None else y
    res = {}
    for tag, fn in (("fused", fused), ("torch", eager)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); res[tag] = (time.perf_counter() - t0) / 20 * 1e3
    macs = sum(l.weight.numel() for l in mlp.net) * N
    print(f"{name}: N={N} fused {res['fused']:.3f} ms ({2 * macs / res['fused'] / 1e9:.1f} TFLOP/s fp32-equivalent), torch {res['torch']:.3f} ms, "
          f"max|diff| {float((fused() - eager()).abs().max()):.2e}")
