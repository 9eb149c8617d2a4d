#!/usr/bin/env python3
"""Shader clock and socket power while the wide head MLP kernel (sn_mlp_wide_forward) runs back to back for several seconds:
is k_mlp_wide's distance from the matrix-core peak a clock (power) effect or idle matrix cores?
usage (GPU box): python tools/mlp_power.py [rows]"""
import os, re, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import raymarching as rm  # noqa: E402
from sanerf_hq_amd.nerf.network import SkipConnMLP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
samples = []
stop = False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().split("\n")
        if len(out) >= 2:
            f = out[1].split(",")
            try:
                samples.append((float(re.sub(r"[^0-9.]", "", f[5])), float(f[-1])))   # sclk MHz, W
            except (ValueError, IndexError):
                pass
        time.sleep(0.4)


for name, mlp, ln in (("mask 143-256-256-2", SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False), None),
                      ("samvit 163-256x4-256+LN", SkipConnMLP(163, 256, 256, 5, skip_layers=[2], bias=True), torch.nn.LayerNorm(256))):
    mlp = mlp.to(dev); ln = ln.to(dev) if ln is not None else None
    x = torch.randn(N, mlp.dim_in, device=dev)
    for _ in range(3):
        rm.mlp_forward(x, mlp, ln)
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < 6.0:
        for _ in range(20):
            rm.mlp_forward(x, mlp, ln)
        torch.cuda.synchronize(); it += 20
    dt = (time.perf_counter() - t0) / it
    stop = True; th.join()
    busy = [s for s in samples if s[1] > 0.8 * max(p for _, p in samples)]
    macs = sum(l.weight.numel() for l in mlp.net) * N
    clk = sorted(c for c, _ in busy)[len(busy) // 2]; pw = sorted(p for _, p in busy)[len(busy) // 2]
    chunks = sum((-(-l.weight.shape[1] // 16)) for l in mlp.net)          # k-steps of 16 over all layers (256 outputs each: 24 MFMAs of 32 cycles)
    floor_us = chunks * 24 * 32 / (clk * 1e6) * 1e6                       # matrix-pipe time of one 128-row workgroup at the sampled clock
    wg_us = dt * 1e6 / (N / 128 / 256)
    print(f"{name}: N={N} {dt * 1e3:.3f} ms/call, {2 * macs / dt / 1e12:.0f} TFLOP/s fp32-equivalent | sclk {clk:.0f} MHz, {pw:.0f} W | "
          f"per 128-row workgroup {wg_us:.1f} us vs matrix-pipe floor {floor_us:.1f} us at that clock ({floor_us / wg_us:.0%})")
