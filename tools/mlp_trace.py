#!/usr/bin/env python3
"""Per-chunk shader cycles of k_mlp_wide (workgroup 0, wave 0) from a -DSN_WIDE_TRACE=1 build of mlp.hip:
usage (GPU box): SN_LIB=ab/wtrace.so python tools/mlp_trace.py"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import _lib, raymarching as rm  # noqa: E402
from sanerf_hq_amd.nerf.network import SkipConnMLP  # noqa: E402
from sanerf_hq_amd.gridencoder import GridEncoder  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = C.CDLL(os.environ["SN_LIB"])
for name, mlp, ln in (("mask 143-256-256-2", SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False), None),
                      ("samvit", SkipConnMLP(163, 256, 256, 5, skip_layers=[2], bias=True), torch.nn.LayerNorm(256))):
    mlp = mlp.to(dev); ln = ln.to(dev) if ln is not None else None
    x = torch.randn(1 << 20, mlp.dim_in, device=dev)
    for _ in range(3):
        rm.mlp_forward(x, mlp, ln)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    lib.sn_mlp_wide_debug_trace(buf, 256)
    t = list(buf)
    chunks = [t[i] for i in range(128) if t[i]]
    layers = [t[128 + i] for i in range(8) if t[128 + i]]
    end = t[160]
    d = [b - a for a, b in zip(chunks, chunks[1:])]
    print(name, "chunks", len(chunks), "cycles chunk->chunk:", d)
    print("   layer starts relative to first:", [l - layers[0] for l in layers], "all layers done at", end - layers[0], "first chunk at", chunks[0] - layers[0])


def report(name):
    buf = (C.c_ulonglong * 256)()
    lib.sn_mlp_wide_debug_trace(buf, 256)
    t = list(buf)
    chunks = [t[i] for i in range(128) if t[i]]
    layers = [t[128 + i] for i in range(8) if t[128 + i]]
    d = [b - a for a, b in zip(chunks, chunks[1:])]
    print(name, "chunks", len(chunks), "cycles chunk->chunk:", d)
    print("   layer starts relative to first:", [l - layers[0] for l in layers], "all layers done at", t[160] - layers[0], "first chunk at", chunks[0] - layers[0])


# fused mask head (k_mlp_wide<3> / k_mlp_wide_j<3>): m_grid L=16 C=8 T=2^19 + 15 geometry channels -> 143-256-256-2, T = 32 samples per ray
enc = GridEncoder(input_dim=3, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
mlp = SkipConnMLP(16 * 8 + 15, 2, 256, 3, skip_layers=[], bias=False).to(dev)
N, T = 32768, 32
xyz = torch.rand(N, T, 3, device=dev) * 2 - 1
extra = torch.randn(N, T, 15, device=dev)
w = torch.rand(N, T, device=dev)
for _ in range(3):
    rm.mask_head(w, xyz, extra, enc, mlp, 1.0)
torch.cuda.synchronize()
report("fused mask head")
