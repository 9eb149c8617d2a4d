#!/usr/bin/env python3
"""Per-k-step shader cycles of the fused mask head inside a real 400x400 mask render (samples in [ray][t] order along the rays):
usage (GPU box): SN_LIB=ab/wtrace.so python tools/mask_trace.py      (-DSN_WIDE_TRACE=1 -DSN_WIDE_TRACE_MID=1 build)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import build, timeit  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
lib = C.CDLL(os.environ["SN_LIB"])
model = build(False, True, dev).eval()
H = W = 400
ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
def run():
    with torch.no_grad():
        return model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)
print("mask render ms", timeit(run, 3, 10) * 1e3)
buf = (C.c_ulonglong * 256)()
lib.sn_mlp_wide_debug_trace(buf, 256)
t = list(buf)
chunks = [t[i] for i in range(29)]
layers = [t[128 + i] for i in range(3)]
print("k-step -> k-step:", [b - a for a, b in zip(chunks, chunks[1:])])
print("layer starts:", [l - layers[0] for l in layers], "all layers done at", t[160] - layers[0], "first k-step at", chunks[0] - layers[0])
