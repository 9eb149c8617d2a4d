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
if os.environ.get("SN_MASK16"):      # experiments builds: SN_MASK16=0 keeps the fused mask head on k_mlp_wide_j<3> (A/B partner of k_mask16, mlp16.inc)
    from sanerf_hq_amd import _lib
    _lib.check(_lib.lib().sn_debug_set(b"mask_head16", int(os.environ["SN_MASK16"])), "debug_set")
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
chunks = [t[i] for i in range(int(os.environ.get("SN_TRACE_CHUNKS", "29")))]
layers = [t[128 + i] for i in range(3)]
print("k-step -> k-step:", [b - a for a, b in zip(chunks, chunks[1:])])
print("layer starts:", [l - layers[0] for l in layers], "all layers done at", t[160] - layers[0], "first k-step at", chunks[0] - layers[0])

if os.environ.get("SN_MASK16"):   # k_mask16: last tile's hand-over points (slots 140..143: narrow start / end, tile end, next tile top done -- the last two are from the tile before)
    n = int(os.environ.get("SN_TRACE_CHUNKS", "29"))
    print("hand-over: narrow start", t[140] - t[103], "after the last chunk's start; narrow", t[141] - t[140], "; tile end", t[142] - t[141],
          "; (tile 6 -> 7) top", t[143] - t[90], "after chunk 90's start")
