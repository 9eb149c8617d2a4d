#!/usr/bin/env python3
"""C3 only (400x400 + SAM-feature head) for rocprofv3 kernel traces."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import build, timeit  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402

dev = torch.device("cuda:0")
model = build(True, False, dev).eval()
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
def c3():
    with torch.no_grad():
        return model.render(ro, rd, staged=False, perturb=False, return_feats=1, H=H, W=W, tile_w=W)
print("C3 ms", timeit(c3, 3, 10) * 1e3)
