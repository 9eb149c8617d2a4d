import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
from bench_configs import build
from sanerf_hq_amd import raymarching as rm, synth
dev = torch.device("cuda:0")
model = build(False, True, dev).eval()
for H in (64, 400):
    W = H
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)
    outs = []
    for i in range(4):
        with torch.no_grad():
            o = model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)
        outs.append(o["instance_mask_logits"].clone())
    model.fused_mask_head = False
    with torch.no_grad():
        ref = model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)["instance_mask_logits"].clone()
    model.fused_mask_head = True
    for i in range(4):
        d = (outs[i] - ref).abs().max(dim=-1).values
        bad = torch.nonzero(d > 1e-4).flatten()
        print(H, "run", i, "max diff vs unfused", float(d.max()), "bad rays", bad.numel(), "first", bad[:16].tolist(), "ray%16 hist", torch.bincount(bad % 16, minlength=16).tolist() if bad.numel() else None)
        if i:
            dd = (outs[i] - outs[0]).abs().max(dim=-1).values
            print("   vs run0: max", float(dd.max()), "differing rays", int((dd > 0).sum()))
