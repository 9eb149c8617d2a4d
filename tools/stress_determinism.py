"""Repeat every fused path many times on identical inputs and require bit-equal outputs (hunts rare races: a missing
wait on an asynchronous LDS copy once showed as wrong rows in one run out of five).  GPU box: python tools/stress_determinism.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_opt, product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402


def repeat(name, fn, reps):
    first = {k: v.clone() for k, v in fn().items() if torch.is_tensor(v)}
    bad = 0
    for _ in range(reps):
        out = fn()
        if not all(torch.equal(out[k], first[k]) for k in first):
            bad += 1
    print(f"{name}: {bad} of {reps} repetitions differ")
    return bad


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    bad = 0
    for steps in ([128], [128, 64, 32]):
        model = product_model(synthetic_params(steps, seed=0), steps, False, dev)
        for H in (800, 400):
            ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, H), H, H, device=dev)
            for dt in (torch.float32, torch.float16):
                plan = rm.RenderPlan(model, steps, dt)
                bad += repeat(f"render {H}x{H} {steps} {dt}", lambda: rm.render_rays(plan, ro, rd, tile_w=H, out={}), reps)
        pix = torch.from_numpy((synth.hash_u01(4096, 99) * (400 * 400)).astype(np.int64)).to(dev)
        plan = rm.RenderPlan(model, steps)
        bad += repeat(f"render 4096 rays linear {steps}", lambda: rm.render_rays(plan, ro[pix].contiguous(), rd[pix].contiguous(), tile_w=0, out={}), reps)
    steps = [128, 64, 32]
    heads = product_model(synthetic_params(steps, heads=True, seed=1), steps, True, dev)
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(400, 400), 400, 400, device=dev)

    def c3():
        with torch.no_grad():
            return heads.render(ro, rd, staged=False, perturb=False, return_feats=1, H=400, W=400, tile_w=400)
    bad += repeat("C3 400x400 + SAM head", c3, reps)

    def mask():
        with torch.no_grad():
            return heads.render(ro[:40000].contiguous(), rd[:40000].contiguous(), staged=False, perturb=False, return_mask=1, H=100, W=400, tile_w=400)
    bad += repeat("mask head 100x400", mask, max(5, reps // 5))
    print("TOTAL differing repetitions:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
