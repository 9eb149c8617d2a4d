"""Linear-tail final stage (the default) against the per-sample form and the oracle: max differences, then timing.
usage (GPU box, repo root): python tools/lt_check.py [--hw 800]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from sanerf_hq_amd import raymarching as rm, synth

ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, default=800)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--field", default="per_sample_form", help="Tuning field that differs between the A side (1) and the B side (0)")
args = ap.parse_args()
dev = torch.device("cuda:0")
pose = synth.orbit_pose(1.0, 20.0, 30.0)
import oracle as orc
from helpers import oracle_cfg

for steps in ([128], [128, 64, 32], [7]):
    params = synth.synthetic_params(steps, seed=23)
    model = synth.product_model(params, steps, False, dev)
    for tdt in (torch.float32, torch.float16):
        plan = rm.RenderPlan(model, steps, tdt)
        for (H, W) in ((64, 64), (48, 80), (40, 24)):
            intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
            ro, rd = rm.generate_rays(pose, intr, H, W, device=dev)
            outs = {}
            for v in ("0", "1"):
                setattr(rm.tuning, args.field, 1 if v == "0" else 0)
                o = rm.render_rays(plan, ro, rd, tile_w=W, want=("f_image",), out={})
                torch.cuda.synchronize()
                outs[v] = {k: t.clone() for k, t in o.items()}
            want = orc.render(oracle_cfg(orc, params, steps, table_f16=(tdt == torch.float16)), ro.cpu().numpy(), rd.cpu().numpy())
            d = {k: float((outs["0"][k] - outs["1"][k]).abs().max()) for k in ("image", "depth", "weights_sum", "f_image")}
            e = {v: float(np.abs(outs[v]["image"].cpu().numpy() - want["image"]).max()) for v in ("0", "1")}
            print(f"steps={steps} {str(tdt).split('.')[-1]} {H}x{W}: LT vs per-sample {d} | vs oracle: per-sample {e['0']:.2e}, LT {e['1']:.2e}")

H = W = args.hw
intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
ro, rd = rm.generate_rays(pose, intr, H, W, device=dev)
for steps in ([128], [128, 64, 32]):
    params = synth.synthetic_params(steps, seed=0)
    model = synth.product_model(params, steps, False, dev)
    for tdt in (torch.float32, torch.float16):
        plan = rm.RenderPlan(model, steps, tdt)
        res = {}
        for v in (0, 1, 0, 1):
            setattr(rm.tuning, args.field, 1 if v == 0 else 0)
            out = {}
            for _ in range(5):
                rm.render_rays(plan, ro, rd, tile_w=W, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                rm.render_rays(plan, ro, rd, tile_w=W, out=out)
            torch.cuda.synchronize()
            res.setdefault(v, []).append((time.perf_counter() - t0) / args.iters * 1e3)
        print(f"{H}x{W} steps={steps} {str(tdt).split('.')[-1]}: default {min(res[0]):.3f} ms, {args.env}=1 {min(res[1]):.3f} ms  ({[round(x, 3) for x in res[0] + res[1]]})")
