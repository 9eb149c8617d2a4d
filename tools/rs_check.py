"""Role-split final stage (SN_RENDER_RS=1) against the default kernels: bit-for-bit comparison of image / depth / weights_sum /
f_image on several image shapes and schedules, both table precisions, then same-process timing of both.
usage (GPU box, repo root): python tools/rs_check.py [--time-only] [--hw 800]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sanerf_hq_amd import raymarching as rm, synth

ap = argparse.ArgumentParser()
ap.add_argument("--time-only", action="store_true")
ap.add_argument("--hw", type=int, default=800)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda:0")
pose = synth.orbit_pose(1.0, 20.0, 30.0)


def render(plan, ro, rd, W, rs, want=("f_image",)):
    os.environ["SN_RENDER_RS"] = "1" if rs else "0"
    out = rm.render_rays(plan, ro, rd, tile_w=W, want=want)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in out.items()}


bad = 0
if not args.time_only:
    for steps in ([128], [128, 64, 32], [7], [33, 17, 9]):
        params = synth.synthetic_params(steps, seed=23)
        model = synth.product_model(params, steps, False, dev)
        for tdt in (torch.float32, torch.float16):
            plan = rm.RenderPlan(model, steps, tdt)
            for (H, W) in ((64, 64), (48, 80), (200, 104), (16, 32), (40, 24)):
                intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
                ro, rd = rm.generate_rays(pose, intr, H, W, device=dev)
                for tile in (W, 0):
                    a = render(plan, ro, rd, tile, False)
                    b = render(plan, ro, rd, tile, True)
                    for k in ("image", "depth", "weights_sum", "f_image"):
                        same = torch.equal(a[k], b[k])
                        if not same:
                            bad += 1
                            d = (a[k] - b[k]).abs()
                            print(f"DIFF steps={steps} {tdt} {H}x{W} tile={tile} {k}: max {float(d.max()):.3e}, {int((d > 0).sum())} of {d.numel()} differ")
    print("bitwise comparison:", "all equal" if bad == 0 else f"{bad} tensors differ")

H = W = args.hw
intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
ro, rd = rm.generate_rays(pose, intr, H, W, device=dev)
for steps in ([128], [128, 64, 32]):
    params = synth.synthetic_params(steps, seed=0)
    model = synth.product_model(params, steps, False, dev)
    for tdt in (torch.float32, torch.float16):
        plan = rm.RenderPlan(model, steps, tdt)
        res = {}
        for rs in (0, 1, 0, 1):
            os.environ["SN_RENDER_RS"] = str(rs)
            out = {}
            for _ in range(5):
                rm.render_rays(plan, ro, rd, tile_w=W, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                rm.render_rays(plan, ro, rd, tile_w=W, out=out)
            torch.cuda.synchronize()
            res.setdefault(rs, []).append((time.perf_counter() - t0) / args.iters * 1e3)
        print(f"{H}x{W} steps={steps} {str(tdt).split('.')[-1]}: default {min(res[0]):.3f} ms, role-split {min(res[1]):.3f} ms  ({[round(x, 3) for x in res[0] + res[1]]})")
sys.exit(1 if bad else 0)
