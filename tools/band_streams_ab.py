#!/usr/bin/env python3
"""Does rendering an image as row bands on SEPARATE streams recover the tail of small launches?  (400x400: 625 workgroups on 512 slots)
One call on one stream vs nb bands (own plan = own workspace each) on one stream vs on nb streams.  usage: band_streams_ab.py [H] [schedule]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")): sys.path.insert(0, p)
from helpers import synthetic_params  # noqa: E402
from sanerf_hq_amd.synth import product_model  # noqa: E402
from sanerf_hq_amd import raymarching as rm, synth  # noqa: E402
dev = torch.device("cuda:0")
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sch = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128,64,32").split(",")]
ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=dev)


def timeit(fn, warm=5, iters=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for f16 in (False, True):
    model = product_model(synthetic_params(sch, seed=1), sch, False, dev)
    td = torch.float16 if f16 else torch.float32
    plan = rm.RenderPlan(model, sch, td)
    base = rm.render_rays(plan, ro, rd, tile_w=W)["image"].clone()
    t1 = timeit(lambda: rm.render_rays(plan, ro, rd, tile_w=W))
    print(f"{H}x{W} {sch} {'f16' if f16 else 'f32'}: one call {t1:.3f} ms", flush=True)
    outs1 = {}
    for nb, frac in ((2, None), (2, 0.4), (2, 0.3), (2, 0.6), (3, None), (4, None)):
        rows = [((H // 8) * i // nb) * 8 for i in range(nb + 1)]
        if frac is not None:
            rows[1] = int(H * frac) // 16 * 16
        rows[-1] = H
        plans = [rm.RenderPlan(model, sch, td) for _ in range(nb)]
        streams = [torch.cuda.Stream() for _ in range(nb)]
        outs = [dict() for _ in range(nb)]
        ros = [ro[rows[i] * W:rows[i + 1] * W] for i in range(nb)]
        rds = [rd[rows[i] * W:rows[i + 1] * W] for i in range(nb)]

        def seq():
            for i in range(nb):
                rm.render_rays(plans[i], ros[i], rds[i], tile_w=W, out=outs[i])

        def par():
            cur = torch.cuda.current_stream()
            for i in range(nb):
                streams[i].wait_stream(cur)
                with torch.cuda.stream(streams[i]):
                    rm.render_rays(plans[i], ros[i], rds[i], tile_w=W, out=outs[i])
            for i in range(nb):
                cur.wait_stream(streams[i])
        ts, tp = timeit(seq), timeit(par)
        # the same as captured HIP graphs (no host launch cost: what an in-library implementation could reach)
        def graphed(fn):
            side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn(); fn()
            torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g
        try:
            g1 = graphed(lambda: rm.render_rays(plan, ro, rd, tile_w=W, out=outs1))
            gp = graphed(par)
            tg1, tgp = timeit(g1.replay), timeit(gp.replay)
            print(f"      as HIP graphs: one call {tg1:.3f} ms, {nb} bands on {nb} streams {tgp:.3f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            print("      graph capture failed:", type(e).__name__, str(e)[:200], flush=True)
        par(); torch.cuda.synchronize()
        img = torch.cat([o["image"] for o in outs])
        print(f"   {nb} bands {rows}: one stream {ts:.3f} ms, {nb} streams {tp:.3f} ms, max|d image| {float((img - base).abs().max()):.1e}", flush=True)
