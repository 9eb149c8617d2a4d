#!/usr/bin/env python3
"""Headline benchmark: rays/s of the full forward render (BASELINE.json `metric`).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): 800x800 rays per GPU, hash grid L=16 T=2^19 F=2, 2x64
sigma/colour MLP, synthetic camera + random-init (hash-generated) weights, rays resident in HBM.
  --schedule flat128 (default): num_steps=[128]  = "128 samples/ray through the L=16 grid", configs[1] literally
  --schedule ref              : num_steps=[128,64,32] with both proposal grids = the reference's own default
One "step" = one whole-image render: sn_rm_render_rays over this rank's row band (+ at N>1 the
RCCL all-gather that assembles the image on every rank; the gather of frame k overlaps the render of
frame k+1, all of them complete inside the timed region).  Weak scaling: the image grows to
800 x (800*N) rows, each rank renders an 800-row band.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     — dominant kernel (final stage) timed with HIP events inside the timed region,
                 algorithmic gather bytes per ray / measured time vs 8 TB/s
  cpu_baseline — the CPU oracle (a port, oracle/) on a bounded sample of the same rays, N=1 only
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip table
MFMA_F32_PEAK = 157.3e12   # FLOP/s dense f32-input MFMA


def algorithmic_bytes_per_ray(steps, s_bytes):
    """sum_stages T_k * L_k * 2^D * F_k * s + 24 (o,d in) + 20 (rgb, depth, wsum out)."""
    L = [5] * (len(steps) - 1) + [16]
    return sum(t * l * 8 * 2 * s_bytes for t, l in zip(steps, L)) + 44


def flops_per_ray(steps):
    macs = sum(t * 176 for t in steps[:-1]) + steps[-1] * 7168 + 2112
    return 2 * macs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--schedule", choices=["flat128", "ref"], default="flat128")
    ap.add_argument("--hw", type=int, default=800)
    ap.add_argument("--tables", choices=["f32", "f16"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true", help="world size 1 only: still go through RCCL (init, all-gather pipeline, barriers)")
    ap.add_argument("--primary-only", action="store_true", help="skip the extra configurations reported under `also`")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = args.force_dist and world == 1      # exercise the RCCL code path on a one-GPU box
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dist:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), rank=rank, world_size=world)
    multi = world > 1 or force_dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from helpers import product_model, synthetic_params
    from sanerf_hq_amd import _lib, raymarching as rm, synth
    from sanerf_hq_amd.dist import PipelinedGather, shard_rows

    W = args.hw
    H = args.hw * world                      # weak scaling: one hw x hw band per rank
    b, e = shard_rows(H, world, rank)
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    intr = synth.pinhole_intrinsics(args.hw, W)   # same focal length at every N
    rays_o, rays_d = rm.generate_rays(pose, (intr[0], intr[1], W / 2.0, H / 2.0), H, W, device=dev, row_begin=b, row_end=e)
    n_local = rays_o.shape[0]
    lib = _lib.lib()
    total_rays = H * W
    models = {}

    def measure(schedule, tables, n_steps, n_warm, rays=None):
        """K timed whole-image renders (+ all-gather at N>1) of one configuration; returns the bench numbers.
        rays = (rays_o, rays_d, width): another image than the bench line's (N = 1 only)."""
        r_o, r_d, r_w = rays if rays is not None else (rays_o, rays_d, W)
        n_total = r_o.shape[0] if rays is not None else total_rays
        steps = [128] if schedule == "flat128" else [128, 64, 32]
        if schedule not in models:
            params = synthetic_params(steps, seed=0)
            models[schedule] = (params, product_model(params, steps, False, dev))
        params, model = models[schedule]
        plan = rm.RenderPlan(model, steps, torch.float16 if tables == "f16" else torch.float32)
        out = {}
        # N > 1: the all-gather of frame k (RCCL, its own stream, over xGMI) overlaps the render of frame k+1; two
        # rotating image buffers, everything in flight is drained inside the timed region
        pipe = PipelinedGather(H, W, 5, dev, depth=2) if multi else None

        def step():
            rm.render_rays(plan, r_o, r_d, tile_w=r_w, out=out)
            if multi:
                band = torch.cat([out["image"], out["depth"].unsqueeze(-1), out["weights_sum"].unsqueeze(-1)], dim=-1)
                pipe.submit(band)

        for _ in range(n_warm):
            step()
        if pipe is not None:
            pipe.drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        lib.sn_rm_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        if pipe is not None:
            pipe.drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        ms = (C.c_float * 8)()
        cnt = (C.c_int32 * 8)()
        _lib.check(lib.sn_rm_profile_read(ms, cnt, 8), "profile_read")
        lib.sn_rm_profile_enable(0)
        if multi:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        s_bytes = 2 if tables == "f16" else 4
        per = lambda i: (ms[i] / cnt[i]) if cnt[i] else None
        return dict(steps=steps, params=params, out=out, elapsed=elapsed, value=n_total / (elapsed / n_steps),
                    ms_per_step=elapsed / n_steps * 1e3, s_bytes=s_bytes, final_ms=per(4), final_launches=int(cnt[4]),
                    pack_ms=per(0), prop_ms=[per(1), per(2)])

    m = measure(args.schedule, args.tables, args.steps, args.warmup)
    steps, params, out = m["steps"], m["params"], m["out"]
    value, ms_per_step, s_bytes, final_ms = m["value"], m["ms_per_step"], m["s_bytes"], m["final_ms"]

    # ---- roofline of the dominant kernel (final stage) on this rank ----
    bytes_final = n_local * (steps[-1] * 16 * 8 * 2 * s_bytes + 44)
    flops_final = n_local * 2 * (steps[-1] * 7168 + 2112)
    achieved = bytes_final / (final_ms * 1e-3)
    # HBM-side bytes per launch of the same kernel: from the committed rocprofv3 --pmc passes of this same command
    # (profiles/latest_traffic.json, regenerated with tools/gpu_pmc_quick.sh); null when no profile matches.
    traffic = traffic_detail = None
    tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath)).get(f"{args.schedule}_{args.tables}")
        if tj and tj.get("rays") == n_local:
            traffic, traffic_detail = tj["hbm_bytes_per_launch"], tj
    roofline = {
        "kernel": "k_final_stage", "bound": "hbm",
        "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic, "traffic_detail": traffic_detail,
        "avg_kernel_ms": round(final_ms, 4), "launches": m["final_launches"],
        "algorithmic_bytes_per_launch": int(bytes_final),
        "note": "achieved = algorithmic gather bytes (SURVEY 8d: every corner fetch counted once, no cache credit) / HIP-event kernel time; "
                "it can exceed the HBM peak because L1/L2 absorb the re-reads of neighbouring rays (see traffic)",
        "mlp_on_matrix_cores": {"achieved_tflops": round(flops_final / (final_ms * 1e-3) / 1e12, 2),
                                "note": "algorithmic fp32-equivalent MLP FLOPs; executed as 3 fp16 MFMA products per fp32 product"},
        "gather_address_rate": {   # the ceiling that binds after the caches (tools/ubench/gathers.hip)
            "wave_instructions_per_launch": int(-(-n_local // 64) * steps[-1] * (5 * 4 + 11 * 8)),
            "cycles_per_instruction_per_cu": 17.5,
            "floor_ms": round(-(-n_local // 64) * steps[-1] * (5 * 4 + 11 * 8) * 17.5 / 256 / 2.4e9 * 1e3, 3),
            "note": "one wave-wide gather per 17.5 cycles per CU (4 lanes/clk address rate, measured); 5 dense levels x 4 paired loads + 11 hashed levels x 8; "
                    "priced at the 2.4 GHz peak clock -- under this kernel the shader clock is ~1.97 GHz (GRBM_GUI_ACTIVE, profiles/r01/pmc_flat128.txt), i.e. floor x 1.22"},
        "other_kernels_ms": {"pack": round(m["pack_ms"], 4) if m["pack_ms"] else None,
                             "prop0": round(m["prop_ms"][0], 4) if m["prop_ms"][0] else None,
                             "prop1": round(m["prop_ms"][1], 4) if m["prop_ms"][1] else None},
        "whole_path": {"algorithmic_bytes_per_ray": algorithmic_bytes_per_ray(steps, s_bytes),
                       "achieved_GBps": round(value / world * algorithmic_bytes_per_ray(steps, s_bytes) / 1e9, 2),
                       "frac": round(value / world * algorithmic_bytes_per_ray(steps, s_bytes) / HBM_PEAK, 4),
                       "flops_per_ray": flops_per_ray(steps)},
    }

    # ---- other configurations of the same path, measured in the same process (N = 1 only, short) ----
    also = None
    if world == 1 and not args.primary_only:
        also = {}
        for sch, tb in (("ref", "f32"), ("flat128", "f16"), ("ref", "f16")):
            if (sch, tb) == (args.schedule, args.tables):
                continue
            r = measure(sch, tb, max(3, args.steps // 2), 2)
            also[f"{sch}_{tb}"] = {"rays_per_s": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 4),
                                   "num_steps": r["steps"], "tables": tb,
                                   "kernel_ms": {"final": round(r["final_ms"], 4),
                                                 "prop0": round(r["prop_ms"][0], 4) if r["prop_ms"][0] else None,
                                                 "prop1": round(r["prop_ms"][1], 4) if r["prop_ms"][1] else None}}
        # BASELINE configs C4 at one GPU: 1600 x 1600 rays of the same view (same field of view: rays twice as dense, so
        # neighbouring lanes share more table lines -- higher rays/s than the 800 x 800 bench line)
        H4 = 1600
        ro4, rd4 = rm.generate_rays(pose, synth.pinhole_intrinsics(H4, H4), H4, H4, device=dev)
        for sch in ("flat128", "ref"):
            r = measure(sch, "f32", 3, 1, rays=(ro4, rd4, H4))
            also[f"c4_1600x1600_{sch}_f32"] = {"rays_per_s": round(r["value"], 1), "ms_per_step": round(r["ms_per_step"], 4),
                                               "num_steps": r["steps"], "tables": "f32", "rays": H4 * H4}
        del ro4, rd4
        out = m["out"]

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        from helpers import oracle_cfg
        cfg = oracle_cfg(orc, params, steps, table_f16=(args.tables == "f16"))
        ro_h = rays_o.cpu().numpy()
        rd_h = rays_d.cpu().numpy()
        pick = (synth.hash_u01(1024, 77) * n_local).astype(np.int64)       # calibration sample
        t0 = time.perf_counter(); orc.render(cfg, ro_h[pick], rd_h[pick]); t_cal = time.perf_counter() - t0
        n_cpu = int(min(n_local, max(2048, args.cpu_seconds * 1024 / max(t_cal, 1e-6))))
        pick = (synth.hash_u01(n_cpu, 78) * n_local).astype(np.int64)
        t0 = time.perf_counter(); ref = orc.render(cfg, ro_h[pick], rd_h[pick]); t_cpu = time.perf_counter() - t0
        err = float(np.abs(out["image"][torch.from_numpy(pick).to(dev)].cpu().numpy() - ref["image"]).max())
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
        except OSError:
            pass
        cpu_baseline = {"value": round(n_cpu / t_cpu, 1), "unit": "rays/s", "cores": orc.num_threads(), "kind": "port",
                        "cpu_model": cpu_model, "nproc": os.cpu_count(),
                        "sample": f"{n_cpu} pseudo-random rays of the same {W}x{H} image, same weights, oracle/liboracle.so (C11+OpenMP), {t_cpu:.1f} s",
                        "max_abs_rgb_diff_vs_gpu": err}

    if rank == 0:
        line = {
            "metric": "rays/s full render (800x800, hashgrid L=16 + 2x64 MLP)", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 tables, positions, interpolation, compositing; the 32-64-64-16 MLP multiplies fp16 hi/lo splits of fp32 operands on the matrix cores with fp32 accumulation (2^-22 per product, RGB within 1e-5 of the fp32 oracle)",
            "config": {"workload": f"BASELINE configs[1]: {W}x{args.hw} rays per GPU ({W}x{H} image), hashgrid L=16 T=2^19 F=2, "
                                   f"32-64-64-16 + 31-32-32-3 MLPs, num_steps={steps} ({args.schedule}), tables {args.tables}, "
                                   "arithmetic fp32, random-init weights, orbit camera",
                       "rays_per_gpu": n_local, "image": [H, W], "schedule": args.schedule,
                       "parallelism": f"ray-tile row bands x{world}" + (" + RCCL all-gather of rgb|depth|wsum" if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "also": also,
        }
        print(json.dumps(line))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
