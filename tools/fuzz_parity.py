"""Randomised parity sweep of the fused renderer against the CPU oracle (not part of the test suite: run on the GPU box,
python tools/fuzz_parity.py [cases] [seed]).  Per case: random schedule (1-3 stages, odd step counts), image size, table
precision, camera, optional per-ray near/far clamps; tiled and linear ray order (the latter takes the several-lanes-per-ray
kernels for small batches) and the opt-in compacting final stage must agree bit for bit with each other; sample indices must equal the oracle's, image / depth /
weights_sum stay within the fp32 contract."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc  # noqa: E402
from helpers import camera_rays, oracle_cfg, product_model, synthetic_params  # noqa: E402
from sanerf_hq_amd import raymarching as rm  # noqa: E402


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def main_any():
    """python tools/fuzz_parity.py any [cases] [seed]: random fields of OTHER sizes than the reference network's (levels, table size, MLP depths and
    widths, geometry channels) through the size-agnostic last stage (k_final_stage_any) behind the fused proposal stages, against the oracle."""
    from sanerf_hq_amd.encoding import get_encoder
    from sanerf_hq_amd.nerf.network import MLP
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        S = int(rng.integers(1, 4))
        steps = [int(rng.integers(3, 70)) for _ in range(S)]
        H, W = int(rng.integers(5, 40)), int(rng.integers(5, 40))
        f16 = bool(rng.integers(0, 2))
        L, log2T = int(rng.integers(1, 17)), int(rng.integers(10, 20))
        geo = int(rng.integers(1, 32))
        ng, nv = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        hid, vhid = int(rng.integers(4, 65)), int(rng.integers(4, 65))
        params = synthetic_params(steps, seed=3000 + c)
        model = product_model(params, steps, False, dev)
        torch.manual_seed(seed * 1000 + c)
        model.grid, d = get_encoder("hashgrid" if rng.integers(0, 4) else "tiledgrid", input_dim=3, level_dim=2, num_levels=L, log2_hashmap_size=log2T,
                                    desired_resolution=int(rng.integers(64, 4096)))
        model.grid_mlp = MLP(d, 1 + geo, hid, ng, bias=False)
        model.view_mlp = MLP(geo + 16, 3, vhid, nv, bias=False)
        model.geom_feat_dim = geo
        model = model.to(dev).eval()
        with torch.no_grad():
            model.grid.embeddings.uniform_(-1.0, 1.0)
            for lin in list(model.grid_mlp.net) + list(model.view_mlp.net):
                lin.weight.mul_(float(rng.uniform(1.0, 4.0)))
        assert model._fused_kind() == "any", model._fused_kind()
        _, _, ro, rd = camera_rays(orc, H, W, radius=float(rng.uniform(0.4, 2.5)), elev=float(rng.uniform(-60, 70)), azim=float(rng.uniform(0, 360)))
        plan = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32)
        tiled = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, dev), T(rd, dev), tile_w=W, want=("inds",)).items()}
        linear = rm.render_rays(plan, T(ro, dev), T(rd, dev), tile_w=0, out={}, want=("inds",))
        cfg = oracle_cfg(orc, params, steps, table_f16=f16)
        emb = model.grid.embeddings.detach().cpu().numpy()
        cfg.grid = orc.make_grid(emb.astype(np.float16) if f16 else emb, model.grid.offsets.cpu().numpy(), model.grid.per_level_scale, model.grid.base_resolution,
                                 gridtype=model.grid.gridtype_id, keep=cfg._keep)
        cfg.grid_mlp = orc.make_mlp([l.weight.detach().cpu().numpy() for l in model.grid_mlp.net], keep=cfg._keep)
        cfg.view_mlp = orc.make_mlp([l.weight.detach().cpu().numpy() for l in model.view_mlp.net], keep=cfg._keep)
        want = orc.render(cfg, ro, rd, debug=True)
        same_order = all(torch.equal(tiled[k], linear[k]) for k in tiled)
        inds_ok = all(np.array_equal(tiled[f"inds{k}"].cpu().numpy(), want[f"inds{k}"]) for k in range(1, S))
        e_img = float(np.abs(tiled["image"].cpu().numpy() - want["image"]).max())
        e_ws = float(np.abs(tiled["weights_sum"].cpu().numpy() - want["weights_sum"]).max())
        ok = same_order and inds_ok and e_img <= 1e-5 and e_ws <= 2e-6
        print(f"case {c}: steps={steps} {H}x{W} f16={f16} L={L} T=2^{log2T} grid_mlp {2 * L}-{hid}x{ng - 1}-{1 + geo} view_mlp {geo + 16}-{vhid}x{nv - 1}-3  "
              f"dRGB={e_img:.1e} dwsum={e_ws:.1e} tiled==linear:{same_order} inds:{inds_ok}  {'ok' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
    print("mismatching cases:", bad)
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "any":
        return main_any()
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        S = int(rng.integers(1, 4))
        steps = [int(rng.integers(3, 140)) for _ in range(S)]
        H, W = int(rng.integers(5, 70)), int(rng.integers(5, 70))
        f16 = bool(rng.integers(0, 2))
        params = synthetic_params(steps, seed=1000 + c, gain=float(rng.uniform(1.0, 8.0)))
        model = product_model(params, steps, False, dev)
        _, _, ro, rd = camera_rays(orc, H, W, radius=float(rng.uniform(0.4, 2.5)), elev=float(rng.uniform(-60, 70)), azim=float(rng.uniform(0, 360)))
        cnf = None
        if rng.integers(0, 2):
            cnf = np.stack([rng.uniform(0.2, 0.7, H * W), rng.uniform(1.5, 40.0, H * W)], -1).astype(np.float32)
        plan = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32)
        kw = dict(cam_near_far=None if cnf is None else T(cnf, dev), want=("inds",))
        # the default kernel's linear-tail form (third layer's geometry rows once per ray) first: oracle tolerances apply to it as well
        rm.tuning.per_sample_form = 0
        lt = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, dev), T(rd, dev), tile_w=W, out={}, **kw).items()}
        # bit identity between the kernels (tile / linear lane mapping, several lanes per ray, compaction) is a property of the per-sample form
        rm.tuning.per_sample_form = 1
        tiled = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, dev), T(rd, dev), tile_w=W, **kw).items()}
        linear = rm.render_rays(plan, T(ro, dev), T(rd, dev), tile_w=0, out={}, **kw)
        want = orc.render(oracle_cfg(orc, params, steps, table_f16=f16), ro, rd, cam_near_far=cnf, debug=True)
        # the opt-in compacting final stage must reproduce the default kernels bit for bit while nothing is skipped (contracted
        # scene: no ray misses the aabb), in both ray orders
        planc = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32, compact_live=True)
        kwc = dict(cam_near_far=kw["cam_near_far"])
        cmp_t = rm.render_rays(planc, T(ro, dev), T(rd, dev), tile_w=W, out={}, **kwc)
        cmp_l = rm.render_rays(planc, T(ro, dev), T(rd, dev), tile_w=0, out={}, **kwc)
        compact_ok = all(torch.equal(tiled[k], cmp_t[k]) and torch.equal(tiled[k], cmp_l[k]) for k in ("image", "depth", "weights_sum"))
        same_order = all(torch.equal(tiled[k], linear[k]) for k in tiled) and compact_ok
        inds_ok = all(np.array_equal(tiled[f"inds{k}"].cpu().numpy(), want[f"inds{k}"]) for k in range(1, S))
        ok = same_order and inds_ok
        e_img = float(np.abs(tiled["image"].cpu().numpy() - want["image"]).max())
        e_ws = float(np.abs(tiled["weights_sum"].cpu().numpy() - want["weights_sum"]).max())
        e_dep = float((np.abs(tiled["depth"].cpu().numpy() - want["depth"]) / (1e-5 + 1e-5 * np.abs(want["depth"]))).max())
        # north_star: RGB within 1e-4 (the split-fp16 MLP's error grows with the MLP gain drawn above; the test suite's scenes stay < 1e-5)
        e_lt = float(np.abs(lt["image"].cpu().numpy() - want["image"]).max())
        d_lt = float((lt["image"] - tiled["image"]).abs().max())
        ok = ok and e_img <= 1e-4 and e_ws <= 2e-6 and e_dep <= 1.0 and e_lt <= 1e-4 and d_lt <= 5e-5
        print(f"case {c}: steps={steps} {H}x{W} f16={f16} cnf={cnf is not None}  dRGB={e_img:.1e} dwsum={e_ws:.1e} ddepth(rel 1e-5 units)={e_dep:.2f} "
              f"tiled==linear==compact:{same_order} inds:{inds_ok} linear-tail: dRGB={e_lt:.1e} vs per-sample {d_lt:.1e}  {'ok' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
    print("mismatching cases:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
