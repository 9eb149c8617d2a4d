"""GPU tests of the fused render's stages beyond tests/test_gpu_render.py: full-size checks of both table precisions and schedules, the linear-tail last stage against the per-sample form and the fixtures, densified levels, wave-tile workgroups of any band shape, row bands on two HIP streams, exact early-outs, the bench route against the reference's fp16-table fixtures, the size-agnostic stage, output strides and argument validation, the experiments-build variants."""
import ctypes as C  # noqa: F401
import os
import subprocess  # noqa: F401
import sys

import numpy as np
import pytest
import torch

from helpers import camera_rays, make_opt, oracle_cfg, product_model, synthetic_params  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("steps,f16", [([128], False), ([128], True), ([128, 64, 32], True)])
def test_full_size_both_table_precisions_and_schedules(gpu, orc, steps, f16):
    """800x800 (BASELINE configs[1]: 128 samples per ray, fp16 tables) at full size: determinism, partition of unity, linear
    lane mapping == tiled, and 512 pseudo-random pixels against the oracle on the same (fp16-rounded) tables, RGB <= 1e-5."""
    from sanerf_hq_amd import raymarching as rm, synth
    params = synthetic_params(steps, seed=19)
    model = product_model(params, steps, False, gpu)
    H = W = 800
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    plan = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32)
    a = rm.render_rays(plan, ro, rd, tile_w=W)
    img = a["image"].clone(); dep = a["depth"].clone(); ws = a["weights_sum"].clone()
    assert torch.isfinite(img).all() and torch.isfinite(dep).all()
    np.testing.assert_allclose(ws.cpu().numpy(), 1.0, atol=3e-6)
    b = rm.render_rays(plan, ro, rd, tile_w=W)
    assert torch.equal(b["image"], img) and torch.equal(b["depth"], dep)
    c = rm.render_rays(plan, ro, rd, tile_w=0, out={})
    assert torch.equal(c["image"], img)
    idx = (synth.hash_u01(512, 11) * (H * W)).astype(np.int64)
    want = orc.render(oracle_cfg(orc, params, steps, table_f16=f16), ro[idx].cpu().numpy(), rd[idx].cpu().numpy())
    np.testing.assert_allclose(img[idx].cpu().numpy(), want["image"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(dep[idx].cpu().numpy(), want["depth"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("steps", [[128], [128, 64, 32], [7], [33, 17, 9]])
def test_role_split_final_stage_is_bit_identical(gpu, orc, steps, monkeypatch, per_sample_form, experiments_build):
    """k_final_stage_rs (experiments builds, tuning.experiment = EXP_ROLE_SPLIT: producer waves gather and blend, consumer waves run the matrix-core MLP and composite,
    hand-over through LDS rings) performs the arithmetic of k_final_stage in its order: every output must be equal bit for bit,
    for ragged image shapes, tiled and linear lane mapping and both table precisions."""
    from sanerf_hq_amd import raymarching as rm, synth
    params = synthetic_params(steps, seed=23)
    model = product_model(params, steps, False, gpu)
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    from sanerf_hq_amd import _lib
    rs = rm.Tuning(per_sample_form=1, experiment=_lib.EXP_ROLE_SPLIT)
    for tdt in (torch.float32, torch.float16):
        plan = rm.RenderPlan(model, steps, tdt)
        for (H, W) in ((64, 64), (48, 80), (200, 104), (16, 32), (40, 24)):
            intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
            ro, rd = rm.generate_rays(pose, intr, H, W, device=gpu)
            for tile in (W, 0):
                a = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=tile, want=("f_image",)).items()}
                b = rm.render_rays(plan, ro, rd, tile_w=tile, want=("f_image",), tuning=rs)
                for k in ("image", "depth", "weights_sum", "f_image"):
                    assert torch.equal(a[k], b[k]), (steps, tdt, H, W, tile, k, float((a[k] - b[k]).abs().max()))
    # against the oracle directly as well (one shape)
    _, _, ro, rd = camera_rays(orc, 32, 32)
    plan = rm.RenderPlan(model, steps, tuning=rs)
    got = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=32)
    assert rm.last_launch_info()["final_kernel"] == "k_final_stage_rs"
    want = orc.render(oracle_cfg(orc, params, steps), ro, rd)
    np.testing.assert_allclose(got["image"].cpu().numpy(), want["image"], rtol=0, atol=1e-5)


def test_range_guard_follows_the_packages_own_adam_and_skips_proposal_only_calls(gpu, orc):
    """sn_adam_step writes parameters through raw pointers; the optimiser now bumps their version counters, so
    the fp16 range guard of a cached plan sees weights that grew past the split-fp16 bound.  And a skip_final call (proposal
    stages only: fp32 vector arithmetic) must not evaluate the guard at all (no host synchronisation in such a step)."""
    import warnings
    from sanerf_hq_amd import optim, raymarching as rm
    steps = [32, 16]
    model = product_model(synthetic_params(steps, seed=21), steps, False, gpu)
    _, _, ro, rd = camera_rays(orc, 16, 16)
    ro, rd = T(ro, gpu), T(rd, gpu)
    with torch.no_grad():
        model.render(ro, rd)
    assert model._plan.cfg.mlp_exact_fp32 == 0
    w = model.grid_mlp.net[0].weight
    w.requires_grad_(True)
    opt = optim.Adam([w], lr=3.0e4, eps=1e-15)                     # one step moves every weight by ~lr
    v0 = w._version
    w.grad = torch.ones_like(w)
    opt.step()
    assert w._version > v0, "sn_adam_step must bump the version counter of the tensor it rewrote"
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = model.render(ro, rd)
    assert model._plan.cfg.mlp_exact_fp32 == 1 and torch.isfinite(out["image"]).all()
    # proposal-only call: the guard is not consulted
    plan = model._plan
    called = []
    orig = plan.check_range
    plan.check_range = lambda: called.append(1) or orig()
    rm.render_rays(plan, ro, rd, skip_final=True)
    assert not called
    rm.render_rays(plan, ro, rd)
    assert called


def test_render_rays_validates_per_ray_table_strides(gpu, orc):
    """a direct C caller passing a per-ray stride shorter than T+1 must get SN_ERR_INVALID, not out-of-bounds reads."""
    from sanerf_hq_amd import _lib, raymarching as rm
    steps = [32, 16]
    model = product_model(synthetic_params(steps, seed=5), steps, False, gpu)
    plan = rm.RenderPlan(model, steps)
    N = 64
    ro = torch.zeros(N, 3, device=gpu); rd = torch.ones(N, 3, device=gpu)
    img = torch.empty(N, 3, device=gpu); dep = torch.empty(N, device=gpu); ws = torch.empty(N, device=gpu)
    work = plan.workspace(N, 0, gpu)
    tab = torch.zeros(N, 40, device=gpu)

    def call(**fields):
        io = _lib.RenderIO()
        io.rays_o, io.rays_d, io.N = ro.data_ptr(), rd.data_ptr(), N
        io.image, io.depth, io.weights_sum = img.data_ptr(), dep.data_ptr(), ws.data_ptr()
        io.workspace, io.workspace_bytes = work.data_ptr(), work.numel()
        for k, v in fields.items():
            if k == "u1":
                io.u_table[1] = v
            elif k == "u1_stride":
                io.u_ray_stride[1] = v
            else:
                setattr(io, k, v)
        return _lib.lib().sn_rm_render_rays(C.byref(plan.cfg), C.byref(io), _lib.stream())

    assert call() == 0
    assert call(bins0_table=tab.data_ptr(), bins0_ray_stride=40) == 0
    for bad in (dict(bins0_table=tab.data_ptr(), bins0_ray_stride=32), dict(bins0_ray_stride=40),
                dict(u1=tab.data_ptr(), u1_stride=16), dict(u1_stride=40)):
        rc = call(**bad)
        assert rc != 0 and b"stride" in _lib.lib().sn_last_error(), (bad, rc)
    torch.cuda.synchronize()


@pytest.mark.parametrize("steps", [[128], [128, 64, 32], [7]])
def test_linear_tail_form_vs_per_sample_form_and_oracle(gpu, orc, steps, monkeypatch):
    """The default final stage (no per-sample outputs) takes the third layer off the matrix cores: density row per sample as an
    fp32 dot product, geometry rows once per ray on sum_j w_j relu(h2_j).  A re-association (like SH(d) * sum_j w_j): it must stay
    in the fp32 round-off class -- against the per-sample form (Tuning.per_sample_form) and against the oracle -- for ragged shapes, both
    lane mappings and both table precisions, with and without the feature stage."""
    from sanerf_hq_amd import raymarching as rm, synth
    params = synthetic_params(steps, heads=True, seed=29)
    model = product_model(params, steps, True, gpu)
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    for tdt in (torch.float32, torch.float16):
        for feat in (None, model.s_grid):
            plan = rm.RenderPlan(model, steps, tdt, feat_encoder=feat)
            for (H, W) in ((64, 64), (40, 24)):
                intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
                ro, rd = rm.generate_rays(pose, intr, H, W, device=gpu)
                a = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, want=("f_image",), out={}, tuning=rm.Tuning(per_sample_form=1)).items()}
                assert "per-sample" in rm.last_launch_info()["final_kernel"]
                b = rm.render_rays(plan, ro, rd, tile_w=W, want=("f_image",), out={})
                assert "<lt" in rm.last_launch_info()["final_kernel"]
                assert float((a["image"] - b["image"]).abs().max()) <= 4e-6
                assert float((a["weights_sum"] - b["weights_sum"]).abs().max()) <= 1e-6
                np.testing.assert_allclose(b["depth"].cpu().numpy(), a["depth"].cpu().numpy(), rtol=2e-6, atol=2e-6)
                fmax = float(a["f_image"].abs().max())
                assert float((a["f_image"] - b["f_image"]).abs().max()) <= 2e-6 * max(fmax, 1.0)
                if feat is not None:
                    assert float((a["f_feat"] - b["f_feat"]).abs().max()) <= 2e-6 * max(float(a["f_feat"].abs().max()), 1.0)
                else:
                    want = orc.render(oracle_cfg(orc, params, steps, table_f16=(tdt == torch.float16)), ro.cpu().numpy(), rd.cpu().numpy())
                    np.testing.assert_allclose(b["image"].cpu().numpy(), want["image"], rtol=0, atol=1e-5)
                    np.testing.assert_allclose(b["depth"].cpu().numpy(), want["depth"], rtol=1e-5, atol=1e-5)


def test_linear_tail_form_vs_reference_fixtures(gpu, orc, monkeypatch):
    """The reference's own outputs (tests/golden/render_sref.npz, render_flat128.npz) through the default call without per-sample
    tensors, i.e. the linear-tail kernel (the several-lanes-per-ray kernels that small batches normally take are switched off):
    RGB within the north-star tolerance 1e-4, depth / weights_sum within 1e-4."""
    from helpers import golden, params_from_spec, spec_of
    from sanerf_hq_amd import raymarching as rm
    monkeypatch.setattr(rm.tuning, "final_sp_max_rays", -1)
    monkeypatch.setattr(rm.tuning, "prop_sp_max_rays", -1)
    for name, steps in (("render_sref", [128, 64, 32]), ("render_flat128", [128])):
        g = golden(name)
        model = product_model(params_from_spec(spec_of(g)), steps, False, gpu)
        u_tables = {k: T(g[f"u{k}"], gpu) for k in range(1, len(steps))} if len(steps) > 1 else None
        plan = rm.RenderPlan(model, steps)
        out = rm.render_rays(plan, T(g["rays_o"], gpu), T(g["rays_d"], gpu), u_tables=u_tables)
        np.testing.assert_allclose(out["image"].cpu().numpy(), g["image"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(out["depth"].cpu().numpy(), g["depth"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), g["weights_sum"], rtol=0, atol=1e-4)


def test_lds_resident_level0_is_bit_identical(gpu, orc, monkeypatch, experiments_build):
    """tuning.experiment = EXP_LDS_LEVEL0 (experiments builds; north_star "LDS staging of per-tile grid voxels"): with fp16 tables the coarsest level of the main
    grid (16^3 vertices, 16 KiB) is staged in LDS by every workgroup and read with ds_read_b32 instead of gathers; same arithmetic
    as the texture-path form, so every output is equal bit for bit (feature slabs move to their unpadded XOR-swizzled layout)."""
    from sanerf_hq_amd import raymarching as rm, synth
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    for steps in ([128], [128, 64, 32], [7]):
        params = synthetic_params(steps, heads=True, seed=31)
        model = product_model(params, steps, True, gpu)
        for feat in (None, model.s_grid):
            plan = rm.RenderPlan(model, steps, torch.float16, feat_encoder=feat)
            for (H, W) in ((64, 64), (48, 80), (40, 24)):
                intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
                ro, rd = rm.generate_rays(pose, intr, H, W, device=gpu)
                a = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, want=("f_image",), out={}).items()}
                b = rm.render_rays(plan, ro, rd, tile_w=W, want=("f_image",), out={}, tuning=rm.Tuning(experiment=2))
                assert "lds-level0" in rm.last_launch_info()["final_kernel"]
                for k in a:
                    assert torch.equal(a[k], b[k]), (steps, feat is not None, H, W, k)


def test_densified_levels_are_bit_identical(gpu, orc, monkeypatch):
    """Tuning.densify (automatic for large fp16-table renders): the first two hashed levels of the main grid (102^3 and 148^3
    vertices) are re-laid out per call as 16-byte pair / quad rows -- fetched through the hash once per vertex by the pack kernel --
    and the final stage reads them like dense levels (4 / 2 coherent gathers instead of 8 scattered ones).  Same values, same
    arithmetic: every output equals the hashed-lookup form bit for bit, both table precisions, tiled and linear lane mapping."""
    from sanerf_hq_amd import raymarching as rm, synth
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    monkeypatch.setattr(rm.tuning, "final_sp_max_rays", -1)             # small linear batches would take the several-lanes-per-ray kernels
    monkeypatch.setattr(rm.tuning, "prop_sp_max_rays", -1)
    for steps in ([128], [128, 64, 32], [7]):
        params = synthetic_params(steps, seed=37)
        model = product_model(params, steps, False, gpu)
        for tdt in (torch.float32, torch.float16):
            plan = rm.RenderPlan(model, steps, tdt)
            for (H, W) in ((64, 64), (48, 80), (40, 24)):
                intr = synth.pinhole_intrinsics(H, W)[:2] + (W / 2.0, H / 2.0)
                ro, rd = rm.generate_rays(pose, intr, H, W, device=gpu)
                for tile in (W, 0):
                    monkeypatch.setattr(rm.tuning, "densify", 1)
                    a = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=tile, want=("f_image",), out={}).items()}
                    assert rm.last_launch_info()["dense_levels"] == 5
                    monkeypatch.setattr(rm.tuning, "densify", 2)
                    b = rm.render_rays(plan, ro, rd, tile_w=tile, want=("f_image",), out={})
                    assert rm.last_launch_info()["dense_levels"] == 7 and rm.last_launch_info()["gathers_per_wave_sample"] == (86 if tdt == torch.float16 else 100)
                    for k in a:
                        assert torch.equal(a[k], b[k]), (steps, tdt, H, W, tile, k)
    # and against the oracle, with the switch forced on
    _, _, ro, rd = camera_rays(orc, 32, 32)
    got = rm.render_rays(rm.RenderPlan(model, [7], torch.float16), T(ro, gpu), T(rd, gpu), tile_w=32)
    want = orc.render(oracle_cfg(orc, params, [7], table_f16=True), ro, rd)
    np.testing.assert_allclose(got["image"].cpu().numpy(), want["image"], rtol=0, atol=1e-5)


def test_random_field_sizes_through_the_size_agnostic_stage(gpu):
    """tools/fuzz_parity.py any: random fields of other sizes than the reference network's (levels, table size, hash / tiled, MLP depths and
    widths, geometry channels, schedules, fp16 / fp32 tables) against the oracle -- indices exact, RGB <= 1e-5, tile == linear order.  (A 120-case
    run of this sweep found what 20-case runs had not: an unrolled layer multiplying a never-written LDS row -- NaN -- by weight 0.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "any", "40", "13"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatching cases: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("H,W,steps,f16", [(24, 104, [128, 64, 32], True), (20, 100, [16], False), (8, 40, [48, 24], True), (40, 8, [33, 17, 9], False),
                                            (9, 17, [128], True), (56, 72, [128, 64, 32], False), (200, 120, [32], True)])
def test_wave_tile_workgroups_any_band_shape(gpu, orc, H, W, steps, f16, monkeypatch):
    """A workgroup of the fused stages is four CONSECUTIVE 8x8 wave tiles (pairs of wave-tile rows column-major, an odd last row left to
    right: render.hip ray_of_lane), so that a band whose height is 8 mod 16 launches ceil(wave tiles / 4) workgroups.  Shapes with an odd
    number of wave-tile rows / columns, partial wave tiles and workgroups that straddle two row pairs: image, depth, weights and sample
    indices equal the linear-order launch bit for bit and the oracle within the fp32 contract."""
    from sanerf_hq_amd import raymarching as rm
    params = synthetic_params(steps, seed=41)
    model = product_model(params, steps, False, gpu)
    _, _, ro, rd = camera_rays(orc, H, W, radius=1.1, elev=15.0, azim=75.0)
    plan = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32)
    tiled = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W, want=("inds", "weights")).items()}
    linear = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, want=("inds", "weights"), out={})
    for k in tiled:
        assert torch.equal(tiled[k], linear[k]), k
    monkeypatch.setattr(rm.tuning, "final_sp_max_rays", -1)       # (small linear-order batches would take the several-lanes-per-ray kernels: per-sample form)
    monkeypatch.setattr(rm.tuning, "prop_sp_max_rays", -1)
    plain_t = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W).items()}      # the default (linear-tail) kernel
    plain_l = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, out={})
    for k in plain_t:
        assert torch.equal(plain_t[k], plain_l[k]), k
    want = orc.render(oracle_cfg(orc, params, steps, table_f16=f16), ro, rd, debug=True)
    for k in range(1, len(steps)):
        assert np.array_equal(tiled[f"inds{k}"].cpu().numpy(), want[f"inds{k}"])
    np.testing.assert_allclose(plain_t["image"].cpu().numpy(), want["image"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(plain_t["depth"].cpu().numpy(), want["depth"], rtol=1e-5, atol=1e-5)


def test_wave_tile_workgroups_feature_stage_and_compaction(gpu, orc, monkeypatch):
    """The feature stage and the compacting final stage share the lane -> ray mapping with the stages in front of them (scratch columns):
    odd band shape, tile order == linear order."""
    from sanerf_hq_amd import raymarching as rm
    monkeypatch.setattr(rm.tuning, "final_sp_max_rays", -1)       # (small linear-order batches would take the several-lanes-per-ray kernels: per-sample form)
    monkeypatch.setattr(rm.tuning, "prop_sp_max_rays", -1)
    steps = [64, 32]
    params = synthetic_params(steps, heads=True, seed=5)
    model = product_model(params, steps, True, gpu)
    H, W = 24, 88
    _, _, ro, rd = camera_rays(orc, H, W)
    plan = rm.RenderPlan(model, steps, feat_encoder=model.s_grid)
    a = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W).items()}
    b = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, out={})
    assert float(a["f_feat"].abs().max()) > 0
    for k in a:
        assert torch.equal(a[k], b[k]), k
    plan_c = rm.RenderPlan(model, steps, compact_live=True)
    c = {k: v.clone() for k, v in rm.render_rays(plan_c, T(ro, gpu), T(rd, gpu), tile_w=W).items()}
    d = rm.render_rays(plan_c, T(ro, gpu), T(rd, gpu), tile_w=0, out={})
    for k in c:
        assert torch.equal(c[k], d[k]), k


@pytest.mark.parametrize("H,W,steps,f16,feat", [(72, 104, [128, 64, 32], True, False), (40, 64, [48, 24], False, False), (33, 40, [32, 16], True, False),
                                                 (48, 48, [128, 64, 32], False, True)])
def test_row_bands_on_two_streams_are_bit_identical(gpu, H, W, steps, f16, feat):
    """tuning.band_streams: a schedule with proposal stages rendered as two row bands whose kernels go to two HIP streams (forked from and joined
    to the caller's stream inside sn_rm_render_rays) -- every output, the per-stage tensors included, equals the single-stream render bit for
    bit (forced on for small images here; automatic from 2048 workgroups); also with the in-render feature stage, and from inside a captured
    HIP graph with allocator traffic between replays."""
    from sanerf_hq_amd import raymarching as rm
    params = synthetic_params(steps, heads=feat, seed=17)
    model = product_model(params, steps, feat, gpu)
    ro, rd = rm.generate_rays(__import__("sanerf_hq_amd").synth.orbit_pose(1.1, 15.0, 75.0), __import__("sanerf_hq_amd").synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    plan = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32, feat_encoder=model.s_grid if feat else None)
    want = ("inds", "weights", "bins") if not feat else ()
    one = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, want=want, tuning=rm.Tuning(band_streams=1)).items()}
    two = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, want=want, tuning=rm.Tuning(band_streams=2), out={}).items()}
    assert set(one) == set(two) and "image" in one
    for k in one:
        assert torch.equal(one[k], two[k]), k
    # inside a captured graph: the fork and the join are part of the capture (no per-stage tensors here: the default linear-tail kernel)
    plain = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, tuning=rm.Tuning(band_streams=1), out={}).items()}
    out = {}
    t2 = rm.Tuning(band_streams=2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        rm.render_rays(plan, ro, rd, tile_w=W, tuning=t2, out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rm.render_rays(plan, ro, rd, tile_w=W, tuning=t2, out=out)
    for _ in range(3):
        out["image"].fill_(float("nan"))
        g.replay()
        junk = torch.full((1 << 20,), float("nan"), device=gpu)
        del junk
    torch.cuda.synchronize()
    for k in plain:
        assert torch.equal(out[k], plain[k]), k


def test_row_bands_automatic_at_800x800_reference_schedule(gpu):
    """At 800x800 (2500 workgroups) a schedule with proposal stages takes the two-stream band split by itself: image, depth and weights
    equal the single-stream render bit for bit (fp16 tables, the bench's `also.ref_f16` configuration), and the single-stage schedule of
    the bench line is left alone (one launch of 2500 workgroups)."""
    from sanerf_hq_amd import raymarching as rm, synth
    H = W = 800
    steps = [128, 64, 32]
    model = product_model(synthetic_params(steps, seed=1), steps, False, gpu)
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    plan = rm.RenderPlan(model, steps, torch.float16)
    auto = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, tuning=rm.Tuning()).items()}
    info = rm.last_launch_info()
    assert info["launches"] == 2 and info["workgroups"] == 1250, info            # two bands of 400 rows
    one = rm.render_rays(plan, ro, rd, tile_w=W, tuning=rm.Tuning(band_streams=1), out={})
    assert rm.last_launch_info()["launches"] == 1
    for k in auto:
        assert torch.equal(auto[k], one[k]), k
    flat = rm.RenderPlan(product_model(synthetic_params([128], seed=1), [128], False, gpu), [128], torch.float16)
    rm.render_rays(flat, ro, rd, tile_w=W, tuning=rm.Tuning())
    assert rm.last_launch_info()["launches"] == 1 and rm.last_launch_info()["workgroups"] == 2500


@pytest.mark.parametrize("name,steps", [("render_flat128_h", [128]), ("render_sref_h", [128, 64, 32])])
def test_bench_route_vs_reference_fixture_with_fp16_tables(gpu, name, steps, monkeypatch):
    """The route the bench line takes -- fp16 table STORAGE, the linear-tail last stage, densified levels 5-6 (forced: the automatic rule wants
    64 M samples) -- against the reference's own outputs on tables of fp16 values (tests/golden/render_*_h.npz, generated by importing the
    reference's Python): RGB within the north-star tolerance 1e-4, depth / weights_sum within 1e-4; and the same without densified levels."""
    from helpers import golden, params_from_spec, spec_of
    from sanerf_hq_amd import raymarching as rm
    monkeypatch.setattr(rm.tuning, "final_sp_max_rays", -1)
    monkeypatch.setattr(rm.tuning, "prop_sp_max_rays", -1)
    g = golden(name)
    model = product_model(params_from_spec(spec_of(g), tables_f16=True), steps, False, gpu)
    u_tables = {k: T(g[f"u{k}"], gpu) for k in range(1, len(steps))} if len(steps) > 1 else None
    plan = rm.RenderPlan(model, steps, torch.float16)
    for densify in (2, 1):
        monkeypatch.setattr(rm.tuning, "densify", densify)
        out = rm.render_rays(plan, T(g["rays_o"], gpu), T(g["rays_d"], gpu), u_tables=u_tables, out={})
        info = rm.last_launch_info()
        assert info["final_kernel"] == ("k_final_stage<lt,K=7>" if densify == 2 else "k_final_stage<lt,K=5>"), info
        np.testing.assert_allclose(out["image"].cpu().numpy(), g["image"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(out["depth"].cpu().numpy(), g["depth"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), g["weights_sum"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("steps,f16", [([128], True), ([128, 64, 32], False), ([64, 32], True)])
def test_exact_early_out_is_bit_identical(gpu, steps, f16):
    """tuning.exact_early_out: once the transmittance of all 64 rays of a wave has underflowed to exactly 0 the last stage stops marching and
    the proposal stages stop evaluating densities (their remaining weights are 0 whatever the density).  On an opaque field (MLP gain 40:
    most rays saturate within a few samples) image, depth and weights_sum of the early-out instantiations equal those of the plain ones bit
    for bit -- also through the in-render feature stage, whose weights behind the early-out are written as zeros."""
    from sanerf_hq_amd import raymarching as rm, synth
    H, W = 96, 104
    model = product_model(synthetic_params(steps, heads=True, seed=3, gain=40.0), steps, True, gpu)
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    td = torch.float16 if f16 else torch.float32
    for feat in (False, True):
        plan = rm.RenderPlan(model, steps, td, feat_encoder=model.s_grid if feat else None)
        off = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, tuning=rm.Tuning(exact_early_out=1)).items()}
        on = rm.render_rays(plan, ro, rd, tile_w=W, tuning=rm.Tuning(exact_early_out=2), out={})
        assert float((off["weights_sum"] > 0.999).float().mean()) > 0.3, "the scene is meant to be mostly opaque"
        assert set(on) == set(off)
        for k in off:
            assert torch.equal(on[k], off[k]), (k, feat)


def test_opaque_field_vs_oracle_with_early_outs(gpu, orc):
    """The exact early-outs (proposal stages: always; last stage: forced on here) against the oracle, which evaluates every
    sample: on an opaque field the resampled indices are bit-exact and image / depth / weights_sum within the fp32 contract."""
    from sanerf_hq_amd import raymarching as rm
    steps = [128, 64, 32]
    params = synthetic_params(steps, seed=3, gain=40.0)
    model = product_model(params, steps, False, gpu)
    H, W = 24, 40
    _, _, ro, rd = camera_rays(orc, H, W, radius=1.0, elev=20.0, azim=30.0)
    plan = rm.RenderPlan(model, steps)
    out = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W, tuning=rm.Tuning(exact_early_out=2))     # no per-stage tensors: every early-out is active
    dbg = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W, want=("inds",), out={})
    want = orc.render(oracle_cfg(orc, params, steps), ro, rd, debug=True)
    assert float((want["weights_sum"] > 0.999).mean()) > 0.3
    for k in (1, 2):
        assert np.array_equal(dbg[f"inds{k}"].cpu().numpy(), want[f"inds{k}"])
    np.testing.assert_allclose(out["image"].cpu().numpy(), want["image"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["depth"].cpu().numpy(), want["depth"], rtol=1e-4, atol=1e-5)    # (gain-40 MLPs amplify the split-fp16 round-off: 3e-5 on 2 of 960 rays)
    np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), want["weights_sum"], rtol=0, atol=1e-5)
    # and the early-out render equals the per-stage-tensor render (whose proposal stages evaluate everything) in what both return
    for k in ("depth", "weights_sum"):
        assert torch.equal(out[k], dbg[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("steps,f16,gain", [([128, 64, 32], False, 4.0), ([128, 64, 32], True, 4.0), ([33, 17, 8], False, 4.0), ([64, 32], True, 40.0),
                                            ([127, 64, 32], False, 40.0)])
def test_two_samples_per_lane_in_the_proposal_stages_are_bit_identical(gpu, orc, steps, f16, gain):
    """tuning.prop_pair = 2 (k_prop_stage<..., UN = 2>: a lane evaluates two consecutive samples at once, their gathers and MLP chains interleave)
    against prop_pair = 1: every per-stage tensor (bins, weights, densities, resampled indices) and the image bit for bit -- on even and odd
    sample counts (the odd one's last sample goes through the one-sample body), on a semi-transparent and on an opaque field (gain 40: the
    exact early-out leaves in the middle of a pair), with and without the per-stage tensors (which switch the early-out off), in image
    order and in linear order beyond the small-batch kernels' range; and the indices still equal the oracle's."""
    from sanerf_hq_amd import raymarching as rm
    params = synthetic_params(steps, seed=5, gain=gain)
    model = product_model(params, steps, False, gpu)
    H, W = 40, 72
    _, _, ro, rd = camera_rays(orc, H, W, radius=1.0, elev=20.0, azim=30.0)
    plan = rm.RenderPlan(model, steps, torch.float16 if f16 else torch.float32)
    want = ("bins", "weights", "sigmas", "inds")
    for tile_w, kw in ((W, {}), (0, {"tuning_extra": {"prop_sp_max_rays": -1}})):
        extra = kw.get("tuning_extra", {})
        one = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=tile_w, want=want, tuning=rm.Tuning(prop_pair=1, **extra), out={}).items()}
        two = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=tile_w, want=want, tuning=rm.Tuning(prop_pair=2, **extra), out={})
        assert set(one) == set(two)
        for k in one:
            assert torch.equal(one[k], two[k]), (k, tile_w)
        plain1 = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=tile_w, tuning=rm.Tuning(prop_pair=1, **extra), out={}).items()}
        plain2 = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=tile_w, tuning=rm.Tuning(prop_pair=2, **extra), out={})
        for k in plain1:
            assert torch.equal(plain1[k], plain2[k]), (k, tile_w, "early-out active")
    ref = orc.render(oracle_cfg(orc, params, steps, table_f16=f16), ro, rd, debug=True)
    for k in range(1, len(steps)):
        assert np.array_equal(two[f"inds{k}"].cpu().numpy(), ref[f"inds{k}"])


@pytest.mark.gpu
@pytest.mark.parametrize("bands", [3, 4, 7])
def test_more_than_two_row_bands_are_bit_identical(gpu, bands):
    """tuning.band_streams = K > 2: K row bands dealt alternately to the two HIP streams (measured slower than two bands, kept as an A/B switch):
    every output, the per-stage tensors included, equals the single-stream render bit for bit, ragged last band included."""
    from sanerf_hq_amd import raymarching as rm, synth
    steps = [48, 24, 16]
    params = synthetic_params(steps, heads=False, seed=23)
    model = product_model(params, steps, False, gpu)
    H, W = 120, 72                                     # 7.5 tile rows of 16: the last band is ragged
    ro, rd = rm.generate_rays(synth.orbit_pose(1.1, 15.0, 75.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    plan = rm.RenderPlan(model, steps, torch.float32)
    want = ("inds", "weights", "bins")
    one = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, want=want, tuning=rm.Tuning(band_streams=1)).items()}
    many = {k: v.clone() for k, v in rm.render_rays(plan, ro, rd, tile_w=W, want=want, tuning=rm.Tuning(band_streams=bands), out={}).items()}
    assert set(one) == set(many) and "image" in one
    for k in one:
        assert torch.equal(one[k], many[k]), k


def test_render_writes_straight_into_a_packed_band(gpu, orc):
    """sn_render_io.out_stride (ABI 10): rgb | depth | weights_sum written as columns of one [N, 5] buffer -- the all-gather payload of
    dist.py / bench.py without a torch.cat -- equal, bit for bit, to the dense outputs; every kernel family that stores them (tile kernels,
    several-lanes-per-ray kernels, the any-field-size kernel)."""
    from sanerf_hq_amd import raymarching as rm
    for steps, H, W in (([128, 64, 32], 48, 64), ([128], 40, 40), ([16], 8, 24)):
        params = synthetic_params(steps, seed=9)
        model = product_model(params, steps, False, gpu)
        _, _, ro, rd = camera_rays(orc, H, W)
        plan = rm.RenderPlan(model, steps, torch.float16)
        dense = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W).items()}
        packed = torch.full((H * W, 7), -7.0, device=gpu)
        got = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W, packed=packed, out={})
        assert got["image"].data_ptr() == packed.data_ptr()
        assert torch.equal(packed[:, :3], dense["image"]) and torch.equal(packed[:, 3], dense["depth"]) and torch.equal(packed[:, 4], dense["weights_sum"])
        assert bool((packed[:, 5:] == -7.0).all())
        lin_dense = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, out={}).items()}      # linear ray order: the several-lanes-per-ray kernels
        lin = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, packed=torch.empty(H * W, 5, device=gpu), out={})
        assert torch.equal(lin["image"], lin_dense["image"]) and torch.equal(lin["depth"], lin_dense["depth"]) and torch.equal(lin["weights_sum"], lin_dense["weights_sum"])
