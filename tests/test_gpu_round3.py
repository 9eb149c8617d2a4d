"""GPU tests added in round 3: full-size checks of both table precisions and both schedules, the role-split final stage,
the cold torch-formulated routes of the training path, guards of the C ABI, the RCCL leg on one GPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import camera_rays, oracle_cfg, product_model, synthetic_params

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


def test_wide_mlp_training_route_needs_at_most_256_inputs(gpu):
    """the fused backward plans the transposed MLP, whose last width is the forward's dim_in (<= 256); a bias-free
    skip-free SkipConnMLP with 300 inputs must take the torch layers (and train) instead of failing in backward."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(0)
    for dim_in, fusable in ((143, True), (256, True), (300, False)):
        mlp = SkipConnMLP(dim_in, 2, 256, 3, skip_layers=[], bias=False).to(gpu)
        x = torch.randn(ops.WIDE_MLP_BACKWARD_MIN_ROWS, dim_in, device=gpu)
        assert ops.wide_mlp_fusable(x, list(mlp.net), []) == fusable, dim_in
        y = mlp(x)
        y.square().mean().backward()
        g = mlp.net[0].weight.grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


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


def test_torch_formulated_routes_of_the_training_path(gpu, monkeypatch):
    """Shapes beyond a kernel's limit take the reference's own torch formulation on the GPU (longer rays than the kernels hold
    in registers: T > 256 for the weights backward, > 512 / 2048 for the loss kernels; head MLPs that are not 256 wide).  Each
    such route is pinned here against the kernel route on shapes both can run, values and gradients."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.raymarching import raymarching as rmm
    from sanerf_hq_amd.nerf import renderer as R
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(1)
    N, Tn = 257, 48
    bins = torch.sort(torch.rand(N, Tn + 1, device=gpu), dim=-1).values
    sig = (torch.rand(N, Tn, device=gpu) * 8).requires_grad_(True)

    def grads(fn):
        sig.grad = None
        out = fn()
        (out * torch.linspace(0.5, 1.5, out.numel(), device=gpu).reshape(out.shape)).sum().backward()
        return out.detach().clone(), sig.grad.clone()

    w_k, g_k = grads(lambda: rm.weights_from_sigma(bins, sig, True))
    monkeypatch.setattr(rmm, "WEIGHTS_BACKWARD_MAX_T", 8)
    w_t, g_t = grads(lambda: rm.weights_from_sigma(bins, sig, True))
    monkeypatch.undo()
    np.testing.assert_allclose(w_t.cpu().numpy(), w_k.cpu().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(g_t.cpu().numpy(), g_k.cpu().numpy(), rtol=2e-4, atol=2e-6)
    with torch.no_grad():                                             # a 300-sample ray through the torch chain == the forward kernel
        b3 = torch.sort(torch.rand(33, 301, device=gpu), dim=-1).values
        s3 = torch.rand(33, 300, device=gpu) * 5
    s3g = s3.clone().requires_grad_(True)
    np.testing.assert_allclose(rm.weights_from_sigma(b3, s3g, True).detach().cpu().numpy(), rm.weights_from_sigma(b3, s3, True).cpu().numpy(), rtol=2e-5, atol=1e-6)

    # losses: kernel vs torch route
    wts = torch.softmax(torch.randn(N, Tn, device=gpu), dim=-1).requires_grad_(True)
    ref_b = torch.sort(torch.rand(N, 25, device=gpu), dim=-1).values
    ref_w = torch.softmax(torch.randn(N, 24, device=gpu), dim=-1)

    def loss_and_grad(fn):
        wts.grad = None
        v = fn()
        v.backward()
        return float(v), wts.grad.clone()

    lk, gk = loss_and_grad(lambda: R.proposal_loss([bins, ref_b], [wts, ref_w]))
    monkeypatch.setattr(rm, "PROPOSAL_LOSS_MAX_T", 0)
    lt, gt = loss_and_grad(lambda: R.proposal_loss([bins, ref_b], [wts, ref_w]))
    monkeypatch.undo()
    assert abs(lk - lt) <= 1e-5 * max(1.0, abs(lt))
    np.testing.assert_allclose(gk.cpu().numpy(), gt.cpu().numpy(), rtol=1e-3, atol=1e-7)
    dk, gdk = loss_and_grad(lambda: R.distort_loss(bins, wts))
    monkeypatch.setattr(rm, "DISTORT_LOSS_MAX_T", 0)
    dt, gdt = loss_and_grad(lambda: R.distort_loss(bins, wts))
    monkeypatch.undo()
    assert abs(dk - dt) <= 1e-5 * max(1.0, abs(dt))
    np.testing.assert_allclose(gdk.cpu().numpy(), gdt.cpu().numpy(), rtol=1e-3, atol=1e-7)

    # a head MLP that the matrix-core kernel does not instantiate (hidden width 128) runs as the torch module, same numbers
    mlp = torch.nn.Sequential(SkipConnMLP(40, 3, 128, 3, skip_layers=[], bias=True)).to(gpu)
    x = torch.randn(1000, 40, device=gpu)
    with torch.no_grad():
        assert torch.equal(R.NeRFRenderer._head_mlp(mlp, x), mlp(x))


def test_rccl_leg_on_one_gpu():
    """One-rank "nccl" (= RCCL) process group in a fresh process: init, all_reduce, the band render with the all-gather forced,
    PipelinedGather over 5 frames -- all equal to the plain render (tools/rccl_selftest.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selftest.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl selftest OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


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


def test_lazy_adam_updates_touched_elements_only(gpu):
    """SURVEY 8 f2 (opt-in, not the reference's optimiser): sanerf_hq_amd.optim.Adam(lazy=True) skips every element whose gradient is
    exactly zero in a step -- moments do not decay, the parameter does not move -- and applies the dense recipe (global step count
    in the bias corrections, as torch.optim.SparseAdam does) to the others.  Checked against that rule written in torch."""
    from sanerf_hq_amd.optim import Adam
    n, lr, b1, b2, eps = 4096 * 9 + 2, 1e-2, 0.9, 0.999, 1e-15
    rng = np.random.default_rng(3)
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(gpu)
    pa = torch.nn.Parameter(p0.clone())
    opt = Adam([dict(params=[pa], lr=lr, lazy=True)], eps=eps)
    p = p0.clone().double(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    never = torch.ones(n, dtype=torch.bool, device=gpu)
    for step in range(1, 7):
        g = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(gpu)
        mask = torch.from_numpy(rng.uniform(size=n) < 0.15).to(gpu)
        g = g * mask
        never &= ~mask
        before = pa.detach().clone()
        pa.grad = g.clone()
        opt.step()
        gd = g.double()
        m_new = m + (1 - b1) * (gd - m); v_new = v * b2 + (1 - b2) * gd * gd
        denom = v_new.sqrt() / (1 - b2 ** step) ** 0.5 + eps
        p_new = p - lr / (1 - b1 ** step) * m_new / denom
        m, v, p = torch.where(mask, m_new, m), torch.where(mask, v_new, v), torch.where(mask, p_new, p)
        assert torch.equal(pa.detach()[~mask], before[~mask]), "untouched elements must not move"
        assert float((pa.detach().double() - p).abs().max()) <= 2e-6 * float(p.abs().max()), step
    st = opt.state[pa]
    assert float((st["exp_avg"].double() - m).abs().max()) <= 1e-6 * float(m.abs().max())
    assert float((st["exp_avg_sq"].double() - v).abs().max()) <= 1e-6 * float(v.abs().max())
    assert torch.equal(pa.detach()[never], p0[never]) and never.any()
    with pytest.raises(RuntimeError, match="lazy"):
        bad = Adam([dict(params=[torch.nn.Parameter(p0.clone())], lr=lr, lazy=True)], eps=eps, weight_decay=1e-3)
        bad.param_groups[0]["params"][0].grad = torch.ones(n, device=gpu)
        bad.step()


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


def test_bench_line_schema():
    """bench.py prints ONE JSON line with the contract's fields; roofline carries bound / achieved / peak / unit / frac / traffic for the
    dominant kernel, the workload names BASELINE configs[1], the metric names the image rendered."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "rays/s" and d["higher_is_better"] is True
    assert "800x800" in d["metric"] and "configs[1]" in d["config"]["workload"] and d["vs_baseline"] is None
    assert abs(d["value"] - 640000 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "fabric_frac", "avg_kernel_ms", "shader_clock_mhz"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["max_abs_rgb_diff_vs_gpu"] < 1e-4
    assert d["also"] and "flat128_f32" in d["also"] and "ref_f16" in d["also"]


@pytest.mark.parametrize("din,dout,nlayers,skip,bias,ln,N", [
    (163, 256, 5, [2], True, True, 1000),     # SAM head MLP: LDS-DMA input tile, skip layer, LayerNorm, rows not a multiple of 128
    (143, 2, 3, [], False, False, 777),       # mask MLP: narrow last layer
    (143, 40, 3, [], False, False, 128 * 3),  # narrow last layer with two output tiles
    (64, 200, 2, [], True, False, 130),       # even input width (padded LDS tile), partial last output tile
    (700, 256, 4, [1, 2], True, False, 259),  # input too wide for LDS (read per k-step), two skip layers
    (17, 7, 1, [], True, False, 5),           # a single layer
])
def test_wide_mlp_just_in_time_kernel_is_bit_identical(gpu, monkeypatch, din, dout, nlayers, skip, bias, ln, N, experiments_build):
    """k_mlp_wide_j (operands made between the MFMAs of the previous k-step, layers handed over through `prev`) computes every
    output with the same products in the same order as k_mlp_wide: the two kernels must agree bit for bit in every input mode."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(din + dout)
    mlp = SkipConnMLP(din, dout, 256, nlayers, skip_layers=skip, bias=bias).to(gpu)
    norm = torch.nn.LayerNorm(dout).to(gpu) if ln else None
    x = torch.randn(N, din, device=gpu)
    from sanerf_hq_amd import _lib
    try:
        _lib.check(_lib.lib().sn_debug_set(b"wide_jit", 0), "debug_set")
        a = rm.mlp_forward(x, mlp, norm)
    finally:
        _lib.check(_lib.lib().sn_debug_set(b"wide_jit", 1), "debug_set")
    b = rm.mlp_forward(x, mlp, norm)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    with torch.no_grad():
        ref = mlp(x) if norm is None else norm(mlp(x))
    assert float((b - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("N,T_,n_inst,L", [(300, 32, 2, 16), (37, 128, 3, 16), (64, 16, 32, 16), (100, 8, 2, 6)])
def test_fused_mask_head_just_in_time_kernel_agrees(gpu, monkeypatch, N, T_, n_inst, L, experiments_build):
    """The fused mask head in k_mlp_wide_j<3>: lanes n and n + 32 share the corner rows of sample n (each fetches one 16-byte half of the
    rows of both levels of a k-step: half the rows per gather instruction), which permutes the first layer's input columns inside a
    k-step -- same products, another summation order within 16 terms: round-off agreement with k_mlp_wide<3>, not bit identity."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(N)
    enc = GridEncoder(input_dim=3, num_levels=L, level_dim=8, base_resolution=16, log2_hashmap_size=15, desired_resolution=512).to(gpu)
    with torch.no_grad():
        enc.embeddings.uniform_(-1.0, 1.0)
    E = 15
    mlp = SkipConnMLP(L * 8 + E, n_inst, 256, 3, skip_layers=[], bias=False).to(gpu)
    xyz = torch.rand(N, T_, 3, device=gpu) * 2.2 - 1.1        # some samples outside the grid's box
    extra = torch.randn(N, T_, E, device=gpu)
    w = torch.rand(N, T_, device=gpu)
    from sanerf_hq_amd import _lib
    try:
        _lib.check(_lib.lib().sn_debug_set(b"wide_jit", 0), "debug_set")
        a = rm.mask_head(w, xyz, extra, enc, mlp, 1.0)
    finally:
        _lib.check(_lib.lib().sn_debug_set(b"wide_jit", 1), "debug_set")
    b = rm.mask_head(w, xyz, extra, enc, mlp, 1.0)
    assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_random_field_sizes_through_the_size_agnostic_stage(gpu):
    """tools/fuzz_parity.py any: random fields of other sizes than the reference network's (levels, table size, hash / tiled, MLP depths and
    widths, geometry channels, schedules, fp16 / fp32 tables) against the oracle -- indices exact, RGB <= 1e-5, tile == linear order.  (A 120-case
    run of this sweep found what 20-case runs had not: an unrolled layer multiplying a never-written LDS row -- NaN -- by weight 0.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "any", "40", "13"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatching cases: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_random_head_mlps_and_mask_heads(gpu):
    """tools/fuzz_wide.py: random 256-wide stacks (widths, depths, skip layers, LayerNorm, row counts) -- k_mlp_wide_j bit-equal to k_mlp_wide and
    within 1e-4 of torch -- and random fused mask heads (levels, appended channels, samples per ray, outputs) against the unfused composition."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_wide.py"), "40", "17"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatching cases: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_other_field_sizes_with_the_feature_heads(gpu):
    """A field of other sizes WITH the SAM-feature and mask heads: the fused call renders RGB and hands the last stage's samples over (the
    in-render feature stage and the compaction exist next to the reference network's last stage only, so f_sam comes from the stand-alone
    grid_composite); everything must agree with the operator-chain route of the same model."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.encoding import get_encoder
    from sanerf_hq_amd.nerf.network import MLP
    steps = [48, 24, 16]
    params = synthetic_params(steps, heads=True, seed=77)
    model = product_model(params, steps, True, gpu)
    torch.manual_seed(3)
    model.grid, d = get_encoder("hashgrid", input_dim=3, level_dim=2, num_levels=10, log2_hashmap_size=15, desired_resolution=512)
    model.grid_mlp = MLP(d, 16, 40, 2, bias=False)            # 15 geometry channels: what the heads of the reference network expect
    model.view_mlp = MLP(31, 3, 24, 3, bias=False)
    model = model.to(gpu).eval()
    with torch.no_grad():
        model.grid.embeddings.uniform_(-1.0, 1.0)
    model.opt.compact_live = True                             # asked for, not available for this field: must be ignored, not fatal
    assert model._fused_kind() == "any" and not model._sam_fusable()
    H = W = 40
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    with torch.no_grad():
        a = model.render(ro, rd, staged=False, perturb=False, return_feats=1, return_mask=1, H=H, W=W, tile_w=W)
        a = {k: v.clone() for k, v in a.items() if torch.is_tensor(v)}
        model.standard_field = False
        b = model.render(ro, rd, staged=False, perturb=False, return_feats=1, return_mask=1, H=H, W=W, tile_w=W)
    for k, tol in (("image", 2e-5), ("depth", 1e-4), ("samvit", 1e-4), ("instance_mask_logits", 1e-4)):
        assert torch.isfinite(a[k]).all(), k
        assert float((a[k].reshape(b[k].shape) - b[k]).abs().max()) <= tol * max(1.0, float(b[k].abs().max())), k
