"""GPU: every stand-alone operator of the C ABI (through the Python operator modules, which are
thin ctypes callers) against the CPU oracle on the same seeded inputs.
Bars: bit-exact for integer outputs (sample indices) and for arithmetic that follows the shared
fp32 recipe; <= a few ulp where device libm (sin/cos) or atomics ordering is involved."""
import numpy as np
import pytest
import torch

from helpers import camera_rays, product_model, synthetic_params

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


GRID_CASES = [
    dict(L=16, C=2, log2T=19, desired=4096, gridtype=0, ac=False, interp=0),   # main field
    dict(L=16, C=8, log2T=19, desired=512, gridtype=0, ac=False, interp=0),    # SAM / mask grids
    dict(L=5, C=2, log2T=17, desired=128, gridtype=0, ac=False, interp=0),     # proposal 0
    dict(L=8, C=2, log2T=14, desired=2048, gridtype=0, ac=False, interp=0),    # config C1
    dict(L=4, C=4, log2T=10, desired=64, gridtype=1, ac=False, interp=0),      # tiled, generic modulo
    dict(L=4, C=1, log2T=12, desired=100, gridtype=0, ac=True, interp=1),      # align_corners + smoothstep
]


def _grid_setup(orc, cfg, B, seed, dev):
    rng = np.random.default_rng(seed)
    offs, pls = orc.grid_layout(3, cfg["L"], cfg["C"], 2, 16, cfg["log2T"], cfg["desired"])
    emb = rng.uniform(-1, 1, (int(offs[-1]), cfg["C"])).astype(np.float32)
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    x[0] = 0; x[1] = 1; x[2] = [0.5, 0.5, 0.5]; x[3] = [1.25, 0.5, 0.5]; x[4] = [0.2, -0.01, 0.3]   # corners + out of range
    return offs, pls, emb, x


@pytest.mark.parametrize("cfg", GRID_CASES)
def test_grid_forward_matches_oracle(gpu, orc, cfg):
    from sanerf_hq_amd.gridencoder import grid_encode
    offs, pls, emb, x = _grid_setup(orc, cfg, 4099, 11, gpu)
    want, want_dd = orc.grid_encode_forward(x, emb, offs, pls, 16, True, cfg["gridtype"], cfg["ac"], cfg["interp"])
    xt = T(x, gpu).requires_grad_(True)
    got = grid_encode(xt, T(emb, gpu), T(offs, gpu), pls, 16, True, cfg["gridtype"], cfg["ac"], cfg["interp"])
    assert got.shape == (4099, cfg["L"] * cfg["C"])
    assert np.array_equal(got.detach().cpu().numpy(), want), "same fmaf chain => bit-identical"
    # input gradient through dy_dx (kernel_input_backward)
    g = np.random.default_rng(12).standard_normal(want.shape).astype(np.float32)
    got.backward(T(g, gpu))
    _, gi = orc.grid_encode_backward(g, x, emb, offs, pls, 16, want_dd, cfg["gridtype"], cfg["ac"], cfg["interp"])
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gi, rtol=1e-5, atol=1e-4)


def test_grid_forward_fp16_table_and_max_level(gpu, orc):
    from sanerf_hq_amd.gridencoder import grid_encode
    cfg = GRID_CASES[0]
    offs, pls, emb, x = _grid_setup(orc, cfg, 2050, 13, gpu)
    emb16 = emb.astype(np.float16)
    want, _ = orc.grid_encode_forward(x, emb16, offs, pls, 16)
    got = grid_encode(T(x, gpu), T(emb16, gpu), T(offs, gpu), pls, 16)
    assert got.dtype == torch.float16                                   # grid.py:49: outputs take the table's dtype
    assert np.array_equal(got.float().cpu().numpy(), want.astype(np.float16).astype(np.float32))
    want, _ = orc.grid_encode_forward(x, emb, offs, pls, 16, max_level=5)
    got = grid_encode(T(x, gpu), T(emb, gpu), T(offs, gpu), pls, 16, False, 0, False, 0, 5)
    assert np.array_equal(got.cpu().numpy(), want) and float(got[:, 10:].abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [GRID_CASES[1], GRID_CASES[2], GRID_CASES[4]])
def test_grid_backward_matches_oracle(gpu, orc, cfg):
    from sanerf_hq_amd.gridencoder import grid_encode
    offs, pls, emb, x = _grid_setup(orc, cfg, 3001, 21, gpu)
    g = np.random.default_rng(22).standard_normal((3001, cfg["L"] * cfg["C"])).astype(np.float32)
    want, _ = orc.grid_encode_backward(g, x, emb, offs, pls, 16, None, cfg["gridtype"], cfg["ac"], cfg["interp"])
    et = T(emb, gpu).requires_grad_(True)
    out = grid_encode(T(x, gpu), et, T(offs, gpu), pls, 16, False, cfg["gridtype"], cfg["ac"], cfg["interp"])
    out.backward(T(g, gpu))
    got = et.grad.cpu().numpy()
    # float atomics commute only up to rounding: compare with a tolerance scaled by the row's |grad| mass
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    assert np.array_equal(got == 0, want == 0), "exactly the same rows are touched"


@pytest.mark.parametrize("cfg,B", [(GRID_CASES[1], 3001), (GRID_CASES[1], 70000), (GRID_CASES[2], 40000), (GRID_CASES[4], 9000),
                                   (GRID_CASES[5], 5000)])
def test_grid_backward_binned_matches_oracle(gpu, orc, cfg, B):
    """The atomics-free scatter (one partition pass by row bin + LDS accumulation, one owner per table row: grid_binned.hip) against the oracle, including heavy row
    sharing (coherent samples on coarse levels), out-of-range samples and the tiled / align_corners variants."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.gridencoder import grid_encode
    offs, pls, emb, x = _grid_setup(orc, cfg, B, 23, gpu)
    x[B // 2:] = np.clip(x[B // 2:B // 2 + 1] + np.random.default_rng(24).normal(0, 0.01, (B - B // 2, 3)), -0.05, 1.05).astype(np.float32)
    g = np.random.default_rng(25).standard_normal((B, cfg["L"] * cfg["C"])).astype(np.float32)
    want, _ = orc.grid_encode_backward(g, x, emb, offs, pls, 16, None, cfg["gridtype"], cfg["ac"], cfg["interp"])
    old = ops.GRID_BACKWARD_MODE
    res = {}
    try:
        for mode in ("binned", "atomic"):
            ops.GRID_BACKWARD_MODE = mode
            et = T(emb, gpu).requires_grad_(True)
            out = grid_encode(T(x, gpu), et, T(offs, gpu), pls, 16, False, cfg["gridtype"], cfg["ac"], cfg["interp"])
            out.backward(T(g, gpu))
            res[mode] = et.grad.cpu().numpy()
    finally:
        ops.GRID_BACKWARD_MODE = old
    scale = np.abs(want).max()
    for mode, got in res.items():
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * scale, err_msg=mode)
        assert np.array_equal(got == 0, want == 0) or (np.logical_xor(got == 0, want == 0).sum() <= 4), mode


def test_grid_empty_and_ragged_batches(gpu, orc):
    from sanerf_hq_amd.gridencoder import GridEncoder
    enc = GridEncoder(num_levels=4, log2_hashmap_size=10, desired_resolution=64).to(gpu)
    assert enc(torch.empty(0, 3, device=gpu)).shape == (0, 8)
    for B in (1, 63, 64, 65, 257):
        x = torch.rand(B, 3, device=gpu) * 2 - 1
        y = enc(x, bound=1)
        want, _ = orc.grid_encode_forward(((x + 1) / 2).cpu().numpy(), enc.embeddings.detach().cpu().numpy(),
                                          enc.offsets.cpu().numpy(), enc.per_level_scale, 16)
        assert np.array_equal(y.detach().cpu().numpy(), want)
    # leading dimensions are preserved (grid.py:159-163)
    assert enc(torch.rand(5, 7, 3, device=gpu)).shape == (5, 7, 8)


def test_grid_tv_and_weight_decay(gpu, orc):
    from sanerf_hq_amd.gridencoder import GridEncoder
    enc = GridEncoder(num_levels=6, level_dim=2, log2_hashmap_size=12, desired_resolution=256).to(gpu)
    rng = np.random.default_rng(31)
    emb = rng.uniform(-1, 1, tuple(enc.embeddings.shape)).astype(np.float32)
    enc.embeddings.data.copy_(T(emb, gpu))
    with pytest.raises(ValueError):
        enc.grad_weight_decay(0.1)                       # grid.py:200-201: needs a grad first
    g0 = rng.standard_normal(emb.shape).astype(np.float32)
    enc.embeddings.grad = T(g0, gpu)
    enc.grad_weight_decay(0.1)
    want = orc.grad_weight_decay(emb, g0.copy(), enc.offsets.cpu().numpy(), 0.1)
    np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    x = rng.uniform(-1, 1, (5000, 3)).astype(np.float32)
    enc.embeddings.grad = T(g0, gpu)
    enc.grad_total_variation(1e-3, T(x, gpu), bound=1)
    want = orc.grad_total_variation((x + 1) / 2, emb, g0.copy(), enc.offsets.cpu().numpy(), 1e-3, enc.per_level_scale, 16)
    np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-6)
    enc.embeddings.grad = T(g0, gpu)
    enc.grad_total_variation(1e-7)                       # default: 1e6 random points (grid.py:172)
    assert torch.isfinite(enc.embeddings.grad).all()


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_matches_oracle(gpu, orc, degree):
    from sanerf_hq_amd.shencoder import sh_encode
    rng = np.random.default_rng(40 + degree)
    d = rng.standard_normal((1000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    want, want_dd = orc.sh_encode_forward(d, degree, True)
    dt = T(d, gpu).requires_grad_(True)
    got = sh_encode(dt, degree, True)
    # the kernel sums monomials (tools/gen_sh.py), the oracle evaluates the reference's factored forms
    # (cancellation grows with the polynomial degree: coefficients reach ~75 at degree 8)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=0, atol=4e-6 if degree <= 6 else 2e-5)
    g = rng.standard_normal(want.shape).astype(np.float32)
    got.backward(T(g, gpu))
    gi = orc.sh_encode_backward(g, d, degree, want_dd)
    np.testing.assert_allclose(dt.grad.cpu().numpy(), gi, rtol=1e-4, atol=2e-4)


def test_sh_encoder_module_normalises(gpu, orc):
    from sanerf_hq_amd.shencoder import SHEncoder
    v = torch.randn(17, 5, 3, device=gpu) * 3
    y = SHEncoder(degree=4)(v)
    assert y.shape == (17, 5, 16)
    n = (v / v.norm(dim=-1, keepdim=True)).reshape(-1, 3).cpu().numpy()
    want, _ = orc.sh_encode_forward(n, 4)
    np.testing.assert_allclose(y.reshape(-1, 16).cpu().numpy(), want, atol=3e-6)


@pytest.mark.parametrize("deg", [1, 4, 6, 10])
def test_freq_matches_oracle(gpu, orc, deg):
    from sanerf_hq_amd.freqencoder import FreqEncoder
    rng = np.random.default_rng(50)
    x = rng.uniform(-1, 1, (513, 3)).astype(np.float32)
    enc = FreqEncoder(3, deg)
    xt = T(x, gpu).requires_grad_(True)
    y = enc(xt)
    want = orc.freq_encode_forward(x, deg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), want, rtol=0, atol=2e-6)   # device vs host sinf/cosf
    g = rng.standard_normal(want.shape).astype(np.float32)
    y.backward(T(g, gpu))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), orc.freq_encode_backward(g, want, 3, deg), rtol=1e-4, atol=1e-3)


def test_generate_rays_and_get_rays(gpu, orc):
    from sanerf_hq_amd import synth
    from sanerf_hq_amd.nerf import get_rays
    from sanerf_hq_amd.raymarching import generate_rays
    H, W = 37, 53
    pose = synth.orbit_pose(1.3, 35.0, -50.0)
    intr = synth.pinhole_intrinsics(H, W)
    ro, rd = generate_rays(pose, intr, H, W, device=gpu)
    wo, wd = orc.generate_rays(pose, *intr, H, W)
    assert np.array_equal(ro.cpu().numpy(), wo) and np.array_equal(rd.cpu().numpy(), wd)
    ro2, rd2 = generate_rays(pose, intr, H, W, device=gpu, row_begin=16, row_end=32)     # a rank's band
    assert torch.equal(rd2, rd[16 * W:32 * W]) and torch.equal(ro2, ro[16 * W:32 * W])
    res = get_rays(torch.from_numpy(pose)[None].to(gpu), np.array(intr, dtype=np.float32), H, W, -1)
    assert torch.equal(res["rays_d"], rd) and res["inds_coarse"].shape == (H * W,)
    sub = get_rays(torch.from_numpy(pose)[None].to(gpu), np.array(intr, dtype=np.float32), H, W, 64, random_sample=True)
    idx = sub["j"] * W + sub["i"]
    assert torch.equal(sub["rays_d"], rd[idx])
    # pixel-subset branch against the reference's own output (tests/golden/units.npz, coords = (row, col))
    from helpers import golden
    g = golden("units")
    fx, fy, cx, cy, Hb, Wb = g["rays_b_intr"]
    sub = get_rays(torch.from_numpy(g["rays_b_pose"])[None].to(gpu), np.array([fx, fy, cx, cy], dtype=np.float32), int(Hb), int(Wb),
                   len(g["rays_b_coords"]), coords=torch.from_numpy(g["rays_b_coords"]).to(gpu))
    assert np.array_equal(sub["rays_o"].cpu().numpy(), g["rays_b_sub_o"])
    np.testing.assert_allclose(sub["rays_d"].cpu().numpy(), g["rays_b_sub_d"], rtol=0, atol=2e-7)
    assert np.array_equal(sub["i"].cpu().numpy(), g["rays_b_sub_i"]) and np.array_equal(sub["j"].cpu().numpy(), g["rays_b_sub_j"])
    assert np.array_equal(sub["inds_coarse"].cpu().numpy(), g["rays_b_sub_inds_coarse"])
    # error-map / incoherent-mask draws (utils.py:214-259): device-side multinomial, pixels land inside the drawn coarse cells
    M = 8
    mask = torch.zeros(M * M, device=gpu)
    hot = torch.tensor([3 * M + 5, 6 * M + 1, 0], device=gpu)
    mask[hot] = 1.0
    tposes, tintr = torch.from_numpy(pose)[None].to(gpu), np.array(intr, dtype=np.float32)
    em = get_rays(tposes, tintr, H, W, 3, patch_size=1, incoherent_mask=mask, incoherent_mask_size=M)
    assert em["inds_coarse"].shape == (1, 3) and set(em["inds_coarse"][0].tolist()) == set(hot.tolist())      # without replacement
    gx, gy = em["inds_coarse"][0] // M, em["inds_coarse"][0] % M
    assert bool(((em["j"] >= (gx * H / M).long()) & (em["j"] <= ((gx + 1) * H / M).long())).all())
    assert bool(((em["i"] >= (gy * W / M).long()) & (em["i"] <= ((gy + 1) * W / M).long())).all())
    assert torch.equal(em["rays_d"], rd[em["j"] * W + em["i"]])
    pt = get_rays(tposes, tintr, H, W, 16, patch_size=4, incoherent_mask=mask, include_incoherent_region=True, incoherent_mask_size=M)
    assert pt["rays_d"].shape == (16, 3) and int(pt["j"].max() - pt["j"].min()) == 3 and int(pt["i"].max() - pt["i"].min()) == 3
    with pytest.raises(RuntimeError, match="incoherent_mask"):
        get_rays(tposes, tintr, H, W, 4, patch_size=1)


def test_near_far_contract_bit_exact(gpu, orc):
    from sanerf_hq_amd import raymarching as rm
    from helpers import golden
    g = golden("units")
    for tag in ("big", "small"):
        n, f = rm.near_far_from_aabb(T(g["nf_o"], gpu), T(g["nf_d"], gpu), torch.from_numpy(g[f"nf_{tag}_aabb"]), 0.2)
        assert np.array_equal(n.cpu().numpy(), g[f"nf_{tag}_near"]) and np.array_equal(f.cpu().numpy(), g[f"nf_{tag}_far"])
    z = rm.contract(T(g["contract_x"], gpu))
    assert np.array_equal(z.cpu().numpy(), g["contract_z"])             # equals the REFERENCE's torch output
    assert rm.contract(torch.rand(3, 5, 3, device=gpu)).shape == (3, 5, 3)


@pytest.mark.parametrize("tag,T_", [("a", 65), ("b", 33), ("c", 17), ("d", 33)])
def test_sample_pdf_indices_bit_exact_vs_oracle(gpu, orc, tag, T_):
    from sanerf_hq_amd import raymarching as rm
    from helpers import golden
    g = golden("units")
    bins, w, u = g[f"pdf_{tag}_bins"], g[f"pdf_{tag}_w"], g[f"pdf_{tag}_u"]
    want_b, want_i = orc.sample_pdf(bins, w, T_, u=u)
    got_b, got_i = rm.sample_pdf(T(bins, gpu), T(w, gpu), T_, return_inds=True, u=T(u, gpu))
    assert np.array_equal(got_i.cpu().numpy(), want_i), "sample indices must be bit-exact"
    assert np.array_equal(got_b.cpu().numpy(), want_b)
    got_b2, got_i2 = rm.sample_pdf(T(bins, gpu), T(w, gpu), T_, return_inds=True)      # kernel's own linspace recipe
    want_b2, want_i2 = orc.sample_pdf(bins, w, T_)
    assert np.array_equal(got_i2.cpu().numpy(), want_i2) and np.array_equal(got_b2.cpu().numpy(), want_b2)
    # against the reference's torch.searchsorted: equal up to the enumerated ties (oracle/README.md)
    assert (got_i.cpu().numpy() != g[f"pdf_{tag}_inds"]).sum() <= 2
    # perturb=True draws per-ray u: results stay sorted and inside the bin range
    pb = rm.sample_pdf(T(bins, gpu), T(w, gpu), T_, perturb=True)
    assert bool((pb[:, 1:] >= pb[:, :-1]).all()) and float(pb.min()) >= float(bins.min()) and float(pb.max()) <= float(bins.max())


def test_sample_pdf_large_random(gpu, orc):
    from sanerf_hq_amd import raymarching as rm
    rng = np.random.default_rng(60)
    N, T0, T_ = 20000, 128, 65
    w = (rng.uniform(0, 1, (N, T0)) ** 6).astype(np.float32)
    b = np.sort(rng.uniform(0, 1, (N, T0 + 1)), axis=1).astype(np.float32)
    want_b, want_i = orc.sample_pdf(b, w, T_)
    got_b, got_i = rm.sample_pdf(T(b, gpu), T(w, gpu), T_, return_inds=True)
    assert np.array_equal(got_i.cpu().numpy(), want_i) and np.array_equal(got_b.cpu().numpy(), want_b)


@pytest.mark.parametrize("T_,opaque", [(1, True), (33, True), (64, False), (128, True), (200, False), (256, True), (257, False), (300, True), (1000, True),
                                       (4097, False)])
def test_weights_from_sigma_autograd(gpu, orc, T_, opaque):
    """The autograd form of weights_from_sigma against torch's own derivative of renderer.py:308-325 evaluated in fp64
    (delta*sigma -> alpha, exclusive cumsum -> transmittance, product, nan_to_num).  Rays of more than 256 samples take the backward kernel's
    long-ray instantiation (segment prefixes in LDS, terms evaluated again on the way back): no torch route at any length."""
    from sanerf_hq_amd import raymarching as rm
    rng = np.random.default_rng(7 + T_)
    N = 777 if T_ <= 1000 else 131
    rb = np.sort(rng.uniform(0.2, 30, (N, T_ + 1)), axis=1).astype(np.float32)
    sg = np.exp(rng.uniform(-6, 3, (N, T_))).astype(np.float32)
    go = rng.standard_normal((N, T_)).astype(np.float32)
    s1 = T(sg, gpu).requires_grad_(True)
    w = rm.weights_from_sigma(T(rb, gpu), s1, opaque)
    if T_ <= rm.raymarching.WEIGHTS_BACKWARD_MAX_T:
        assert np.array_equal(w.detach().cpu().numpy(), orc.weights_from_sigma(rb, sg, opaque))
    else:   # longer rays fall back to torch's chain (fp32 cumsum on the device): close, not bit-equal
        np.testing.assert_allclose(w.detach().cpu().numpy(), orc.weights_from_sigma(rb, sg, opaque), rtol=2e-4, atol=1e-7)
    w.backward(T(go, gpu))
    s2 = T(sg, gpu).double().requires_grad_(True)
    rbd = T(rb, gpu).double()
    ds = (rbd[..., 1:] - rbd[..., :-1]) * s2
    if opaque:
        ds = torch.cat([ds[..., :-1], torch.full_like(ds[..., -1:], torch.inf)], dim=-1)
    alphas = 1 - torch.exp(-ds)
    trans = torch.cumsum(ds[..., :-1], dim=-1)
    trans = torch.exp(-torch.cat([torch.zeros_like(ds[..., :1]), trans], dim=-1))
    ((alphas * trans).nan_to_num(0) * T(go, gpu).double()).sum().backward()
    ref = s2.grad.float()
    err = float((s1.grad - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    assert err < (2e-5 if T_ <= rm.raymarching.WEIGHTS_BACKWARD_MAX_T else 1e-3), err
    if opaque:
        assert float(s1.grad[:, -1].abs().max()) == 0.0, "the opaque last sample is a constant"


def test_sample_positions_match_the_fused_renderer(gpu, orc):
    """rm.sample_positions (the training path's geometry kernel) == what the fused renderer computes for the same bins:
    bit-equal to its per-sample positions, and real_bins/rays_t equal to the torch expressions of renderer.py:277-282."""
    from sanerf_hq_amd import raymarching as rm
    steps = [48]
    params = synthetic_params(steps, seed=12)
    model = product_model(params, steps, False, gpu)
    _, _, ro, rd = camera_rays(orc, 17, 29, radius=1.3, elev=-10.0, azim=77.0)
    plan = rm.RenderPlan(model, steps)
    out = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, want=("bins", "xyzs_last"), out={})
    nears, fars = rm.near_far_from_aabb(T(ro, gpu), T(rd, gpu), model.aabb_infer, model.min_near)
    real_bins, rays_t, xyzs = rm.sample_positions(T(ro, gpu), T(rd, gpu), nears, fars, out["bins0"], contract=True)
    assert torch.equal(xyzs, out["xyzs_last"])

    def g(x):
        return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))

    def ginv(x):
        return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))
    rb = ginv(g(nears) * (1 - out["bins0"]) + g(fars) * out["bins0"])
    np.testing.assert_allclose(real_bins.cpu().numpy(), rb.cpu().numpy(), rtol=2e-6, atol=0)
    np.testing.assert_allclose(rays_t.cpu().numpy(), ((rb[:, 1:] + rb[:, :-1]) / 2).cpu().numpy(), rtol=2e-6, atol=0)
    _, _, flat = rm.sample_positions(T(ro, gpu), T(rd, gpu), nears, fars, out["bins0"], contract=False)
    assert torch.equal(rm.contract(flat), xyzs)


def test_wave_per_ray_and_lane_per_ray_operators_agree(gpu, orc):
    """sample_pdf / weights_from_sigma switch to one wave per ray up to 65 536 rays: same bits as the one-lane kernels
    (the fp64 running sums keep their order), including perturbed per-ray u rows and step counts that are not
    multiples of 64."""
    from sanerf_hq_amd import raymarching as rm
    rng = np.random.default_rng(11)
    big, small = 66000, 1500                                   # lane-per-ray kernel, wave-per-ray kernel
    for T0, T_ in ((128, 65), (37, 90), (64, 33)):
        w = (rng.uniform(0, 1, (big, T0)) ** 5).astype(np.float32)
        b = np.sort(rng.uniform(0, 1, (big, T0 + 1)), axis=1).astype(np.float32)
        u = np.sort(rng.uniform(0, 1, (big, T_)), axis=1).astype(np.float32)
        for uu in (None, u):
            gb, gi = rm.sample_pdf(T(b, gpu), T(w, gpu), T_, return_inds=True, u=None if uu is None else T(uu, gpu))
            sb, si = rm.sample_pdf(T(b[:small], gpu), T(w[:small], gpu), T_, return_inds=True, u=None if uu is None else T(uu[:small], gpu))
            assert torch.equal(gb[:small], sb) and torch.equal(gi[:small], si)
        rb = np.sort(rng.uniform(0.2, 40, (big, T0 + 1)), axis=1).astype(np.float32)
        sg = np.exp(rng.uniform(-6, 4, (big, T0))).astype(np.float32)
        for opaque in (True, False):
            assert torch.equal(rm.weights_from_sigma(T(rb, gpu), T(sg, gpu), opaque)[:small],
                               rm.weights_from_sigma(T(rb[:small], gpu), T(sg[:small], gpu), opaque))


@pytest.mark.parametrize("T_,Tr", [(128, 32), (64, 32), (33, 17), (1, 5), (200, 300), (512, 512), (600, 700), (1500, 64), (100, 1300), (5000, 3000)])
def test_proposal_loss_kernel_matches_the_torch_expression(gpu, T_, Tr):
    """sn_rm_proposal_loss (forward value and gradient w.r.t. the proposal weights) against the torch statement of
    nerf/renderer.py:30-57 in fp64, on sorted random bins that do and do not line up between the two stages.  More than 512 samples per ray
    (either stage) take the long-ray instantiation: the ray's arrays in a workspace instead of LDS, a wave walking several rays."""
    from sanerf_hq_amd import raymarching as rm
    rng = np.random.default_rng(T_ * 1000 + Tr)
    N = 513 if max(T_, Tr) <= 1500 else 4099            # (the long launch holds at most 64 MiB of arrays: 4099 rays of 5000 samples share them)
    b = np.sort(rng.uniform(0, 1, (N, T_ + 1)), axis=1).astype(np.float32)
    rb = np.sort(rng.uniform(0, 1, (N, Tr + 1)), axis=1).astype(np.float32)
    rb[: N // 4, ::3] = b[: N // 4, : rb[:, ::3].shape[1]] if T_ + 1 >= rb[:, ::3].shape[1] else rb[: N // 4, ::3]   # shared edges
    rb = np.sort(rb, axis=1)
    w = (rng.uniform(0, 1, (N, T_)) ** 3).astype(np.float32); w /= w.sum(1, keepdims=True)
    rw = (rng.uniform(0, 1, (N, Tr)) ** 3).astype(np.float32); rw /= rw.sum(1, keepdims=True)
    w1 = T(w, gpu).requires_grad_(True)
    l1 = rm.proposal_loss_stage(T(b, gpu), w1, T(rb, gpu), T(rw, gpu))
    (l1 * 3.0).backward()
    w2 = T(w, gpu).double().requires_grad_(True)
    bd, rbd, rwd = T(b, gpu).double(), T(rb, gpu).double(), T(rw, gpu).double()
    cum = torch.cat([torch.zeros_like(w2[..., :1]), torch.cumsum(w2, dim=-1)], dim=-1)
    last = T_ - 1
    lo = (torch.searchsorted(bd[..., :-1].contiguous(), rbd[..., :-1].contiguous(), right=True) - 1).clamp(0, last)
    hi = torch.searchsorted(bd[..., 1:].contiguous(), rbd[..., 1:].contiguous(), right=True).clamp(0, last)
    bound = torch.take_along_dim(cum[..., 1:], hi, dim=-1) - torch.take_along_dim(cum[..., :-1], lo, dim=-1)
    l2 = ((rwd - bound).clamp(min=0) ** 2 / (rwd + 1e-8)).mean()
    (l2 * 3.0).backward()
    assert abs(float(l1) - float(l2)) <= 2e-6 * max(1.0, abs(float(l2))) + 1e-9
    # rw - bound cancels in fp32 (cum ~ 1, differences ~ 1e-2): the per-term error of the fp32 statement itself is ~1e-4
    # relative, so the bar is the project's gradient bar (relative L2 < 1e-3), not ulps
    ref = w2.grad.float()
    assert float((w1.grad - ref).norm()) <= 1e-3 * float(ref.norm()) + 1e-12
    assert float((w1.grad - ref).abs().max()) <= 5e-3 * float(ref.abs().max()) + 1e-12
    l3 = rm.proposal_loss_stage(T(b, gpu), T(w, gpu), T(rb, gpu), T(rw, gpu))
    assert float(l3) == float(l1), "deterministic"


@pytest.mark.parametrize("C_,E,B", [(8, 15, 100003), (2, 3, 777), (4, 1, 64)])
def test_grid_forward_cat_equals_encode_then_cat(gpu, C_, E, B):
    """GridEncoder.forward_cat (sn_grid_encode_forward_cat, the mask head's MLP input) == cat([encoder(x), extra]) bit
    for bit, including a last partial 64-row block and row widths that are not multiples of 4 floats."""
    from sanerf_hq_amd.ops import GridEncoder
    torch.manual_seed(C_ * 10 + E)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=C_, base_resolution=16, log2_hashmap_size=15, desired_resolution=512).to(gpu)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = (torch.rand(B, 3, device=gpu) * 4.4 - 2.2)          # some samples outside [-2, 2]: zero rows (gridencoder.cu:113-130)
    extra = torch.randn(B, E, device=gpu)
    with torch.no_grad():
        want = torch.cat([enc(x, bound=2), extra], dim=-1)
    got = enc.forward_cat(x, extra, bound=2)
    assert got.shape == (B, 16 * C_ + E) and torch.equal(got, want)
    got3 = enc.forward_cat(x.view(-1, 1, 3), extra.view(-1, 1, E), bound=2)
    assert got3.shape == (B, 1, 16 * C_ + E) and torch.equal(got3.view(B, -1), want)


@pytest.mark.parametrize("T_", [1, 32, 64, 100, 200, 2048, 2500])
def test_distort_loss_kernel(gpu, T_):
    """sn_rm_distort_loss against the O(T^2) definition sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 d_i in fp64 (value and
    autograd gradient), and against the cumulative-sum form the CPU path uses (what eff_distloss computes)."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.nerf import renderer as R
    rng = np.random.default_rng(T_)
    N = 301 if T_ <= 200 else 9                         # (the fp64 definition below is O(N T^2) memory; beyond 2048 samples the kernel reads in place)
    b = np.sort(rng.uniform(0, 1, (N, T_ + 1)), axis=1).astype(np.float32)
    w = (rng.uniform(0, 1, (N, T_)) ** 4).astype(np.float32)
    w1 = T(w, gpu).requires_grad_(True)
    l1 = rm.distort_loss(T(b, gpu), w1)
    (l1 * 0.7).backward()
    w2 = T(w, gpu).double().requires_grad_(True)
    bd = T(b, gpu).double()
    d = bd[:, 1:] - bd[:, :-1]
    m = bd[:, :-1] + d / 2
    l2 = ((w2[:, :, None] * w2[:, None, :] * (m[:, :, None] - m[:, None, :]).abs()).sum((1, 2)) + (w2 * w2 * d).sum(1) / 3).mean()
    (l2 * 0.7).backward()
    assert abs(float(l1) - float(l2)) <= 1e-5 * abs(float(l2)) + 1e-9
    assert float((w1.grad - w2.grad.float()).abs().max()) <= 1e-5 * float(w2.grad.abs().max()) + 1e-12
    w3 = torch.from_numpy(w).double().requires_grad_(True)
    l3 = R.distort_loss(torch.from_numpy(b).double(), w3)      # CPU tensors: the cumulative-sum statement
    (l3 * 0.7).backward()
    assert abs(float(l3) - float(l2)) <= 1e-9 * abs(float(l2)) + 1e-12
    assert float((w3.grad - w2.grad.cpu()).abs().max()) <= 1e-9 * float(w2.grad.abs().max()) + 1e-14


def _distort_cases(T_, seed):
    """Intervals and weights that stress the O(T) kernel: random, zero-width intervals (repeated edges), all mass in one bin, one-hot mass at
    either end, equal weights on equal intervals, a ray with no mass at all."""
    rng = np.random.default_rng(seed)
    N = 64
    b = np.sort(rng.uniform(0, 1, (N, T_ + 1)), axis=1)
    w = rng.uniform(0, 1, (N, T_)) ** 4
    b[1, T_ // 3: 2 * T_ // 3 + 1] = b[1, T_ // 3]                         # a run of zero-width intervals
    b[2] = np.repeat(b[2, ::2], 2)[:T_ + 1] if T_ > 1 else b[2]              # every other interval has zero width
    w[3] = 0.0; w[3, T_ // 2] = 1.0                                          # all mass in one bin
    w[4] = 0.0; w[4, 0] = 1.0
    w[5] = 0.0; w[5, -1] = 1.0
    b[6] = np.linspace(0, 1, T_ + 1); w[6] = 1.0 / T_                        # uniform
    w[7] = 0.0                                                               # nothing hit
    b[8] = 0.5                                                               # the whole ray collapsed to one point
    w[9] = 0.0; w[9, 0] = 0.5; w[9, -1] = 0.5                                # two far-apart spikes: the pairwise term at its largest
    return b.astype(np.float32), w.astype(np.float32)


@pytest.mark.parametrize("T_", [32, 64, 128])
def test_distort_loss_degenerate_intervals_vs_fp64_brute_force(gpu, T_):
    """The pin of sn_rm_distort_loss (verdict round 5, item 8): the package the reference calls (torch_efficient_distloss, requirements.txt:21,
    renderer.py:14,25) is not installable here (no network), so the kernel is held to the PUBLISHED definition evaluated by brute force in
    fp64 -- sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 d_i, mean over rays -- on random AND degenerate rays, value per ray and gradient per
    element; the torch cumulative-sum statement used for T > 2048 is held to the same."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.nerf import renderer as R
    b, w = _distort_cases(T_, 100 + T_)
    N = b.shape[0]
    w1 = T(w, gpu).requires_grad_(True)
    l1 = rm.distort_loss(T(b, gpu), w1)
    l1.backward()
    bd, wd = T(b, gpu).double(), T(w, gpu).double().requires_grad_(True)
    d = bd[:, 1:] - bd[:, :-1]
    m = bd[:, :-1] + d / 2
    per_ray = (wd[:, :, None] * wd[:, None, :] * (m[:, :, None] - m[:, None, :]).abs()).sum((1, 2)) + (wd * wd * d).sum(1) / 3
    per_ray.mean().backward()
    assert abs(float(l1) - float(per_ray.mean())) <= 2e-6 * abs(float(per_ray.mean())) + 1e-10
    gref = wd.grad
    assert float((w1.grad.double() - gref).abs().max()) <= 2e-6 * float(gref.abs().max()) + 1e-12
    assert float(w1.grad[7].abs().max()) <= 2.0 / N and float(gref[8].abs().max()) == 0.0 and float(w1.grad[8].abs().max()) == 0.0
    w3 = torch.from_numpy(w).double().requires_grad_(True)
    l3 = R.distort_loss(torch.from_numpy(b).double(), w3)                     # CPU tensors: the cumulative-sum statement
    l3.backward()
    assert abs(float(l3) - float(per_ray.mean())) <= 1e-9 * abs(float(per_ray.mean())) + 1e-13
    assert float((w3.grad - gref.cpu()).abs().max()) <= 1e-9 * float(gref.abs().max()) + 1e-14


def test_weights_and_composite(gpu, orc):
    from sanerf_hq_amd import raymarching as rm
    rng = np.random.default_rng(70)
    rb = np.sort(rng.uniform(0.2, 50, (3000, 33)), axis=1).astype(np.float32)
    sg = np.exp(rng.uniform(-5, 5, (3000, 32))).astype(np.float32)
    for opaque in (True, False):
        w = rm.weights_from_sigma(T(rb, gpu), T(sg, gpu), opaque)
        assert np.array_equal(w.cpu().numpy(), orc.weights_from_sigma(rb, sg, opaque)), "shared exp recipe => bit-identical"
    v = rng.standard_normal((3000, 32, 7)).astype(np.float32)
    wt = w.clone().requires_grad_(True)
    vt = T(v, gpu).requires_grad_(True)
    out = rm.composite(wt, vt)
    ref = (w.unsqueeze(-1).double() * T(v, gpu).double()).sum(1)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    go = torch.randn_like(out)
    out.backward(go)
    np.testing.assert_allclose(vt.grad.cpu().numpy(), (w.unsqueeze(-1) * go.unsqueeze(1)).cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), (T(v, gpu) * go.unsqueeze(1)).sum(-1).cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert rm.composite(w, T(v[..., 0], gpu)).shape == (3000,)


@pytest.mark.parametrize("cfg,bound,tile_w,f16", [
    (GRID_CASES[1], 2.0, 0, False),      # SAM grid, the C3 head
    (GRID_CASES[1], 2.0, 40, True),      # image tiling (23 rows: partial 16x16 tiles), fp16 table
    (GRID_CASES[0], 1.5, 0, False),      # bound not a power of two -> IEEE division
    (GRID_CASES[4], 1.0, 24, False),     # tiled grid, generic modulo
    (GRID_CASES[5], 1.0, 0, False),      # align_corners + smoothstep
])
def test_grid_composite_matches_oracle(gpu, orc, cfg, bound, tile_w, f16):
    """sn_rm_grid_composite == composite(weights, grid(xyzs, bound)) (renderer.py:301-302 + 361)."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.gridencoder import GridEncoder
    rng = np.random.default_rng(91)
    N = tile_w * 23 if tile_w else 777
    Tn = 9
    offs, pls = orc.grid_layout(3, cfg["L"], cfg["C"], 2, 16, cfg["log2T"], cfg["desired"])
    emb = rng.uniform(-1, 1, (int(offs[-1]), cfg["C"])).astype(np.float32)
    if f16:
        emb = emb.astype(np.float16)
    xyz = rng.uniform(-1.08 * bound, 1.08 * bound, (N, Tn, 3)).astype(np.float32)     # a few samples out of range
    w = rng.uniform(0, 1, (N, Tn)).astype(np.float32)
    x01 = ((xyz.reshape(-1, 3) + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
    feat, _ = orc.grid_encode_forward(x01, emb, offs, pls, 16, False, cfg["gridtype"], cfg["ac"], cfg["interp"])
    want = (w[..., None].astype(np.float64) * feat.reshape(N, Tn, -1).astype(np.float64)).sum(1)
    enc = GridEncoder(3, cfg["L"], cfg["C"], 2, 16, cfg["log2T"], cfg["desired"],
                      gridtype="hash" if cfg["gridtype"] == 0 else "tiled", align_corners=cfg["ac"],
                      interpolation="linear" if cfg["interp"] == 0 else "smoothstep").to(gpu)
    got = rm.grid_composite(T(w, gpu), T(xyz, gpu), enc, bound, tile_w=tile_w, table=T(emb, gpu))
    assert got.shape == (N, cfg["L"] * cfg["C"]) and got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=2e-6)
    assert rm.grid_composite(T(w[:0], gpu), T(xyz[:0], gpu), enc, bound, table=T(emb, gpu)).shape == (0, cfg["L"] * cfg["C"])


@pytest.mark.parametrize("kind,N", [("samvit", 1000), ("samvit", 1), ("mask", 777), ("mask", 128 * 3)])
def test_wide_mlp_matches_oracle_and_torch(gpu, orc, kind, N):
    """sn_mlp_wide_forward (matrix cores, fp16 hi/lo split products) vs the oracle's fp32 fmaf chains and vs the
    torch module on the GPU: SkipConnMLP + LayerNorm of the SAM head (network.py:101-116) and the mask head's
    143-256-256-n_inst MLP (network.py:118-123).  fp32 contract: 1e-4 (north_star); measured ~1e-6."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(0)
    if kind == "samvit":
        mlp = SkipConnMLP(163, 256, 256, 5, skip_layers=[2], bias=True).to(gpu)
        ln = torch.nn.LayerNorm(256).to(gpu)
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.2, 0.2)
    else:
        mlp = SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False).to(gpu)
        ln = None
    with torch.no_grad():
        for i, lin in enumerate(mlp.net):        # deterministic, well-scaled weights (activations stay O(1))
            lin.weight.copy_(T(synth.linear_weight(lin.weight.shape[0], lin.weight.shape[1], 300 + i, 2.0), gpu))
            if lin.bias is not None:
                lin.bias.copy_(T(synth.hash_uniform((lin.bias.shape[0],), 400 + i, -0.1, 0.1), gpu))
    x = T(np.random.default_rng(3).standard_normal((N, mlp.dim_in)).astype(np.float32), gpu)
    got = rm.mlp_forward(x, mlp, ln)
    with torch.no_grad():
        ref_t = mlp(x)
        if ln is not None:
            ref_t = ln(ref_t)
    ws = [l.weight.detach().cpu().numpy() for l in mlp.net]
    bs = [l.bias.detach().cpu().numpy() if l.bias is not None else None for l in mlp.net]
    om = orc.make_mlp(ws, bs if kind == "samvit" else None, "leaky", mlp.skip_layers)
    ref_o = orc.mlp_forward(om, x.cpu().numpy())
    if ln is not None:
        mu = ref_o.astype(np.float64).mean(-1, keepdims=True)
        var = ((ref_o - mu) ** 2).mean(-1, keepdims=True)
        ref_o = ((ref_o - mu) / np.sqrt(var + ln.eps) * ln.weight.detach().cpu().numpy() + ln.bias.detach().cpu().numpy()).astype(np.float32)
    assert got.shape == ref_t.shape
    scale = float(np.abs(ref_o).max())
    assert np.abs(got.cpu().numpy() - ref_o).max() <= 2e-5 * max(scale, 1.0), "vs oracle"
    assert float((got - ref_t).abs().max()) <= 1e-4 * max(scale, 1.0), "vs torch / rocBLAS"


@pytest.mark.parametrize("N,T_,n_inst,bound", [(300, 32, 2, 2.0), (37, 128, 2, 2.0), (501, 8, 5, 2.0), (130, 64, 3, 1.5), (1000, 1, 2, 2.0), (64, 16, 32, 2.0)])
def test_fused_mask_head_equals_its_unfused_composition(gpu, orc, N, T_, n_inst, bound):
    """sn_rm_mask_head (renderer.py:304-305, 376-385 in one kernel: every lane interpolates one m_grid level of its own
    sample into the first layer's B operand, logits composited in the epilogue) against the three-kernel composition it
    replaces -- m_grid.forward_cat -> mlp_forward -> composite, each pinned to the oracle elsewhere -- and against the
    oracle's fp32 MLP on those features.  Same feature arithmetic and MLP kernel; only the summation tree of the
    compositing differs."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=15, desired_resolution=512).to(gpu)
    with torch.no_grad():
        enc.embeddings.copy_(T(synth.hash_uniform(tuple(enc.embeddings.shape), 77, -1.0, 1.0), gpu))
    mlp = SkipConnMLP(143, n_inst, 256, 3, skip_layers=[], bias=False).to(gpu)
    with torch.no_grad():
        for i, lin in enumerate(mlp.net):
            lin.weight.copy_(T(synth.linear_weight(lin.weight.shape[0], lin.weight.shape[1], 500 + i, 2.0), gpu))
    rng = np.random.default_rng(N + T_)
    xyz = T(rng.uniform(-bound, bound, (N, T_, 3)).astype(np.float32), gpu)
    xyz[0, 0, 0] = bound * 1.5                                        # outside the grid: zero features (gridencoder.cu:105-130)
    geo = T(rng.standard_normal((N, T_, 15)).astype(np.float32), gpu)
    w = T(rng.uniform(0, 0.2, (N, T_)).astype(np.float32), gpu)
    assert rm.mask_head_fusable(enc, mlp, T_, 15)
    got = rm.mask_head(w, xyz, geo, enc, mlp, bound)
    with torch.no_grad():
        mlp_in = enc.forward_cat(xyz, geo, bound=bound)
        logits = rm.mlp_forward(mlp_in.reshape(-1, 143), mlp).reshape(N, T_, n_inst)
        ref = rm.composite(w, logits)
    assert got.shape == (N, n_inst)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= 2e-6 * scale * max(1, T_ // 8)
    om = orc.make_mlp([l.weight.detach().cpu().numpy() for l in mlp.net], None, "leaky", [])
    ref_o = (orc.mlp_forward(om, mlp_in.reshape(-1, 143).cpu().numpy()).reshape(N, T_, n_inst) * w.cpu().numpy()[..., None]).sum(1)
    assert np.abs(got.cpu().numpy() - ref_o).max() <= 1e-4 * scale
    again = rm.mask_head(w, xyz, geo, enc, mlp, bound)
    assert torch.equal(got, again)                                    # fixed reduction tree: deterministic


def test_fused_mask_head_rejects_what_it_cannot_do(gpu):
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    enc8 = GridEncoder(input_dim=3, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=12, desired_resolution=64).to(gpu)
    enc2 = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=12, desired_resolution=64).to(gpu)
    mlp = SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False).to(gpu)
    assert not rm.mask_head_fusable(enc8, mlp, 33, 15)                # samples per ray not a power of two
    assert not rm.mask_head_fusable(enc2, SkipConnMLP(47, 2, 256, 3, skip_layers=[], bias=False).to(gpu), 32, 15)
    x, g, w = torch.zeros(4, 33, 3, device=gpu), torch.zeros(4, 33, 15, device=gpu), torch.zeros(4, 33, device=gpu)
    with pytest.raises(RuntimeError, match="power of two"):
        rm.mask_head(w, x, g, enc8, mlp, 2.0)


def test_wide_mlp_rejects_unsupported_shapes(gpu):
    """sn_mlp_wide_forward states its limits through the error channel instead of computing something else."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.nerf.network import MLP, SkipConnMLP
    x = torch.randn(10, 32, device=gpu)
    with pytest.raises(RuntimeError, match="hidden width must be 256"):
        rm.mlp_forward(x, SkipConnMLP(32, 4, 64, 3).to(gpu))
    with pytest.raises(RuntimeError, match="output width"):
        rm.mlp_forward(x, SkipConnMLP(32, 300, 256, 2).to(gpu))
    y = rm.mlp_forward(x, MLP(32, 5, 256, 2, bias=False).to(gpu))          # ReLU perceptron (network.py:9-29) also maps onto it
    assert y.shape == (10, 5)
    ref = MLP(32, 5, 256, 2, bias=False)
    torch.manual_seed(1); m = MLP(32, 5, 256, 2, bias=False).to(gpu)
    with torch.no_grad():
        np.testing.assert_allclose(rm.mlp_forward(x, m).cpu().numpy(), m(x).cpu().numpy(), rtol=0, atol=2e-5)


def test_wide_mlp_large_batch_is_race_free(gpu):
    """C3-sized batch (160 000 rows = 1250 tiles) of the SAM head MLP, repeated: the input tile reaches LDS by an
    asynchronous DMA that the first k-step must wait for (a missing wait once showed up as two wrong rows -- a tile's
    last -- in roughly one run out of five).  Every repetition must be bit-equal and close to the torch module."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(0)
    mlp = SkipConnMLP(163, 256, 256, 5, skip_layers=[2], bias=True).to(gpu)
    ln = torch.nn.LayerNorm(256).to(gpu)
    x = torch.randn(160000, 163, device=gpu)
    with torch.no_grad():
        ref = ln(mlp(x))
    first = rm.mlp_forward(x, mlp, ln)
    assert float((first - ref).abs().max()) < 5e-5
    for _ in range(25):
        assert torch.equal(rm.mlp_forward(x, mlp, ln), first)


@pytest.mark.parametrize("M,N,K,bias,act", [(1000, 64, 32, False, 0), (4099, 128, 163, True, 1), (65, 1, 10, False, 0), (1, 300, 70, True, 2), (20000, 16, 10, False, 1),
                                            (333, 257, 129, True, 0), (64, 64, 0, True, 0)])
def test_general_fp32_matrix_product(gpu, M, N, K, bias, act):
    """sn_gemm_f32 (csrc/linear.hip: every nn.Linear shape the fused kernels do not cover, true fp32 on the matrix cores) in its three uses --
    layer forward with bias and activation, input gradient, weight gradient -- against fp64 torch; exact-fp32 means one k-ascending fmaf chain
    per output: the forward equals that chain evaluated in numpy bit for bit on a slice; deterministic; ragged edges of all three dimensions."""
    from sanerf_hq_amd import ops
    rng = np.random.default_rng(M + 7 * N + 13 * K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / max(K, 1) ** 0.5).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    xt, wt = T(x, gpu), T(w, gpu)
    bt = T(b, gpu) if bias else None
    y = ops.linear_forward(xt, wt, bt, act)
    ref = xt.double() @ wt.double().t() + (bt.double() if bias else 0.0)
    ref = torch.relu(ref) if act == 1 else (torch.nn.functional.leaky_relu(ref, 0.01) if act == 2 else ref)
    assert y.shape == (M, N)
    assert float((y - ref.float()).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    assert torch.equal(y, ops.linear_forward(xt, wt, bt, act)), "deterministic"
    # the chain itself, bit for bit: acc = fma(x[k], w[k], acc) for k ascending, then + bias, then the activation (fp32 throughout)
    rows, cols = min(M, 5), min(N, 7)
    want = np.zeros((rows, cols), np.float32)
    for i in range(rows):
        for j in range(cols):
            acc = np.float32(0.0)
            for k in range(K):
                acc = np.float32(np.float64(x[i, k]) * np.float64(w[j, k]) + np.float64(acc))      # fp64 product of two fp32 is exact, one rounding: an fma
            v = np.float32(acc + (b[j] if bias else np.float32(0.0)))
            want[i, j] = max(v, np.float32(0.0)) if act == 1 else (max(v, np.float32(v * np.float32(0.01))) if act == 2 else v)
    assert np.array_equal(y[:rows, :cols].cpu().numpy(), want)
    if K == 0:
        return
    gy = rng.standard_normal((M, N)).astype(np.float32)
    gyt = T(gy, gpu)
    gx = ops.linear_backward_input(gyt, wt)
    refx = gyt.double() @ wt.double()
    assert float((gx - refx.float()).abs().max()) <= 2e-6 * max(1.0, float(refx.abs().max()))
    gw = torch.empty(N, K, device=gpu)
    ops.linear_wgrad_general(xt, gyt, gw)
    refw = gyt.double().t() @ xt.double()
    assert float((gw - refw.float()).abs().max()) <= 1e-5 * max(1.0, float(refw.abs().max()))     # (an M-long fp32 chain)


def test_general_fp32_matrix_product_random_shapes_and_layouts(gpu):
    """sn_gemm_f32 on 40 random shapes (1..300 per dimension, ragged against the 64 x 64 x 32 tiling) in all four operand layouts (each operand row- or
    column-major), with and without bias, against fp64 torch."""
    from sanerf_hq_amd import ops
    rng = np.random.default_rng(11)
    for trial in range(40):
        M, N, K = (int(rng.integers(1, 301)) for _ in range(3))
        ta, tb, bias = bool(trial & 1), bool(trial & 2), bool(trial & 4)
        a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
        b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
        bv = rng.standard_normal(N).astype(np.float32) if bias else None
        at, bt = T(a, gpu), T(b, gpu)
        out = torch.empty(M, N, device=gpu)
        ops._gemm(at, 1 if ta else K, M if ta else 1, bt, 1 if tb else N, K if tb else 1, T(bv, gpu) if bias else None, ops.ACT_NONE, M, N, K, out)
        ref = (at.double().t() if ta else at.double()) @ (bt.double().t() if tb else bt.double()) + (T(bv, gpu).double() if bias else 0.0)
        assert float((out - ref.float()).abs().max()) <= 3e-6 * max(1.0, float(ref.abs().max())), (M, N, K, ta, tb, bias)


def test_layers_of_other_widths_run_the_library_product_under_autograd(gpu):
    """ops.small_linear on shapes no fused kernel covers (a 40-128-128-3 head with biases, a 300-wide layer): forward, input gradient, weight and
    bias gradients through sn_gemm_f32 / sn_linear_wgrad equal the torch layer's (rocBLAS) within fp32 round-off -- with and without autograd,
    for few rows and many; COLD_GEMM_NATIVE = False gives the torch route back."""
    from sanerf_hq_amd import ops
    torch.manual_seed(3)
    for rows, widths in ((50, (40, 128, 3)), (20000, (40, 128, 128, 3)), (1000, (17, 300, 5))):
        layers = [torch.nn.Linear(a, b, bias=True).to(gpu) for a, b in zip(widths[:-1], widths[1:])]
        x = torch.randn(rows, widths[0], device=gpu, requires_grad=True)

        def run(native):
            ops.COLD_GEMM_NATIVE = native
            try:
                for l in layers:
                    l.zero_grad()
                x.grad = None
                h = x
                for l in layers[:-1]:
                    h = torch.relu(ops.small_linear(h, l))
                y = ops.small_linear(h, layers[-1])
                (y * torch.linspace(0.5, 1.5, y.numel(), device=gpu).reshape(y.shape)).sum().backward()
                return [y.detach().clone(), x.grad.clone()] + [l.weight.grad.clone() for l in layers] + [l.bias.grad.clone() for l in layers]
            finally:
                ops.COLD_GEMM_NATIVE = True
        a, b = run(True), run(False)
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max())), (rows, widths)
        with torch.no_grad():
            y0 = ops.small_linear(x, layers[0])
        assert float((y0 - torch.nn.functional.linear(x, layers[0].weight, layers[0].bias)).abs().max()) <= 1e-5


@pytest.mark.parametrize("M,K,N", [(131072, 32, 64), (5000, 10, 16), (70001, 16, 1), (4096, 31, 32), (1, 64, 64), (524288, 10, 16), (262147, 16, 1), (130, 10, 16), (63, 16, 1)])
def test_linear_wgrad_matches_matmul(gpu, M, K, N):
    """sn_linear_wgrad: dw = dy^T x for the <= 64-wide layers of the radiance / proposal MLPs."""
    import ctypes as C
    from sanerf_hq_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(gpu)
    dy = torch.randn(M, N, generator=g).to(gpu)
    lib = _lib.lib()
    need = int(lib.sn_linear_wgrad_workspace_bytes(M, K, N))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=gpu)
    dw = torch.empty(N, K, device=gpu)
    _lib.check(lib.sn_linear_wgrad(x.data_ptr(), dy.data_ptr(), M, K, N, dw.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "wgrad")
    ref = (dy.double().t() @ x.double()).float()
    scale = float(ref.abs().max()) + 1e-6
    assert float((dw - ref).abs().max()) <= 2e-5 * scale * max(1.0, (M / 1e4) ** 0.5)
    dw2 = torch.empty_like(dw)
    _lib.check(lib.sn_linear_wgrad(x.data_ptr(), dy.data_ptr(), M, K, N, dw2.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "wgrad")
    assert torch.equal(dw, dw2), "fixed summation order: bit-reproducible"
    assert lib.sn_linear_wgrad_workspace_bytes(10, 65, 257) == 0, "more than 256 outputs: not this kernel's shape"


@pytest.mark.parametrize("M,K,N", [(131072, 256, 256), (131072, 143, 256), (131072, 256, 2), (20001, 419, 256), (16385, 77, 100),
                                   (4097, 256, 33), (70000, 128, 128), (3, 200, 256), (50000, 66, 30)])
def test_linear_wgrad_wide_layers(gpu, M, K, N):
    """The matrix-core path of sn_linear_wgrad (mask / SAM head MLP shapes, network.py:31-66): fp32 products, slabs summed
    in a fixed order.  Shapes cover the vector and scalar operand loads, the 2-output layer, column groups beyond 256,
    ragged row counts and fewer rows than one slab."""
    from sanerf_hq_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(gpu)
    dy = torch.randn(M, N, generator=g).to(gpu)
    lib = _lib.lib()
    need = int(lib.sn_linear_wgrad_workspace_bytes(M, K, N))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=gpu)
    dw = torch.full((N, K), float("nan"), device=gpu)
    _lib.check(lib.sn_linear_wgrad(x.data_ptr(), dy.data_ptr(), M, K, N, dw.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "wgrad")
    ref = (dy.double().t() @ x.double()).float()
    scale = float(ref.abs().max()) + 1e-6
    assert float((dw - ref).abs().max()) <= 2e-5 * scale * max(1.0, (M / 1e4) ** 0.5)
    dw2 = torch.empty_like(dw)
    _lib.check(lib.sn_linear_wgrad(x.data_ptr(), dy.data_ptr(), M, K, N, dw2.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "wgrad")
    assert torch.equal(dw, dw2), "fixed summation order: bit-reproducible"
    # unaligned operands take the scalar-load instantiation: same values
    xo = torch.empty(M * K + 1, device=gpu)[1:].view(M, K).copy_(x)
    _lib.check(lib.sn_linear_wgrad(xo.data_ptr(), dy.data_ptr(), M, K, N, dw2.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream()), "wgrad")
    assert float((dw2 - ref).abs().max()) <= 2e-5 * scale * max(1.0, (M / 1e4) ** 0.5)


def test_small_linear_autograd_equals_nn_linear(gpu):
    from sanerf_hq_amd.ops import small_linear
    torch.manual_seed(0)
    lin = torch.nn.Linear(32, 64, bias=True).to(gpu)
    x = torch.randn(40000, 32, device=gpu, requires_grad=True)
    y1 = small_linear(x, lin); (y1 * y1).sum().backward()
    g1 = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None; lin.zero_grad()
    y2 = lin(x); (y2 * y2).sum().backward()
    assert float((y1 - y2).abs().max()) <= 2e-6 * float(y2.abs().max())        # (the library's own fp32 product against rocBLAS's: summation order)
    for a, b in zip(g1, (x.grad, lin.weight.grad, lin.bias.grad)):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    wide = torch.nn.Linear(143, 256, bias=False).to(gpu)       # mask head, first layer (network.py:180)
    xw = torch.randn(40000, 143, device=gpu)
    small_linear(xw, wide).square().sum().backward()
    gw = wide.weight.grad.clone(); wide.zero_grad()
    wide(xw).square().sum().backward()
    assert float((gw - wide.weight.grad).abs().max()) <= 1e-4 * float(wide.weight.grad.abs().max())


def test_device_exp_matches_the_oracle_bit_for_bit(gpu, orc):
    """The deterministic exp shared by csrc/ and oracle/ (numerics contract): bit-identical over the
    whole argument range incl. the overflow / underflow ends, subnormal results, infinities and NaN."""
    import ctypes as C
    from sanerf_hq_amd import _lib
    rng = np.random.default_rng(5)
    x = np.concatenate([
        rng.uniform(-110.0, 95.0, 400000), rng.uniform(-1.0, 1.0, 100000), rng.uniform(-104.5, -86.0, 200000),      # subnormal results
        rng.uniform(88.0, 89.5, 50000), np.linspace(-104.2, -103.8, 20001), np.linspace(88.70, 88.75, 20001),
        [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, -1e-30, 88.72283935546875, -103.97208404541015625, 89.0, -104.0, 1e30, -1e30],
    ]).astype(np.float32)
    xt = torch.from_numpy(x).to(gpu)
    yt = torch.empty_like(xt)
    _lib.check(_lib.lib().sn_debug_eval(0, xt.data_ptr(), None, x.size, yt.data_ptr(), _lib.stream()), "debug_eval")
    got = yt.cpu().numpy()
    want = orc.expf(x)
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32)), \
        f"{int((got.view(np.uint32) != want.astype(np.float32).view(np.uint32)).sum())} of {x.size} values differ"
    ok = np.isfinite(want) & (want > 1e-37)
    ref = np.exp(x[ok].astype(np.float64))
    assert np.max(np.abs(got[ok] - ref) / ref) < 2.5e-7            # <= ~2 ulp of the correctly rounded value
    assert got[x == np.float32(-np.inf)][0] == 0.0 and np.isinf(got[x == np.float32(np.inf)][0]) and np.isnan(got[np.isnan(x)]).all()


def test_get_rays_one_camera_per_ray_matches_the_reference(gpu):
    """provider.py:908-913 (`random_image_batch`): one image index per ray -> poses [N,4,4], intrinsics [N,4].  Fixture =
    the reference's own get_rays output (tools/gen_rays_fixture.py)."""
    from helpers import golden
    from sanerf_hq_amd.nerf import get_rays
    from sanerf_hq_amd import raymarching as rm
    g = golden("rays_multi")
    H, W = [int(v) for v in g["HW"]]
    cams, intr = torch.from_numpy(g["cams"]).to(gpu), torch.from_numpy(g["intr"]).to(gpu)
    index = torch.from_numpy(g["index"]).to(gpu)
    coords = torch.from_numpy(g["coords"]).to(gpu)
    res = get_rays(cams[index], intr[index], H, W, coords.shape[0], coords=coords, incoherent_mask_size=32)
    assert np.array_equal(res["rays_o"].cpu().numpy(), g["rays_o"])
    np.testing.assert_allclose(res["rays_d"].cpu().numpy(), g["rays_d"], rtol=0, atol=3e-7)
    assert np.array_equal(res["i"].cpu().numpy(), g["i"]) and np.array_equal(res["j"].cpu().numpy(), g["j"])
    assert np.array_equal(res["inds_coarse"].cpu().numpy(), g["inds_coarse"])
    # every ray equals the ray of the same pixel in its own camera's full image (bit for bit: same kernel arithmetic)
    for k in range(cams.shape[0]):
        sel = (index == k).nonzero().reshape(-1)
        if sel.numel() == 0:
            continue
        ro, rd = rm.generate_rays(cams[k], intr[k].tolist(), H, W, device=gpu)
        flat = coords[sel, 0] * W + coords[sel, 1]
        assert torch.equal(res["rays_d"][sel], rd[flat]) and torch.equal(res["rays_o"][sel], ro[flat])
    # uniform draw with per-ray cameras, the training call of provider.py:972-977
    draw = get_rays(cams[index], intr[index], H, W, index.numel(), random_sample=True)
    assert draw["rays_d"].shape == (index.numel(), 3) and int(draw["i"].max()) < W and int(draw["j"].max()) < H
    with pytest.raises(RuntimeError, match="poses"):
        get_rays(cams[:3], intr[:1], H, W, 10, random_sample=True)


@pytest.mark.parametrize("n,wd", [(4096 * 33 + 3, 0.0), (1 << 20, 0.0), (1000, 1e-3)])
def test_single_pass_adam_matches_torch_adam(gpu, n, wd):
    """sanerf_hq_amd.optim.Adam (sn_adam_step: one kernel per tensor) against torch.optim.Adam -- the reference's optimiser,
    main.py:283, eps=1e-15 -- over 6 steps with a sparse gradient pattern (most rows untouched on most steps, like a hash
    table): parameters, both moments and the state_dict layout agree; rows never touched stay bit-identical."""
    from sanerf_hq_amd.optim import Adam
    rng = np.random.default_rng(n)
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(gpu)
    pa, pb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa = Adam([dict(params=[pa], lr=1e-2)], eps=1e-15, weight_decay=wd)
    ob = torch.optim.Adam([dict(params=[pb], lr=1e-2)], eps=1e-15, weight_decay=wd, foreach=False)
    never = torch.ones(n, dtype=torch.bool, device=gpu)
    for step in range(6):
        g = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(gpu)
        mask = torch.from_numpy(rng.uniform(size=n) < 0.2).to(gpu)
        mask[: n // 3] = False                                      # the first third never receives a gradient
        g = g * mask
        never &= ~mask
        pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step()
        scale = float(pb.abs().max())
        assert float((pa - pb).abs().max()) <= 2e-6 * scale, step
    sa, sb = oa.state[pa], ob.state[pb]
    assert float(sa["step"]) == float(sb["step"]) == 6
    for k in ("exp_avg", "exp_avg_sq"):
        assert float((sa[k] - sb[k]).abs().max()) <= 1e-6 * max(float(sb[k].abs().max()), 1e-30)
    if wd == 0.0:
        assert torch.equal(pa.detach()[never], p0[never])           # exactly-zero updates are skipped, exactly
        assert never.any()
    assert set(oa.state_dict()["state"][0]) == set(ob.state_dict()["state"][0])
    ob2 = torch.optim.Adam([dict(params=[pb], lr=1e-2)], eps=1e-15, weight_decay=wd)
    ob2.load_state_dict(oa.state_dict())                            # a checkpoint written with one loads into the other


def test_collate_rays_draws_cameras_pixels_and_supervision_on_the_device(gpu):
    """the device-side core of NeRFDataset.collate (provider.py:894-1114).  Every returned ray must be the
    ray of pixel (j, i) of camera `index` (checked against full-image rays of that camera), and every gathered
    supervision value the dataset tensor's entry at exactly that (camera, pixel) -- in the one-camera-per-ray mode
    (random_image_batch), the single-image error-map mode, and with the mixed local patches appended."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import collate_rays
    torch.manual_seed(0)
    M, H, W, S = 5, 48, 64, 16
    poses = torch.stack([torch.from_numpy(synth.orbit_pose(1.0 + 0.1 * k, 10.0 * k, 40.0 * k)) for k in range(M)]).to(gpu)
    fx0, fy0, cx0, cy0 = synth.pinhole_intrinsics(H, W)
    intr = torch.tensor([[fx0 * (1 + 0.05 * k), fy0 * (1 + 0.03 * k), cx0 + k, cy0 - k] for k in range(M)], device=gpu)   # one camera model per image
    images = torch.randint(0, 256, (M, H, W, 3), dtype=torch.uint8, device=gpu)
    masks = torch.randint(0, 3, (M, H, W, 1), device=gpu)
    emap = torch.rand(M, S * S, device=gpu) + 0.01
    cnf = torch.rand(M, 2, device=gpu)
    full = [rm.generate_rays(poses[k], [float(v) for v in intr[k]], H, W, device=gpu) for k in range(M)]

    def check(res, n_main, n_total):
        idx, i, j = res["index"], res["i"], res["j"]
        cam = idx if idx.numel() == n_main else idx.expand(n_main)
        pix = j * W + i
        for k in range(M):
            sel = cam == k
            if sel.any():
                assert torch.equal(res["rays_o"][:n_main][sel], full[k][0][pix[sel]]) and torch.equal(res["rays_d"][:n_main][sel], full[k][1][pix[sel]])
        assert torch.equal(res["images"], images[cam, j, i].float() / 255)
        assert torch.equal(res["masks"][:n_main], masks[cam, j, i].view(-1, 1))
        assert torch.equal(res["error_maps"][:n_main], emap[cam, (j * (S / H)).long() * S + (i * (S / W)).long()])
        assert torch.equal(res["cam_near_far"][:idx.numel()], cnf[idx])                 # [1, 2] in the single-image mode, like the reference
        assert res["rays_o"].shape == (n_total, 3) and res["masks"].shape[0] == n_total
        assert res["cam_near_far"].shape[0] == idx.numel() + (n_total - n_main)
        # appended local patches: every ray is a ray of ITS image's own camera model (pose and intrinsics), i.e. one of that image's pixels
        for r in range(n_main, n_total):
            k = int(torch.nonzero((poses == res["poses"][r]).flatten(1).all(1))[0])
            hit = (full[k][1] == res["rays_d"][r]).all(-1)
            assert bool(hit.any()) and torch.equal(res["rays_o"][r], full[k][0][0]), (r, k)

    a = collate_rays(poses, intr, H, W, 512, images=images, masks=masks, error_map=emap, cam_near_far=cnf,
                     random_image_batch=True, error_map_size=S)
    assert a["index"].shape == (512,) and len(torch.unique(a["index"])) == M
    check(a, 512, 512)
    b = collate_rays(poses, intr, H, W, 128, index=torch.tensor([3]), images=images, masks=masks, error_map=emap, cam_near_far=cnf,
                     random_image_batch=False, use_error_map=True, error_map_size=S)
    assert b["inds_coarse"].shape == (1, 128) and len(torch.unique(b["inds_coarse"])) == 128      # drawn without replacement
    check(b, 128, 128)
    c = collate_rays(poses, intr, H, W, 256, images=images, masks=masks, error_map=emap, cam_near_far=cnf,
                     random_image_batch=True, error_map_size=S, num_local_sample=3, local_patch_size=4)
    check(c, 256, 256 + 3 * 16)


@pytest.mark.parametrize("N,din,n_out,leaky", [(20000, 143, 2, True), (16384 + 77, 64, 5, True), (33000, 143, 2, False)])
def test_wide_mlp_fused_backward_matches_autograd(gpu, N, din, n_out, leaky):
    """sn_mlp_wide_backward (one kernel: gradient of the input and of every hidden pre-activation on the matrix cores, masks
    from the forward's saved outputs) + sn_linear_wgrad against torch autograd over the same forward (rocBLAS fp32).  The
    upstream gradient rows span 8 orders of magnitude like a training step's (a sample's weight multiplies its row): the
    per-row power-of-two scaling must keep small rows accurate."""
    from sanerf_hq_amd import ops, synth
    ws = [T(synth.linear_weight(256, din, 700, 2.0), gpu), T(synth.linear_weight(256, 256, 701, 2.0), gpu), T(synth.linear_weight(n_out, 256, 702, 2.0), gpu)]
    rng = np.random.default_rng(N)
    x = T(rng.standard_normal((N, din)).astype(np.float32), gpu)
    gy = T((rng.standard_normal((N, n_out)) * 10.0 ** rng.uniform(-9, -1, (N, 1))).astype(np.float32), gpu)
    gy[5] = 0.0                                                           # an all-zero row
    act = (lambda t: torch.nn.functional.leaky_relu(t)) if leaky else torch.relu

    def run(fused):
        xs = x.clone().requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        if fused:
            y = ops._wide_mlp_train.apply(xs, leaky, *wl)
        else:
            y = torch.nn.functional.linear(act(torch.nn.functional.linear(act(torch.nn.functional.linear(xs, wl[0])), wl[1])), wl[2])
        y.backward(gy)
        return y.detach(), xs.grad, [w.grad for w in wl]

    ops.WIDE_MLP_FORWARD_F16X3 = False         # the BLAS forward on both sides: this test is about the backward (a forward that rounds
    try:                                       # differently flips a few LeakyReLU branches, which per-row tolerances would flag)
        assert not ops.WIDE_MLP_FORWARD_NATIVE
        ya, gxa, gwa = run(True)
    finally:
        ops.WIDE_MLP_FORWARD_F16X3 = True
    yb, gxb, gwb = run(False)
    assert torch.equal(ya, yb)                                            # the forward is the same GEMM chain
    # per-row accuracy of the input gradient: relative to each row's own magnitude (small rows are not swamped)
    rn = gxb.norm(dim=1)
    rel = (gxa - gxb).norm(dim=1) / rn.clamp_min(1e-30)
    assert float(rel[rn > 0].max()) < 2e-5, float(rel[rn > 0].max())
    assert float(gxa[5].abs().max()) == 0.0
    for a_, b_ in zip(gwa, gwb):
        assert float((a_ - b_).norm() / b_.norm()) < 1e-5


@pytest.mark.parametrize("N,din,n_out", [(20000, 143, 2), (33000, 64, 5)])
def test_wide_mlp_default_forward_and_sign_bits_backward_per_row(gpu, N, din, n_out):
    """The DEFAULT training route of the wide MLP (split-fp16 x3 forward with saved outputs + backward from sign bits, advisor round 5): per
    row against torch autograd in fp64.  A unit whose fp64 pre-activation is within 4e-6 of zero may take the other LeakyReLU branch under a
    differently rounded forward (pre-activations are O(1-10) here, the forward is good to ~2e-7 relative); rows that contain such a unit are
    excluded from the per-row bound (and counted: they must be rare)."""
    from sanerf_hq_amd import ops, synth
    assert ops.WIDE_MLP_FORWARD_F16X3 and ops.WIDE_MLP_SIGN_BITS and not ops.WIDE_MLP_FORWARD_NATIVE
    ws = [T(synth.linear_weight(256, din, 710, 2.0), gpu), T(synth.linear_weight(256, 256, 711, 2.0), gpu), T(synth.linear_weight(n_out, 256, 712, 2.0), gpu)]
    rng = np.random.default_rng(N + 1)
    x = T(rng.standard_normal((N, din)).astype(np.float32), gpu)
    gy = T((rng.standard_normal((N, n_out)) * 10.0 ** rng.uniform(-6, -1, (N, 1))).astype(np.float32), gpu)
    xs = x.clone().requires_grad_(True)
    wl = [w.clone().requires_grad_(True) for w in ws]
    y = ops._wide_mlp_train.apply(xs, True, *wl)
    y.backward(gy)
    x64 = x.double().requires_grad_(True)
    w64 = [w.double().requires_grad_(True) for w in ws]
    p1 = torch.nn.functional.linear(x64, w64[0])
    p2 = torch.nn.functional.linear(torch.nn.functional.leaky_relu(p1), w64[1])
    y64 = torch.nn.functional.linear(torch.nn.functional.leaky_relu(p2), w64[2])
    y64.backward(gy.double())
    assert float((y.double() - y64).norm() / y64.norm()) < 2e-6
    risky = ((p1.abs() < 4e-6).any(dim=1) | (p2.abs() < 4e-6).any(dim=1))
    assert int(risky.sum()) < max(8, N // 100), int(risky.sum())
    rn = x64.grad.norm(dim=1)
    rel = (xs.grad.double() - x64.grad).norm(dim=1) / rn.clamp_min(1e-30)
    keep = (rn > 0) & ~risky
    assert float(rel[keep].max()) < 5e-5, float(rel[keep].max())
    for a_, b_ in zip([w.grad for w in wl], [w.grad for w in w64]):
        assert float((a_.double() - b_).norm() / b_.norm()) < 1e-4


def test_wide_mlp_training_forward_reports_a_left_fp16_range(gpu):
    """ops.WIDE_MLP_RANGE_CHECK_EVERY (advisor round 5, medium): the split-fp16 training forward has a range contract (|v| < 65504); an input
    beyond it makes the loss NaN.  The periodic check of the library's sticky flag turns that into a RuntimeError that names the cause."""
    from sanerf_hq_amd import ops, raymarching as rm, synth
    ws = [T(synth.linear_weight(256, 143, 720, 2.0), gpu).requires_grad_(True), T(synth.linear_weight(256, 256, 721, 2.0), gpu).requires_grad_(True),
          T(synth.linear_weight(2, 256, 722, 2.0), gpu).requires_grad_(True)]
    x = torch.randn(20000, 143, device=gpu)
    rm.mlp_wide_overflow()                                                # clear the flag
    old = ops.WIDE_MLP_RANGE_CHECK_EVERY
    ops.WIDE_MLP_RANGE_CHECK_EVERY = 1
    try:
        ops._wide_mlp_train.apply(x, True, *ws)                           # in range: no complaint
        # a STALE flag (raised by some other call of the head kernels, e.g. an inference render) must not fail an in-range training forward
        from sanerf_hq_amd.nerf.network import SkipConnMLP
        with torch.no_grad():
            rm.mlp_forward(x * 1e6, SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False).to(gpu), check_range=False)
        ops._wide_mlp_train.apply(x, True, *ws)
        with pytest.raises(RuntimeError, match="fp16 range"):
            ops._wide_mlp_train.apply(x * 1e6, True, *ws)
        ops.WIDE_MLP_FORWARD_F16X3 = False
        y = ops._wide_mlp_train.apply(x * 1e6, True, *ws)                 # the BLAS forward has no such limit
        assert bool(torch.isfinite(y).all())
    finally:
        ops.WIDE_MLP_RANGE_CHECK_EVERY = old
        ops.WIDE_MLP_FORWARD_F16X3 = True
        rm.mlp_wide_overflow()


def test_new_entry_points_accept_empty_batches(gpu):
    """N = 0 through the round-2 entry points: nothing is launched, nothing faults, shapes are right."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    from sanerf_hq_amd.optim import Adam
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=12, desired_resolution=64).to(gpu)
    mlp = SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False).to(gpu)
    out = rm.mask_head(torch.zeros(0, 32, device=gpu), torch.zeros(0, 32, 3, device=gpu), torch.zeros(0, 32, 15, device=gpu), enc, mlp, 2.0)
    assert out.shape == (0, 2)
    p = torch.nn.Parameter(torch.zeros(0, device=gpu))
    p.grad = torch.zeros(0, device=gpu)
    Adam([p], lr=1e-3).step()
    q = torch.nn.Parameter(torch.ones(7, device=gpu))                   # a tail shorter than one 16-byte vector
    q.grad = torch.full((7,), 0.5, device=gpu)
    ref = torch.nn.Parameter(q.detach().clone()); ref.grad = q.grad.clone()
    Adam([q], lr=1e-2, eps=1e-15).step(); torch.optim.Adam([ref], lr=1e-2, eps=1e-15).step()
    assert float((q - ref).abs().max()) < 1e-6
