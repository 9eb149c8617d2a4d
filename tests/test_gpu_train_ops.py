"""GPU tests of the fused training operators of the RGB-mode step (round 6): the small perceptrons on the matrix cores
(csrc/mlp_small.hip), the per-ray head, the jitter kernel, unit-cube sample positions, the one-node proposal loss, and the renderer's
training route built from them against the operator chain it replaces.  References are plain torch fp32 / fp64 on the same inputs; the
reference's own autograd pins the whole step in test_gpu_render.py::test_rgb_training_step_vs_reference_fixture (tests/golden/train_rgb.npz)."""
import numpy as np
import pytest
import torch

from helpers import make_opt, synthetic_params

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


SHAPES = [(10, 16, 1), (32, 64, 64, 16), (31, 32, 32, 3), (16, 32, 16), (31, 32, 3)]


def _layers(dims, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    layers = [torch.nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])]
    for l in layers:
        l.weight.data = (torch.rand(l.weight.shape, generator=g) * 2 - 1) * (1.5 / np.sqrt(l.weight.shape[1]))
    return [l.to(dev) for l in layers]


def _torch_mlp(x, layers):
    h = x
    for l in layers[:-1]:
        h = torch.relu(torch.nn.functional.linear(h, l.weight))
    return torch.nn.functional.linear(h, layers[-1].weight)


@pytest.mark.parametrize("dims", SHAPES)
@pytest.mark.parametrize("rows", [1, 63, 64, 1000, 70001])
def test_small_mlp_forward_backward_vs_torch(gpu, dims, rows):
    """sn_mlp_small_forward_train / sn_mlp_small_backward + sn_linear_wgrad against torch autograd in fp64 for every instantiated shape;
    rows that are not multiples of the 64-row wave tile; tolerance: fp32 round-off (1e-5 relative L2 per tensor; 1e-4 for the large batch, where
    one ReLU unit in ~1e7 has a pre-activation within fp32 round-off of 0 and takes the other branch than the fp64 reference)."""
    from sanerf_hq_amd import ops
    assert ops.SMALL_MLP_FUSED
    layers = _layers(dims, gpu, 7 + len(dims))
    g = torch.Generator(device=gpu).manual_seed(rows)
    x = (torch.rand(rows, dims[0], device=gpu, generator=g) * 2 - 1).requires_grad_(True)
    gy = torch.randn(rows, dims[-1], device=gpu, generator=g)
    assert ops.small_mlp_fusable(x, layers)
    out, aux = ops.small_mlp_train(x, layers)
    assert aux is None
    out.backward(gy)
    got = [x.grad.clone()] + [l.weight.grad.clone() for l in layers]
    x64 = x.detach().double().requires_grad_(True)
    l64 = [torch.nn.Linear(a, b, bias=False).to(gpu).double() for a, b in zip(dims[:-1], dims[1:])]
    for a, b in zip(l64, layers):
        a.weight.data = b.weight.data.double()
    ref = _torch_mlp(x64, l64)
    ref.backward(gy.double())
    assert rel(out, ref) < 2e-6
    want = [x64.grad] + [l.weight.grad for l in l64]
    for i, (a, b) in enumerate(zip(got, want)):
        assert rel(a, b) < (1e-5 if rows <= 1000 else 1e-4), (i, rel(a, b))


@pytest.mark.parametrize("dims", [(10, 16, 1), (32, 64, 64, 16)])
def test_small_mlp_trunc_exp_output(gpu, dims):
    """act = TRUNC_EXP0: sigma = exp(raw[:, 0]) forward, g * exp(clamp(raw, -15, 15)) backward (activation.py:5-17), gradients arriving through
    BOTH outputs (the geometry channels and the density) -- against torch's trunc_exp on the torch MLP."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.activation import trunc_exp
    rows = 5000
    layers = _layers(dims, gpu, 3)
    g = torch.Generator(device=gpu).manual_seed(1)
    x = (torch.rand(rows, dims[0], device=gpu, generator=g) * 4 - 2).requires_grad_(True)
    gs = torch.randn(rows, device=gpu, generator=g)
    gr = torch.randn(rows, dims[-1], device=gpu, generator=g)
    raw, sig = ops.small_mlp_train(x, layers, ops.SMALL_ACT_TRUNC_EXP0)
    ((raw * gr).sum() + (sig * gs).sum()).backward()
    got = [x.grad.clone()] + [l.weight.grad.clone() for l in layers]
    x.grad = None
    for l in layers:
        l.weight.grad = None
    old = ops.SMALL_MLP_FUSED
    ops.SMALL_MLP_FUSED = False
    try:
        raw2 = _torch_mlp(x, layers)
        sig2 = trunc_exp(raw2[:, 0])
        ((raw2 * gr).sum() + (sig2 * gs).sum()).backward()
    finally:
        ops.SMALL_MLP_FUSED = old
    assert rel(raw, raw2) < 2e-6 and rel(sig, sig2) < 5e-6
    for a, b in zip(got, [x.grad] + [l.weight.grad for l in layers]):
        assert rel(a, b) < 2e-5, rel(a, b)
    # only the density carries a gradient (proposal stages): the raw output's gradient is absent, not a zero tensor
    x.grad = None
    raw, sig = ops.small_mlp_train(x, layers, ops.SMALL_ACT_TRUNC_EXP0)
    (sig * gs).sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    (trunc_exp(_torch_mlp(x, layers)[:, 0]) * gs).sum().backward()
    assert rel(g1, x.grad) < 2e-5


def test_small_mlp_sigmoid_background_output(gpu):
    """act = SIGMOID_BG on view_mlp's shape: image = sigmoid(raw) + (1 - weights_sum) * bg (renderer.py:349-353), gradient to the input, the
    weights and weights_sum."""
    from sanerf_hq_amd import ops
    dims, rows, bg = (31, 32, 32, 3), 4096, 1.0
    layers = _layers(dims, gpu, 5)
    g = torch.Generator(device=gpu).manual_seed(2)
    x = (torch.rand(rows, 31, device=gpu, generator=g) * 2 - 1).requires_grad_(True)
    ws = torch.rand(rows, device=gpu, generator=g).requires_grad_(True)
    gi = torch.randn(rows, 3, device=gpu, generator=g)
    raw, img = ops.small_mlp_train(x, layers, ops.SMALL_ACT_SIGMOID_BG, ws, bg)
    (img * gi).sum().backward()
    got = [x.grad.clone(), ws.grad.clone()] + [l.weight.grad.clone() for l in layers]
    x.grad = ws.grad = None
    for l in layers:
        l.weight.grad = None
    img2 = torch.sigmoid(_torch_mlp(x, layers)) + (1 - ws).unsqueeze(-1) * bg
    (img2 * gi).sum().backward()
    assert float((img - img2).abs().max()) < 2e-6
    for a, b in zip(got, [x.grad, ws.grad] + [l.weight.grad for l in layers]):
        assert rel(a, b) < 2e-5, rel(a, b)


def test_mlp_module_takes_the_fused_route_and_keeps_the_torch_route(gpu):
    """nerf.network.MLP: fused for the instantiated bias-free shapes under autograd; torch layers for a shape outside the list, with a bias,
    or without autograd -- same numbers either way."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.nerf.network import MLP
    torch.manual_seed(0)
    m = MLP(32, 16, 64, 3, bias=False).to(gpu)
    x = torch.rand(3000, 32, device=gpu)
    assert ops.small_mlp_fusable(x, list(m.net))
    y = m(x)
    with torch.no_grad():
        y0 = m(x)
    assert rel(y, y0) < 2e-6
    other = MLP(20, 16, 64, 3, bias=False).to(gpu)
    assert not ops.small_mlp_fusable(torch.rand(8, 20, device=gpu), list(other.net))
    assert other(torch.rand(8, 20, device=gpu)).shape == (8, 16)
    biased = MLP(32, 16, 64, 3, bias=True).to(gpu)
    assert not ops.small_mlp_fusable(x, list(biased.net))


def test_ray_composite_forward_backward_vs_torch(gpu):
    """sn_rm_ray_composite[_backward] against the reference's expressions (renderer.py:327-347, network.py:164-170: per-sample colour =
    cat([geo_feat, SH(d)]) composited with the weights) in fp64 torch with the package's SH encoder."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.shencoder import SHEncoder
    N, T = 777, 32
    g = torch.Generator(device=gpu).manual_seed(4)
    w = torch.rand(N, T, device=gpu, generator=g).requires_grad_(True)
    tm = torch.rand(N, T, device=gpu, generator=g) * 5
    raw = torch.randn(N, T, 16, device=gpu, generator=g).requires_grad_(True)
    d = torch.randn(N, 3, device=gpu, generator=g) * 1.3
    gws, gd, gf = (torch.randn(N, device=gpu, generator=g), torch.randn(N, device=gpu, generator=g), torch.randn(N, 31, device=gpu, generator=g))
    ws, depth, f = rm.ray_composite(w, tm, raw, d)
    ((ws * gws).sum() + (depth * gd).sum() + (f * gf).sum()).backward()
    got = (w.grad.clone(), raw.grad.clone())
    w.grad = raw.grad = None
    sh = SHEncoder(degree=4)(d / d.norm(dim=-1, keepdim=True)).double()
    w64, raw64 = w.detach().double().requires_grad_(True), raw.detach().double().requires_grad_(True)
    color = torch.cat([raw64[..., 1:], sh.unsqueeze(1).expand(N, T, 16)], dim=-1)
    ws2, depth2, f2 = w64.sum(-1), (w64 * tm.double()).sum(-1), (w64.unsqueeze(-1) * color).sum(1)
    ((ws2 * gws.double()).sum() + (depth2 * gd.double()).sum() + (f2 * gf.double()).sum()).backward()
    assert rel(ws, ws2) < 1e-6 and rel(depth, depth2) < 1e-6 and rel(f, f2) < 2e-6
    assert rel(got[0], w64.grad) < 2e-6
    assert rel(got[1], raw64.grad) < 1e-6
    assert float(got[1][..., 0].abs().max()) == 0.0


def test_jitter_kernel_matches_the_reference_expressions(gpu):
    from sanerf_hq_amd import raymarching as rm
    N = 300
    for T in (129, 65, 33, 2):
        r = torch.rand(N, T, device=gpu)
        b = rm.jitter(r, N, T, 0)
        want = (torch.linspace(0, 1, T, device=gpu).unsqueeze(0) + (r - 0.5) / (T - 1)).clamp(0, 1)      # renderer.py:262-270
        assert float((b - want).abs().max()) < 2e-7
        u = rm.jitter(r, N, T, 1)
        want = torch.linspace(0.5 / T, 1 - 0.5 / T, steps=T, device=gpu).unsqueeze(0) + (r - 0.5) / T       # renderer.py:97-102
        assert float((u - want).abs().max()) < 2e-7
        # no jitter: the plain linspace rows (aten's scalar recipe, as sn_rm_sample_pdf computes its own u; torch's vectorised CPU kernel and its
        # device kernel round some entries the other way: one ulp)
        assert float((rm.jitter(None, N, T, 0, device=gpu).cpu() - torch.linspace(0, 1, T).unsqueeze(0)).abs().max()) <= 6e-8
        assert float((rm.jitter(None, N, T, 1, device=gpu).cpu() - torch.linspace(0.5 / T, 1 - 0.5 / T, steps=T).unsqueeze(0)).abs().max()) <= 6e-8


def test_sample_positions_unit_cube_output(gpu):
    from sanerf_hq_amd import raymarching as rm
    N, T = 500, 64
    g = torch.Generator(device=gpu).manual_seed(9)
    ro = torch.randn(N, 3, device=gpu, generator=g) * 0.3
    rd = torch.randn(N, 3, device=gpu, generator=g)
    nears, fars = rm.near_far_from_aabb(ro, rd, torch.tensor([-128.0] * 3 + [128.0] * 3), 0.2)
    bins = torch.sort(torch.rand(N, T + 1, device=gpu, generator=g), dim=-1).values
    rb, rt, xyz = rm.sample_positions(ro, rd, nears, fars, bins, contract=True)
    rb2, rt2, x01 = rm.sample_positions(ro, rd, nears, fars, bins, contract=True, grid_bound=2.0)
    assert torch.equal(rb, rb2) and torch.equal(rt, rt2)
    assert torch.equal(x01, (xyz + 2.0) / 4.0)                      # gridencoder/grid.py:156 (a power-of-two bound: exact either way)


def test_proposal_loss_single_node_equals_the_per_stage_nodes(gpu):
    from sanerf_hq_amd import raymarching as rm
    N = 600
    g = torch.Generator(device=gpu).manual_seed(12)

    def stage(T):
        b = torch.sort(torch.rand(N, T + 1, device=gpu, generator=g), dim=-1).values
        w = torch.rand(N, T, device=gpu, generator=g)
        return b, (w / w.sum(-1, keepdim=True)).requires_grad_(True)
    (b0, w0), (b1, w1), (b2, w2) = stage(128), stage(64), stage(32)
    one = rm.proposal_loss_all([b0, b1, b2], [w0, w1, w2])
    (one * 1.7).backward()
    g_one = (w0.grad.clone(), w1.grad.clone())
    assert w2.grad is None
    w0.grad = w1.grad = None
    two = rm.proposal_loss_stage(b0, w0, b2, w2) + rm.proposal_loss_stage(b1, w1, b2, w2)
    (two * 1.7).backward()
    assert abs(float(one) - float(two)) < 1e-6 * max(1.0, abs(float(two)))
    assert rel(g_one[0], w0.grad) < 1e-6 and rel(g_one[1], w1.grad) < 1e-6


@pytest.mark.parametrize("update_proposal", [True, False])
def test_fused_training_route_equals_the_operator_chain(gpu, update_proposal):
    """NeRFRenderer._run_autograd_unit against _run_autograd (the rounds 2-5 route through grid_encode + torch layers + SH + composite) on the
    same model, perturb=False: image, depth, losses and the gradient of every parameter within fp32 round-off."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    opt = make_opt()
    opt.lambda_proposal, opt.lambda_distort = 1.0, 0.01
    N = 2048
    H = W = 128
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    pix = torch.from_numpy((synth.hash_u01(N, 5) * (H * W)).astype(np.int64)).to(gpu)
    ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
    gt = torch.from_numpy(synth.hash_uniform((N, 3), 42, 0.0, 1.0)).to(gpu)
    res = {}
    for fused in (True, False):
        model = NeRFNetwork(opt)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], seed=1).items()}, strict=False)
        model = model.to(gpu).train()
        model.fused_training_ops = fused
        from sanerf_hq_amd import ops
        ops.SMALL_MLP_FUSED = fused
        try:
            o = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=update_proposal)
            loss = torch.nn.functional.mse_loss(o["image"], gt) + 0.01 * o["distort_loss"]
            if update_proposal:
                loss = loss + o["proposal_loss"]
            loss.backward()
        finally:
            ops.SMALL_MLP_FUSED = True
        res[fused] = (o, loss, {n: p.grad for n, p in model.named_parameters() if p.grad is not None})
    (o1, l1, g1), (o0, l0, g0) = res[True], res[False]
    assert float((o1["image"] - o0["image"]).abs().max()) < 5e-6
    assert rel(o1["depth"], o0["depth"]) < 2e-6 and rel(o1["weights_sum"], o0["weights_sum"]) < 2e-6
    assert abs(float(l1) - float(l0)) < 1e-6 * max(1.0, abs(float(l0)))
    assert set(g1) == set(g0) and len(g1) == (13 if update_proposal else 7)
    for n in g0:
        assert rel(g1[n], g0[n]) < 2e-4, (n, rel(g1[n], g0[n]))


def test_fused_training_route_with_jitter_runs_and_is_seed_reproducible(gpu):
    """perturb=True: one torch.rand call feeds every stage; the same seed gives the same image and gradients bit for bit."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    opt = make_opt()
    opt.lambda_proposal = 1.0
    N = 1024
    H = W = 64
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    ro, rd = roF[:N].contiguous(), rdF[:N].contiguous()
    model = NeRFNetwork(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], seed=1).items()}, strict=False)
    model = model.to(gpu).train()
    runs = []
    for _ in range(2):
        torch.manual_seed(123)
        model.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=True, update_proposal=True)
        (o["image"].square().mean() + o["proposal_loss"]).backward()
        runs.append((o["image"].detach().clone(), model.grid_mlp.net[0].weight.grad.clone(), model.prop_mlp[0].net[0].weight.grad.clone()))
    assert bool(torch.isfinite(runs[0][0]).all())
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    torch.manual_seed(124)
    o2 = model.render(ro, rd, staged=False, bg_color=1, perturb=True, update_proposal=True)
    assert not torch.equal(o2["image"], runs[0][0])


def test_sam_head_input_written_in_place_equals_the_concatenation(gpu):
    """sn_render_io.head_stride: the fused render writes f_sam | f_image | rgb | depth straight into the [N, 163] input of samvit_mlp
    (renderer.py:366 concatenates four tensors).  Bit-equal to the concatenation of the dense outputs, for image-order tiles and for a small
    linear-order batch (the several-lanes-per-ray kernels), and the SAM feature map is unchanged."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    model = NeRFNetwork(make_opt(with_sam=True))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], heads=True, seed=1).items()}, strict=False)
    model = model.to(gpu).eval()
    for H, W, tile in ((96, 96, True), (40, 50, False)):
        ro, rd = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
        plan = model._get_plan(with_feat=True)
        with torch.no_grad():
            a = rm.render_rays(plan, ro, rd, tile_w=W if tile else 0, want=("f_image",), out={})
            b = rm.render_rays(plan, ro, rd, tile_w=W if tile else 0, want=("f_image",), out={}, head_input=True)
            want = torch.cat([a["f_feat"], a["f_image"], a["image"], a["depth"].unsqueeze(-1)], dim=-1)
            assert b["head_input"].shape == (H * W, 163) and torch.equal(b["head_input"], want)
            assert torch.equal(b["image"], a["image"]) and torch.equal(b["depth"], a["depth"]) and torch.equal(b["weights_sum"], a["weights_sum"])
            o1 = model.render(ro, rd, staged=False, bg_color=1, perturb=False, return_feats=1, H=H, W=W, tile_w=W if tile else 0)
            ref = model._head_mlp(model.samvit_mlp, want).view(H, W, -1)
            assert torch.equal(o1["samvit"], ref)


@pytest.mark.parametrize("f16", [False, True])
def test_feature_stage_lds_patch_is_bit_identical(gpu, f16):
    """sn_render_tuning.feat_patch = 1 (per-wave LDS staging of the dense levels' voxels in k_feat_stage; opt-in because it is
    slower): f_feat equals the direct-gather kernel's bit for bit, image-order tiles and an odd-sized image, fp32 and fp16 tables."""
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    model = NeRFNetwork(make_opt(with_sam=True))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], heads=True, seed=1).items()}, strict=False)
    model = model.to(gpu).eval()
    for H, W in ((96, 96), (37, 53)):
        ro, rd = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
        plan = rm.RenderPlan(model, [128, 64, 32], torch.float16 if f16 else torch.float32, feat_encoder=model.s_grid)
        a = rm.render_rays(plan, ro, rd, tile_w=W, out={}, tuning=rm.Tuning(feat_patch=0))
        for lg in (0, 1, 4):
            b = rm.render_rays(plan, ro, rd, tile_w=W, out={}, tuning=rm.Tuning(feat_patch=1, feat_levels=lg))
            assert torch.equal(a["f_feat"], b["f_feat"]), (H, W, lg)
