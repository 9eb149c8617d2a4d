"""GPU tests of the training path: the small perceptrons on the matrix cores (csrc/mlp_small.hip) and the fused training operators of the RGB step against torch, the wide training MLP (forward modes, one-kernel backward, sign bits, weight gradients beside the backward pass), losses, optimiser, HIP-graph replay of the steps, the torch-formulated cold routes.  The reference's own autograd pins the whole steps in tests/test_gpu_render.py (tests/golden/train_*.npz)."""
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
    assert float((o1["image"] - o0["image"]).abs().max()) < 1e-5            # (two fp32 routes through a sigmoid head: the chain of layer-by-layer products against the one-kernel perceptrons)
    assert rel(o1["depth"], o0["depth"]) < 2e-6 and rel(o1["weights_sum"], o0["weights_sum"]) < 2e-6
    assert abs(float(l1) - float(l0)) < 1e-6 * max(1.0, abs(float(l0)))
    assert set(g1) == set(g0) and len(g1) == (13 if update_proposal else 7)
    for n in g0:      # (2.0e-4 on grid.embeddings since the chain's layers run the library's own product: two fp32 routes, a few ReLU units at zero flip; the project bar is 1e-3)
        assert rel(g1[n], g0[n]) < 4e-4, (n, rel(g1[n], g0[n]))


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


@pytest.mark.parametrize("N,K", [(4096, 2), (777, 5), (33, 32), (1, 1)])
def test_mask_nll_kernel_vs_the_trainers_torch_lines(gpu, N, K):
    """rm.mask_nll = nerf/trainer.py:419-428 (softmax -> clamp(eps, 1 - eps) -> gather -> -log) per ray, value and gradient from one kernel:
    against those torch lines, including logits large enough for the clamp to bind (no gradient there, as torch.clamp's backward)."""
    from sanerf_hq_amd import raymarching as rm
    torch.manual_seed(N + K)
    logits = (torch.randn(N, K, device=gpu) * 6.0).requires_grad_(True)
    labels = torch.randint(0, K, (N,), device=gpu)
    eps = 1e-6
    pm = torch.softmax(logits, dim=-1).clamp(min=eps, max=1 - eps)
    ref = -torch.log(torch.gather(pm, -1, labels[..., None]))
    (gref,) = torch.autograd.grad(ref.mean(), logits)
    lg2 = logits.detach().clone().requires_grad_(True)
    got = rm.mask_nll(lg2, labels, eps)
    assert got.shape == ref.shape
    got.mean().backward()
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(lg2.grad.cpu().numpy(), gref.cpu().numpy(), rtol=1e-5, atol=1e-9)


def test_capturable_adam_equals_the_host_counter_form(gpu):
    """optim.Adam(capturable=True) reads its step count from the device when the kernel runs: same parameters as the host-counter form."""
    from sanerf_hq_amd.optim import Adam
    torch.manual_seed(3)
    w0 = torch.randn(1000, 7, device=gpu)
    grads = [torch.randn_like(w0) * (torch.rand_like(w0) > 0.3) for _ in range(6)]
    res = []
    for cap in (False, True):
        w = w0.clone().requires_grad_(True)
        opt = Adam([w], lr=1e-2, eps=1e-15, capturable=cap)
        for g in grads:
            w.grad = g.clone()
            opt.step()
        res.append(w.detach().clone())
        assert float(opt.state[w]["step"]) == 6.0 and opt.state[w]["step"].is_cuda == cap
    assert float((res[0] - res[1]).abs().max()) <= 1e-7


def _c5_like_step(gpu, seed, capturable):
    from helpers import make_opt
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    from sanerf_hq_amd.optim import Adam
    params = synthetic_params([128, 64, 32], heads=True, seed=1)
    model = NeRFNetwork(make_opt(with_sam=False, with_mask=True))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model = model.to(gpu).train()
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("m_grid") or n_.startswith("mask_mlp"))
    H = W = 128
    N = 2048
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    pix = torch.from_numpy((synth.hash_u01(N, seed) * (H * W)).astype(np.int64)).to(gpu)
    ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
    labels = torch.from_numpy((synth.hash_u01(N, seed + 1) < 0.5).astype(np.int64)).to(gpu)
    # eps = 1e-8 here (the trainer's 1e-15 turns a gradient of 1e-12 -- summation-order noise of the binned scatter -- into a full +-lr move,
    # which would make two correct runs differ by whole steps on elements that receive no signal)
    opt = Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-8, capturable=capturable)

    def step():
        opt.zero_grad(set_to_none=True)
        o = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=False, return_mask=1)
        loss = rm.mask_nll(o["instance_mask_logits"], labels, 1e-6).mean()
        loss.backward()
        opt.step()
        return loss.detach()
    return model, step


def test_mask_training_step_replayed_as_a_hip_graph_tight_bounds_with_the_blas_forward(gpu):
    """The same replay-vs-eager comparison with the fp32 BLAS forward (ops.WIDE_MLP_FORWARD_F16X3 = False), under which no hidden unit of this
    set-up sits within round-off of zero: the bounds of rounds 3-4 hold (1e-3 max-abs, 1e-4 relative norm) -- tight enough to catch a single
    stale buffer or a kernel missing from the captured graph in a 6-step run (advisor, round 5)."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.graph import GraphedStep
    steps = 6
    ops.WIDE_MLP_FORWARD_F16X3 = False
    try:
        m_eager, step_eager = _c5_like_step(gpu, 99, False)
        losses_e = [float(step_eager()) for _ in range(steps)]
        m_graph, step_graph = _c5_like_step(gpu, 99, True)
        g = GraphedStep(step_graph, warmup=2)
        losses_g = [float(g()) for _ in range(steps - 2)]
        torch.cuda.synchronize()
    finally:
        ops.WIDE_MLP_FORWARD_F16X3 = True
    assert all(np.isfinite(losses_g)) and abs(losses_g[-1] - losses_e[-1]) <= 1e-5 * max(1.0, abs(losses_e[-1]))
    for (n1, p1), (n2, p2) in zip(m_eager.named_parameters(), m_graph.named_parameters()):
        if p1.requires_grad:
            assert float((p1 - p2).abs().max()) <= 1e-3, n1
            assert float((p1 - p2).double().norm() / (p1.double().norm() + 1e-12)) <= 1e-4, n1


def test_mask_training_step_replayed_as_a_hip_graph(gpu):
    """BASELINE configs[4] as ONE HIP graph (sanerf_hq_amd.graph.GraphedStep): frozen-field render, m_grid, mask MLP, fused NLL, binned grid
    backward, capturable single-pass Adam -- captured once, replayed; after the same number of steps the parameters equal the eager run's
    (the binned scatter adds in a scheduling-dependent order: last-bit differences, amplified by Adam's normalisation on tiny gradients)."""
    from sanerf_hq_amd.graph import GraphedStep
    steps = 6
    m_eager, step_eager = _c5_like_step(gpu, 99, False)
    losses_e = [float(step_eager()) for _ in range(steps)]
    m_graph, step_graph = _c5_like_step(gpu, 99, True)
    g = GraphedStep(step_graph, warmup=2)                       # 2 warm-up steps + the capture pass (which does not execute)
    losses_g = []
    for _ in range(steps - 2):
        losses_g.append(float(g()))
        junk = torch.full((1 << 20,), float("nan"), device=gpu)     # allocator traffic between replays: the graph must own everything it reads
        del junk
    torch.cuda.synchronize()
    assert all(np.isfinite(losses_g)) and abs(losses_g[-1] - losses_e[-1]) <= 1e-4 * max(1.0, abs(losses_e[-1]))
    for (n1, p1), (n2, p2) in zip(m_eager.named_parameters(), m_graph.named_parameters()):
        if p1.requires_grad:
            d = float((p1 - p2).abs().max())
            # two EAGER runs of this step differ from each other just as much (tools/graph_vs_eager.py eager2): Adam's normalisation
            # amplifies last-bit differences of near-zero gradients (2.5e-4 on ~1e3 table elements after 6 steps), and when that noise lands on a
            # sample whose hidden unit sits within 1e-7 of zero the LeakyReLU branch of that unit flips in one run and not the other (one such unit
            # exists in this set-up with the fp32-MFMA forward: then 2.4e-3 on ~1e5 elements, always the same numbers).  A broken replay -- a stale
            # buffer, a kernel missing from the graph -- moves every element by whole steps (6e-3) and the norm by O(1); bounds: 5 steps of lr, 1e-3.
            assert d <= 5e-3, (n1, d)
            assert float((p1 - p2).double().norm() / (p1.double().norm() + 1e-12)) <= 1e-3, n1


def test_wide_mlp_weight_gradients_beside_the_backward_pass(gpu):
    """ops.WGRAD_SIDE_STREAM: the wide training MLP's weight gradients run on a second stream and the backward pass joins it in an engine callback
    at its end.  Same gradients as the inline launch (bit for bit: same kernels, same inputs); with a gradient already present on a parameter
    (accumulation: AccumulateGrad launches an add) the launch stays inline; allocator traffic right after backward() must not disturb them."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(7)
    mlp = SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False).to(gpu)
    x = torch.randn(32768, 143, device=gpu, requires_grad=True)

    def grads(side, keep=False):
        ops.WGRAD_SIDE_STREAM = side
        if not keep:
            for p in mlp.parameters():
                p.grad = None
        x.grad = None
        (mlp(x) ** 2).sum().backward()
        junk = torch.full((1 << 22,), float("nan"), device=gpu)          # (would land in freed blocks of the backward pass)
        del junk
        return [p.grad.clone() for p in mlp.parameters()] + [x.grad.clone()]
    try:
        assert ops.wide_mlp_fusable(x, list(mlp.net), [])
        a, b = grads(True), grads(False)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        c = grads(True, keep=True)                                        # accumulation on top of b's gradients: inline path
        for u, v in zip(c[:-1], b[:-1]):
            assert torch.allclose(u, 2 * v, rtol=1e-6, atol=0)
    finally:
        ops.WGRAD_SIDE_STREAM = False


def test_wgrad_side_stream_is_opt_in_and_safe_for_a_shared_weight(gpu):
    """ops.WGRAD_SIDE_STREAM (advisor, round 4): off by default; when on, an MLP applied TWICE inside one graph (the engine adds the two
    weight gradients on the main stream before AccumulateGrad) keeps both launches inline -- gradients equal those of the flag off, bit for
    bit -- and a parameter with a post-accumulate-grad hook never takes the side stream."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    assert ops.WGRAD_SIDE_STREAM is False
    torch.manual_seed(3)
    mlp = SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False).to(gpu)
    xa = torch.randn(32768, 143, device=gpu)
    xb = torch.randn(32768, 143, device=gpu)

    def grads(side):
        ops.WGRAD_SIDE_STREAM = side
        for p in mlp.parameters():
            p.grad = None
        ((mlp(xa) ** 2).sum() + (mlp(xb) * 3.0).sum()).backward()
        junk = torch.full((1 << 22,), float("nan"), device=gpu)
        del junk
        return [p.grad.clone() for p in mlp.parameters()]
    try:
        assert ops.wide_mlp_fusable(xa, list(mlp.net), [])
        a, b = grads(False), grads(True)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        assert not ops._shared_now                       # cleared by the engine callback at the end of the pass
        seen = []
        h = mlp.net[0].weight.register_post_accumulate_grad_hook(lambda p: seen.append(float(p.grad.abs().sum())))
        ops.WGRAD_SIDE_STREAM = True
        assert not ops._beside_ok([mlp.net[0].weight])
        for p in mlp.parameters():
            p.grad = None
        (mlp(xa) ** 2).sum().backward()
        torch.cuda.synchronize()
        assert seen and abs(seen[0] - float(mlp.net[0].weight.grad.abs().sum())) <= 1e-3 * seen[0]
        h.remove()
    finally:
        ops.WGRAD_SIDE_STREAM = False


@pytest.mark.gpu
@pytest.mark.parametrize("N,din,n_out,nl,leaky", [(20000, 143, 16, 3, True), (64 * 7 + 5, 143, 2, 3, True), (16384 + 77, 64, 5, 3, False),
                                                  (4096, 256, 256, 2, True), (1000, 1, 40, 4, True), (33, 37, 33, 1, True)])
def test_wide_mlp_native_fp32_forward(gpu, N, din, n_out, nl, leaky):
    """sn_mlp_wide_forward_train (one kernel, v_mfma_f32_32x32x2_f32, activations fused, hidden outputs saved) against the layer-by-layer
    torch forward in fp32 and in fp64: as close to the fp64 result as BLAS fp32 is (both are fp32 sums in another order), hidden outputs
    equal to what torch's in-place activation leaves for autograd up to that round-off, ragged row counts and widths."""
    import ctypes as C
    from sanerf_hq_amd import _lib, synth
    dims = [din] + [256] * (nl - 1) + [n_out]
    ws = [torch.from_numpy(synth.linear_weight(dims[i + 1], dims[i], 900 + i, 2.0)).to(gpu) for i in range(nl)]
    rng = np.random.default_rng(N)
    x = torch.from_numpy(rng.standard_normal((N, din)).astype(np.float32)).to(gpu)
    act = (lambda t: torch.nn.functional.leaky_relu(t)) if leaky else torch.relu

    def ref(dtype):
        h, hs = x.to(dtype), []
        for i, w in enumerate(ws):
            h = torch.nn.functional.linear(h, w.to(dtype))
            if i + 1 < nl:
                h = act(h)
                hs.append(h)
        return h, hs

    y32, h32 = ref(torch.float32)
    y64, h64 = ref(torch.float64)
    desc = _lib.MlpDesc()
    desc.num_layers, desc.activation, desc.skip_mask = nl, 1 if leaky else 0, 0
    desc.dims[0] = din
    for i, w in enumerate(ws):
        desc.weight[i], desc.bias[i], desc.dims[i + 1] = w.data_ptr(), None, w.shape[0]
    hs = [torch.full((N, 256), float("nan"), device=gpu) for _ in range(nl - 1)]
    y = torch.full((N, n_out), float("nan"), device=gpu)
    hid = (C.c_void_p * max(nl - 1, 1))(*[t.data_ptr() for t in hs])
    _lib.check(_lib.lib().sn_mlp_wide_forward_train(C.byref(desc), x.data_ptr(), N, hid, y.data_ptr(), _lib.stream()), "sn_mlp_wide_forward_train")
    torch.cuda.synchronize()

    def err(a, b64):
        return float((a.double() - b64).norm() / b64.norm().clamp_min(1e-30))

    assert torch.isfinite(y).all()
    e_native, e_blas = err(y, y64), err(y32, y64)
    assert e_native < max(2.0 * e_blas, 2e-7), (e_native, e_blas)
    for a, b32, b64 in zip(hs, h32, h64):
        assert torch.isfinite(a).all()
        assert err(a, b64) < max(2.0 * err(b32, b64), 2e-7)
    # and through the autograd route of the training MLP: same outputs, saved tensors feed the fused backward
    from sanerf_hq_amd import ops
    if nl >= 2 and N >= ops.WIDE_MLP_BACKWARD_MIN_ROWS and n_out <= 256:
        xs = x.clone().requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        ops.WIDE_MLP_FORWARD_NATIVE = True             # (opt-in: measured slower than the BLAS forward, csrc/mlp_f32.inc)
        try:
            ya = ops._wide_mlp_train.apply(xs, leaky, *wl)
        finally:
            ops.WIDE_MLP_FORWARD_NATIVE = False
        assert torch.equal(ya, y)
        ya.backward(torch.ones_like(ya))
        # the default training forward (split-fp16 x3 on the inference kernel, hidden outputs saved): outputs and saved tensors within the
        # split's 2^-22 per product of the fp64 result, gradients through the same fused backward
        xs2 = x.clone().requires_grad_(True)
        wl2 = [w.clone().requires_grad_(True) for w in ws]
        assert ops.WIDE_MLP_FORWARD_F16X3
        yc = ops._wide_mlp_train.apply(xs2, leaky, *wl2)
        assert err(yc.detach(), y64) < 2e-6
        for a, b64 in zip(yc.grad_fn.saved_tensors[1:nl], h64):
            assert err(a, b64) < 2e-6 and torch.isfinite(a).all()
        yc.backward(torch.ones_like(yc))
        for wa_, wb_ in zip(wl2, wl):
            assert float((wa_.grad - wb_.grad).norm() / wb_.grad.norm()) < 1e-3
        assert all(w.grad is not None and torch.isfinite(w.grad).all() for w in wl) and torch.isfinite(xs.grad).all()


@pytest.mark.gpu
def test_forward_cat_under_autograd_matches_cat_of_the_encoder(gpu):
    """GridEncoder.forward_cat in training (ops._grid_encode_cat: grid features and the detached extra channels in one forward kernel) against
    torch.cat([enc(x), extra.detach()]) as the reference writes it (renderer.py:380): same values, same table gradient (both through the
    binned scatter: equal up to its summation order), no gradient to the extra channels."""
    from sanerf_hq_amd.gridencoder import GridEncoder
    torch.manual_seed(5)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=15, desired_resolution=256).to(gpu)
    B, E = 40000, 15
    x = (torch.rand(B, 3, device=gpu) * 2 - 1) * 0.98
    extra = torch.randn(B, E, device=gpu, requires_grad=True)
    gy = torch.randn(B, 16 * 8 + E, device=gpu)
    a = enc.forward_cat(x, extra, bound=1.0)
    assert a.requires_grad
    a.backward(gy)
    ga = enc.embeddings.grad.clone()
    assert extra.grad is None                       # detached, as in the reference
    enc.embeddings.grad = None
    b = torch.cat([enc(x, bound=1.0), extra.detach()], dim=-1)
    b.backward(gy)
    gb = enc.embeddings.grad.clone()
    assert torch.equal(a.detach(), b.detach())
    assert float((ga - gb).norm() / gb.norm()) < 1e-6
    assert torch.equal(ga != 0, gb != 0)
    with torch.no_grad():                           # inference route: the same values
        c = enc.forward_cat(x, extra, bound=1.0)
    assert torch.equal(c, b.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("N,din,n_out,leaky", [(20000, 143, 2, True), (16384 + 77, 64, 5, False), (40000, 143, 16, True)])
def test_wide_mlp_backward_from_sign_bits_is_bit_identical(gpu, N, din, n_out, leaky):
    """The split-fp16 training forward also writes one sign bit per hidden unit (32 bytes per row and layer); the backward data path that reads
    those (sn_mlp_wide_backward_bits, k_mlp_wide<5>) must give exactly what the one reading the [N, 256] fp32 outputs gives (k_mlp_wide<4>):
    same arithmetic, the branch of every unit taken from a bit instead of a comparison."""
    from sanerf_hq_amd import ops, synth
    ws = [torch.from_numpy(synth.linear_weight(o, i, 950 + k, 2.0)).to(gpu) for k, (o, i) in enumerate([(256, din), (256, 256), (n_out, 256)])]
    rng = np.random.default_rng(N)
    x = torch.from_numpy(rng.standard_normal((N, din)).astype(np.float32)).to(gpu)
    gy = torch.from_numpy((rng.standard_normal((N, n_out)) * 10.0 ** rng.uniform(-9, -1, (N, 1))).astype(np.float32)).to(gpu)

    def run(bits):
        ops.WIDE_MLP_SIGN_BITS = bits
        try:
            xs = x.clone().requires_grad_(True)
            wl = [w.clone().requires_grad_(True) for w in ws]
            y = ops._wide_mlp_train.apply(xs, leaky, *wl)
            assert bool(y.grad_fn.sign_bits) == bits
            y.backward(gy)
            return y.detach(), xs.grad, [w.grad for w in wl]
        finally:
            ops.WIDE_MLP_SIGN_BITS = True

    ya, gxa, gwa = run(True)
    yb, gxb, gwb = run(False)
    assert torch.equal(ya, yb) and torch.equal(gxa, gxb)
    for a_, b_ in zip(gwa, gwb):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("steps", [[64, 32], [32], [48, 24, 16], [128, 64, 32]])
@pytest.mark.parametrize("N", [1, 77, 1500])
def test_fused_training_route_other_schedules_and_ray_counts(gpu, steps, N):
    """The fused training route on other schedules than the reference's default (one, two, three stages; sample counts that are not powers
    of two) and ray counts that fill neither a wave nor a 64-row matrix tile: image, proposal loss and every gradient against the operator chain.
    Bounds: two correct fp32 implementations of a RESAMPLING render differ by more than round-off -- a 1e-7 change of a proposal weight moves the
    next stage's sample positions, and the hash field turns that into ~1e-5 on the weights (measured: single stage 1e-7, [128,64,32] 4e-6,
    [48,24,16] 9e-6; image 3e-5) -- so the image is held to the north star's 1e-4; and in a batch of a few thousand samples ONE ReLU unit that takes
    the other branch under a differently ordered sum moves a gradient tensor by ~1e-3 relative (77 rays x 32 samples: 1.1e-3 on the table), so
    small batches get 5e-3.  A wrong layout or a missing term shows as O(1)."""
    from sanerf_hq_amd import ops, raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    opt = make_opt()
    opt.num_steps = list(steps)
    opt.lambda_proposal, opt.lambda_distort = 1.0, 0.0
    H = W = 64
    roF, rdF = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 60.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    pix = torch.from_numpy((synth.hash_u01(N, 5) * (H * W)).astype(np.int64)).to(gpu)
    ro, rd = roF[pix].contiguous(), rdF[pix].contiguous()
    res = {}
    for fused in (True, False):
        model = NeRFNetwork(opt)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_params([128, 64, 32], seed=1).items()}, strict=False)
        model = model.to(gpu).train()
        model.fused_training_ops = fused
        ops.SMALL_MLP_FUSED = fused
        try:
            o = model.render(ro, rd, staged=False, bg_color=0.5, perturb=False, update_proposal=True)
            loss = o["image"].square().mean() + o["proposal_loss"]
            loss.backward()
        finally:
            ops.SMALL_MLP_FUSED = True
        res[fused] = (o, float(loss), {n: p.grad for n, p in model.named_parameters() if p.grad is not None})
    (o1, l1, g1), (o0, l0, g0) = res[True], res[False]
    assert float((o1["image"] - o0["image"]).abs().max()) < 1e-4
    assert abs(l1 - l0) < 2e-5 * max(1.0, abs(l0))
    assert set(g1) == set(g0)
    for n in g0:
        assert rel(g1[n], g0[n]) < (5e-3 if N < 1000 else 1e-3) or float((g1[n] - g0[n]).abs().max()) < 1e-9, (n, rel(g1[n], g0[n]))


def test_training_operators_accept_empty_batches(gpu):
    """N = 0 through the round-6 entry points: nothing is launched, shapes are right."""
    from sanerf_hq_amd import ops, raymarching as rm
    layers = _layers((32, 64, 64, 16), gpu, 1)
    x = torch.zeros(0, 32, device=gpu, requires_grad=True)
    raw, sig = ops.small_mlp_train(x, layers, ops.SMALL_ACT_TRUNC_EXP0)
    assert raw.shape == (0, 16) and sig.shape == (0,)
    (raw.sum() + sig.sum()).backward()
    assert x.grad.shape == (0, 32)
    ws, depth, f = rm.ray_composite(torch.zeros(0, 32, device=gpu), torch.zeros(0, 32, device=gpu), torch.zeros(0, 32, 16, device=gpu), torch.zeros(0, 3, device=gpu))
    assert ws.shape == (0,) and f.shape == (0, 31)
    assert rm.jitter(None, 0, 33, 0, device=gpu).shape == (0, 33)
