"""GPU tests of the feature heads: the 256-wide perceptrons on the matrix cores (k_mlp_wide_j, k_mask16), the one-kernel mask head against the unfused composition, exact skipping of zero weights, the in-render feature stage (incl. its opt-in LDS patch) and the SAM head's input written in place."""
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


def test_heads_skip_exactly_zero_weights(gpu):
    """An opaque field (synthetic MLPs with gain 40: sigma is ~0 or huge) leaves most last-stage samples with weight EXACTLY 0 -- behind the
    surface the transmittance has underflowed, in front of it alpha = 0.  The fused mask head skips a 128-sample tile whose weights are all
    zero, the in-render feature stage a sample index at which a whole wave's weights are zero: both results must equal what the kernels that
    evaluate every sample produce (stand-alone grid_composite: bit for bit; three-kernel mask route: summation order apart)."""
    from helpers import make_opt
    from sanerf_hq_amd import raymarching as rm, synth
    from sanerf_hq_amd.nerf import NeRFNetwork
    H = W = 128
    ro, rd = rm.generate_rays(synth.orbit_pose(1.0, 20.0, 30.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    params = synthetic_params([128, 64, 32], heads=True, seed=3, gain=40.0)
    model = NeRFNetwork(make_opt(with_sam=True, with_mask=True))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    model = model.to(gpu).eval()
    with torch.no_grad():
        o = rm.render_rays(model._get_plan(with_feat=True), ro, rd, tile_w=W, want=("weights_last", "xyzs_last"))
        wl = o["weights_last"]
        assert float((wl == 0).float().mean()) > 0.5, "the scene is meant to have mostly exact-zero weights"
        ref = rm.grid_composite(wl, o["xyzs_last"], model.s_grid, model.bound, tile_w=W)
        assert torch.equal(o["f_feat"], ref)
        model.fused_mask_head = True
        a = model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)["instance_mask_logits"].clone()
        model.fused_mask_head = False
        b = model.render(ro, rd, staged=False, perturb=False, return_mask=1, H=H, W=W, tile_w=W)["instance_mask_logits"]
    assert torch.isfinite(a).all()
    assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("N", [128, 1000, 160000])
def test_wide_mlp_pair_interleaved_chunks_match_torch(gpu, N):
    """k_mlp_wide_j walks a k-step's eight output tiles in interleaved pairs (round 5: no filler instruction between an MFMA and the MFMA
    that consumes its accumulator).  Same products in the same order per accumulator as rounds 3-4: the SAM head MLP (skip layer, biases,
    LayerNorm: network.py:107-116) and the mask MLP (network.py:118-123) against torch fp32 at the split-fp16 contract (2^-21 per product)."""
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(N)
    for mlp, ln in ((SkipConnMLP(163, 256, 256, 5, skip_layers=[2], bias=True), torch.nn.LayerNorm(256)),
                    (SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False), None)):
        mlp = mlp.to(gpu)
        ln = ln.to(gpu) if ln is not None else None
        x = torch.randn(N, mlp.dim_in, device=gpu)
        with torch.no_grad():
            want = mlp(x.double().float())
            want = ln(want) if ln is not None else want
            got = rm.mlp_forward(x, mlp, ln)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0)


def _mask_head_case(gpu, N, T_, n_inst, L, E):
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    torch.manual_seed(N + L)
    enc = GridEncoder(input_dim=3, num_levels=L, level_dim=8, base_resolution=16, log2_hashmap_size=15, desired_resolution=512).to(gpu)
    with torch.no_grad():
        enc.embeddings.uniform_(-1.0, 1.0)
    mlp = SkipConnMLP(L * 8 + E, n_inst, 256, 3, skip_layers=[], bias=False).to(gpu)
    xyz = torch.rand(N, T_, 3, device=gpu) * 2.2 - 1.1             # some samples outside the grid's box
    extra = torch.randn(N, T_, max(E, 1), device=gpu)[..., :E].contiguous()
    w = torch.rand(N, T_, device=gpu)
    w[::7] = 0.0                                                   # rays whose samples all carry weight 0
    return enc, mlp, xyz, extra, w


@pytest.mark.parametrize("N,T_,n_inst,L,E", [(300, 32, 2, 16, 15), (37, 128, 3, 16, 15), (64, 16, 16, 16, 7), (100, 8, 2, 6, 15), (33, 4, 1, 3, 0), (4000, 32, 2, 16, 15)])
def test_mask16_kernel_vs_unfused_composition_and_reproducible(gpu, N, T_, n_inst, L, E):
    """k_mask16 (mlp16.inc, round 5: the fused mask head on v_mfma_f32_16x16x32_f16 tiles of 16 rows, two waves per SIMD, the tile ahead's
    first-layer operands made by background units under the hidden layer and staged in LDS) takes the reference's mask-head shape
    (network.py:118-123).  Against the unfused composition -- grid encoder -> cat -> wide MLP kernel -> composite (renderer.py:376-385) --
    within the split-fp16 contract, and bit-equal run to run (a version whose units' gathers stayed in flight across a chunk boundary was not:
    profiles/r05/mask16_units_ab.txt): partial ray groups, samples outside the box, level counts that are no multiple of four, few / no
    appended channels, tiles whose weights are all zero, 4000 rays (every CU busy)."""
    from sanerf_hq_amd import raymarching as rm
    enc, mlp, xyz, extra, w = _mask_head_case(gpu, N, T_, n_inst, L, E)
    outs = [rm.mask_head(w, xyz, extra, enc, mlp, 1.0).clone() for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    with torch.no_grad():
        feats = enc(xyz.reshape(-1, 3), bound=1.0)
        logits = rm.mlp_forward(torch.cat([feats, extra.reshape(N * T_, E)], dim=-1), mlp, None).reshape(N, T_, n_inst)
        want = (w.unsqueeze(-1) * logits).sum(1)
    assert torch.isfinite(outs[0]).all() and float((outs[0] - want).abs().max()) <= 5e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("N,T_,n_inst,L,E", [(300, 32, 2, 16, 15), (37, 128, 3, 16, 15), (64, 16, 16, 16, 7), (100, 8, 2, 6, 15)])
def test_mask16_kernel_agrees_with_the_round4_kernel(gpu, N, T_, n_inst, L, E, experiments_build):
    """Same-box A/B partner of k_mask16: k_mlp_wide_j<3> (rounds 3-4: 32-row tiles, one wave per SIMD), selected with sn_debug_set("mask_head16", 0)
    in experiments builds: same products, an MFMA sums 32 of them instead of 16 -- round-off agreement."""
    from sanerf_hq_amd import _lib, raymarching as rm
    enc, mlp, xyz, extra, w = _mask_head_case(gpu, N, T_, n_inst, L, E)
    b = rm.mask_head(w, xyz, extra, enc, mlp, 1.0).clone()
    try:
        _lib.check(_lib.lib().sn_debug_set(b"mask_head16", 0), "debug_set")
        a = rm.mask_head(w, xyz, extra, enc, mlp, 1.0).clone()
    finally:
        _lib.check(_lib.lib().sn_debug_set(b"mask_head16", 8), "debug_set")
    assert torch.isfinite(b).all() and float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


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
