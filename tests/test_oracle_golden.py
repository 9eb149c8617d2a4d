"""CPU: the oracle against the golden vectors captured from the reference's own Python
(tools/gen_golden.py).  This is what pins the oracle (oracle/README.md)."""
import json

import numpy as np
import pytest

from helpers import golden, oracle_cfg, params_from_spec, spec_of


def test_get_rays(orc):
    g = golden("units")
    for tag in "ab":
        fx, fy, cx, cy, H, W = g[f"rays_{tag}_intr"]
        ro, rd = orc.generate_rays(g[f"rays_{tag}_pose"], fx, fy, cx, cy, int(H), int(W))
        assert np.array_equal(ro, g[f"rays_{tag}_o"])
        np.testing.assert_allclose(rd, g[f"rays_{tag}_d"], rtol=0, atol=2e-7)   # torch bmm vs fmaf chain: 1 ulp
    # known answer from SURVEY.md §8a: 4x6 image, fovy 60, identity pose
    np.testing.assert_allclose(g["rays_a_d"][0], [-0.7217, 0.4330, -1.0], atol=1e-4)
    # pixel-subset branch of the reference (coords = (row, col), utils.py:211-212): rows of the full image
    fx, fy, cx, cy, H, W = g["rays_b_intr"]
    ro, rd = orc.generate_rays(g["rays_b_pose"], fx, fy, cx, cy, int(H), int(W))
    idx = g["rays_b_coords"][:, 0] * int(W) + g["rays_b_coords"][:, 1]
    assert np.array_equal(ro[idx], g["rays_b_sub_o"])
    np.testing.assert_allclose(rd[idx], g["rays_b_sub_d"], rtol=0, atol=2e-7)
    assert np.array_equal(g["rays_b_sub_i"], g["rays_b_coords"][:, 1]) and np.array_equal(g["rays_b_sub_j"], g["rays_b_coords"][:, 0])


def test_near_far_bit_exact(orc):
    g = golden("units")
    for tag in ("big", "small"):
        n, f = orc.near_far_from_aabb(g["nf_o"], g["nf_d"], g[f"nf_{tag}_aabb"], 0.2)
        assert np.array_equal(n, g[f"nf_{tag}_near"]) and np.array_equal(f, g[f"nf_{tag}_far"])
    assert (g["nf_big_near"] == 1e9).sum() > 0, "fixture must contain rays that miss the box"


def test_contract_bit_exact(orc):
    g = golden("units")
    assert np.array_equal(orc.contract(g["contract_x"]), g["contract_z"])


@pytest.mark.parametrize("tag,T", [("a", 65), ("b", 33), ("c", 17), ("d", 33)])
def test_sample_pdf_indices(orc, tag, T):
    """inds equal torch.searchsorted's except exact ties that torch.sum's unspecified reduction
    order creates (SURVEY.md §8a contract); every mismatch must be such a tie."""
    g = golden("units")
    bins, w, u = g[f"pdf_{tag}_bins"], g[f"pdf_{tag}_w"], g[f"pdf_{tag}_u"]
    out, inds = orc.sample_pdf(bins, w, T, u=u)
    ref_i, ref_o = g[f"pdf_{tag}_inds"], g[f"pdf_{tag}_out"]
    bad = np.argwhere(inds != ref_i)
    assert len(bad) <= 2, f"{len(bad)} index mismatches"
    for r, j in bad:
        assert abs(int(inds[r, j]) - int(ref_i[r, j])) == 1
        wr = w[r].astype(np.float64) + np.float64(np.float32(0.01))
        cdf = np.concatenate([[0.0], np.cumsum(wr / wr.sum())])
        k = min(int(inds[r, j]), int(ref_i[r, j]))
        assert abs(cdf[k] - u[j]) < 4e-7, "mismatch that is not a tie"
    np.testing.assert_allclose(out, ref_o, rtol=0, atol=1e-5)
    # the oracle's own u recipe equals torch.linspace for these sizes
    assert np.array_equal(orc.linspace(0.5 / T, 1 - 0.5 / T, T), u)


def test_linspace_tables(orc):
    g = golden("units")
    for steps in (129, 65, 33, 17):
        assert np.array_equal(orc.linspace(0, 1, steps), g[f"linspace01_{steps}"])
    for steps in (49, 97):   # torch's vectorised arange path differs by 1 ulp at a few entries
        np.testing.assert_allclose(orc.linspace(0, 1, steps), g[f"linspace01_{steps}"], rtol=0, atol=1.2e-7)


@pytest.mark.parametrize("deg", [4, 6, 10])
def test_freq_vs_freqencoder_torch(orc, deg):
    g = golden("units")
    np.testing.assert_allclose(orc.freq_encode_forward(g[f"freq{deg}_x"], deg), g[f"freq{deg}_y"], rtol=0, atol=2e-6)


def _render_case(orc, name, heads=False):
    g = golden(name)
    params = params_from_spec(spec_of(g))
    steps = [int(t) for t in g["num_steps"]]
    cfg = oracle_cfg(orc, params, steps, heads=heads)
    S = len(steps)
    u_tables = {k: g[f"u{k}"] for k in range(1, S)}
    got = orc.render(cfg, g["rays_o"], g["rays_d"], debug=True, u_tables=u_tables)
    return g, got, steps


def test_render_sref(orc):
    g, got, steps = _render_case(orc, "render_sref")
    assert np.array_equal(got["bins0"], g["bins0"])
    for k in (1, 2):
        assert np.array_equal(got[f"inds{k}"], g[f"inds{k}"]), f"sample indices of stage {k}"
        np.testing.assert_allclose(got[f"bins{k}"], g[f"bins{k}"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got["weights0"], g["weights0"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got["weights1"], g["weights1"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got["image"], g["image"], rtol=0, atol=1e-4)          # north_star tolerance
    np.testing.assert_allclose(got["depth"], g["depth"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(got["weights_sum"], g["weights_sum"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["render_flat128_h", "render_sref_h"])
def test_render_with_fp16_valued_tables(orc, name):
    """The reference's own render (imported Python, fp32 arithmetic) on tables whose values are fp16-representable -- what a half-precision
    table copy holds, the storage BASELINE configs[1] names -- against the oracle in BOTH table modes: fp32 storage of the rounded values and
    fp16 storage (the oracle widens rows on the fly): same numbers."""
    g = golden(name)
    assert int(g["tables_f16"]) == 1
    params = params_from_spec(spec_of(g), tables_f16=True)
    steps = [int(t) for t in g["num_steps"]]
    u_tables = {k: g[f"u{k}"] for k in range(1, len(steps))}
    for f16 in (False, True):
        got = orc.render(oracle_cfg(orc, params, steps, table_f16=f16), g["rays_o"], g["rays_d"], debug=True, u_tables=u_tables)
        for k in range(1, len(steps)):
            assert np.array_equal(got[f"inds{k}"], g[f"inds{k}"]), f"sample indices of stage {k}"
        np.testing.assert_allclose(got["image"], g["image"], rtol=0, atol=1e-4)          # north_star tolerance
        np.testing.assert_allclose(got["depth"], g["depth"], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(got["weights_sum"], g["weights_sum"], rtol=0, atol=1e-6)


def test_render_flat128(orc):
    g, got, _ = _render_case(orc, "render_flat128")
    np.testing.assert_allclose(got["sigmas0"], g["sigmas0"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(got["image"], g["image"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got["depth"], g["depth"], rtol=1e-6, atol=2e-6)


def test_render_heads(orc):
    g, got, _ = _render_case(orc, "render_heads", heads=True)
    np.testing.assert_allclose(got["image"], g["image"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(got["samvit"], g["samvit"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(got["instance_mask_logits"], g["instance_mask_logits"], rtol=0, atol=1e-4)


def test_render_c1(orc):
    """BASELINE configs[0]: 64x64, hashgrid L=8 T=2^14, 1-hidden-x32 MLP, 32 samples/ray (CPU plumbing)."""
    from sanerf_hq_amd import synth
    g = golden("render_c1")
    params = params_from_spec(spec_of(g))
    H, W = [int(v) for v in g["HW"]]
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    ro, rd = orc.generate_rays(g["pose"], fx, fy, cx, cy, H, W)
    keep = orc._Keep()
    cfg = orc.OrcRenderCfg()
    cfg.num_stages = 1
    cfg.num_steps[0] = 32
    offs, pls = orc.grid_layout(3, 8, 2, 2, 16, 14, 2048)
    cfg.grid = orc.make_grid(params["grid.embeddings"], offs, pls, 16, keep=keep)
    cfg.grid_mlp = orc.make_mlp([params["grid_mlp.net.0.weight"], params["grid_mlp.net.1.weight"]], keep=keep)
    cfg.view_mlp = orc.make_mlp([params["view_mlp.net.0.weight"], params["view_mlp.net.1.weight"]], keep=keep)
    cfg.sh_degree = 4
    for i, v in enumerate([-128.0] * 3 + [128.0] * 3):
        cfg.aabb[i] = v
    cfg.min_near, cfg.bound, cfg.contract, cfg.last_sample_opaque, cfg.bg_color = 0.2, 2.0, 1, 1, 1.0
    got = orc.render(cfg, ro, rd)   # rays from the oracle's own generate_rays (1 ulp from torch's bmm)
    for k in ("image", "depth", "weights_sum"):
        np.testing.assert_allclose(got[k], g[k], rtol=0, atol=1e-5)


def test_train_fixture_forward(orc):
    """config C5 forward: mask logits of the oracle vs the reference's; loss restated from trainer.py:419-428."""
    g = golden("train_c5")
    params = params_from_spec(spec_of(g))
    cfg = oracle_cfg(orc, {**params, **{k: v for k, v in _dummy_sam().items() if k not in params}}, [128, 64, 32], heads=True)
    cfg.with_sam = 0
    got = orc.render(cfg, g["rays_o"], g["rays_d"])
    np.testing.assert_allclose(got["instance_mask_logits"], g["logits"], rtol=0, atol=1e-4)
    z = got["instance_mask_logits"].astype(np.float64)
    p = np.exp(z - z.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    p = np.clip(p, float(g["epsilon"]), 1 - float(g["epsilon"]))
    loss = -np.log(p[np.arange(len(p)), g["labels"]]).mean()
    assert abs(loss - float(g["loss"])) < 1e-5


def _dummy_sam():
    """s_grid / samvit placeholders so oracle_cfg(heads=True) can be reused with with_sam switched off."""
    from helpers import GRIDS
    import oracle as orc
    g = GRIDS["s_grid"]
    offs, _ = orc.grid_layout(3, g["num_levels"], g["level_dim"], 2, 16, g["log2_hashmap_size"], g["desired_resolution"])
    d = {"s_grid.embeddings": np.zeros((int(offs[-1]), 8), np.float32)}
    for i, (o, k) in enumerate(((256, 163), (256, 256), (256, 419), (256, 256), (256, 256))):
        d[f"samvit_mlp.0.net.{i}.weight"] = np.zeros((o, k), np.float32)
        d[f"samvit_mlp.0.net.{i}.bias"] = np.zeros((o,), np.float32)
    d["samvit_mlp.1.weight"] = np.ones(256, np.float32)
    d["samvit_mlp.1.bias"] = np.zeros(256, np.float32)
    return d


def test_fixture_reports_are_within_contract():
    """The numbers gen_golden.py measured (oracle vs reference) when the fixtures were written."""
    for name, keys in (("render_sref", ("image", "depth", "weights_sum")), ("render_flat128", ("image",)),
                       ("render_heads", ("image", "samvit", "mask")), ("render_c1", ("image", "depth"))):
        stats = json.loads(str(golden(name)["oracle_vs_reference"]))
        for k in keys:
            assert stats[k] <= 1e-4, (name, k, stats[k])
        for k, v in stats.items():
            if k.endswith("_mismatch"):
                assert v <= 2, (name, k, v)
