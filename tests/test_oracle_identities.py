"""CPU: closed-form identities that pin the encoder arithmetic the reference ships only as CUDA
(hash grid, SH) and the build's own scalar recipes (exp, half conversion)."""
import os

import numpy as np
import pytest
import torch

import oracle as orc_mod


def test_expf_accuracy_and_specials(orc):
    x = np.concatenate([np.linspace(-104, 89, 20001), np.linspace(-1, 1, 4001), [0.0, -0.0]]).astype(np.float32)
    got = orc.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    fin = np.isfinite(ref) & (ref > 1e-37) & (ref < 3e38)
    rel = np.abs(got[fin] - ref[fin]) / ref[fin]
    assert rel.max() < 2.0e-7, rel.max()            # <= ~1.5 ulp
    assert orc.lib().orc_expf(np.float32(-np.inf)) == 0.0
    assert orc.lib().orc_expf(np.float32(np.inf)) == np.inf
    assert np.isnan(orc.lib().orc_expf(np.float32(np.nan)))
    assert orc.lib().orc_expf(np.float32(0.0)) == 1.0
    # subnormal results are produced, not flushed
    assert 0 < orc.lib().orc_expf(np.float32(-100.0)) < 1.2e-38


def test_half_conversions_roundtrip(orc):
    l = orc.lib()
    bits = np.arange(0, 65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)
    for b in list(range(0, 65536, 97)) + [0x0001, 0x03ff, 0x0400, 0x7bff, 0x7c00, 0xfc00, 0x8000]:
        f = l.orc_half_to_float(int(b))
        if np.isnan(ref[b]):
            assert np.isnan(f)
            continue
        assert f == ref[b]
        assert l.orc_float_to_half(np.float32(ref[b])) == b
    vals = (np.random.default_rng(0).standard_normal(4000) * np.exp(np.random.default_rng(1).uniform(-18, 11, 4000))).astype(np.float32)
    want = vals.astype(np.float16).view(np.uint16)
    got = np.array([l.orc_float_to_half(v) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_level_resolutions_match_survey_appendix_c(orc):
    """Kernel-side resolutions (gridencoder.cu:133) of every grid the network builds."""
    offs, pls = orc.grid_layout(3, 16, 2, 2, 16, 19, 4096)
    assert orc.level_resolutions(16, np.log2(pls), 16) == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096]
    assert int(offs[-1]) == 6299960 and list(offs[:6]) == [0, 4096, 17920, 57224, 174880, 532792]
    offs, pls = orc.grid_layout(3, 16, 8, 2, 16, 19, 512)
    assert orc.level_resolutions(16, np.log2(pls), 16) == [16, 21, 26, 32, 41, 51, 64, 81, 102, 128, 162, 204, 256, 323, 407, 512]
    assert int(offs[-1]) == 5258512
    offs, pls = orc.grid_layout(3, 5, 2, 2, 16, 17, 128)
    assert orc.level_resolutions(5, np.log2(pls), 16) == [16, 27, 46, 77, 128] and int(offs[-1]) == 383264
    offs, pls = orc.grid_layout(3, 5, 2, 2, 16, 17, 256)
    assert orc.level_resolutions(5, np.log2(pls), 16) == [16, 32, 64, 128, 256] and int(offs[-1]) == 430080


def _torch_grid_reference(x01, emb, offsets, pls, H, gridtype=0, align_corners=False):
    """Independent formulation: python-int hash (arbitrary precision, masked to 32 bit) + torch gather + autograd."""
    L = len(offsets) - 1
    S = np.float32(np.log2(pls))
    B, D = x01.shape
    outs = []
    primes = [1, 2654435761, 805459861, 3674653429, 2097192037]
    for l in range(L):
        res = int(orc_mod.lib().orc_level_resolution(l, S, H))
        size = int(offsets[l + 1] - offsets[l])
        x = x01.astype(np.float32)
        if align_corners:
            pos = x * np.float32(res - 1)
            cell = np.minimum(np.floor(pos).astype(np.int64), res - 2)
        else:
            pos = np.clip(x * np.float32(res) - np.float32(0.5), 0, res - 1).astype(np.float32)
            cell = np.floor(pos).astype(np.int64)
        frac = torch.from_numpy((pos - cell).astype(np.float32))
        stride, nd = 1, 0
        while nd < D and stride <= size:
            stride *= res; nd += 1
        hashed = gridtype == 0 and stride > size
        acc = 0
        for corner in range(1 << D):
            w = torch.ones(B)
            idx = []
            for d in range(D):
                if corner >> d & 1:
                    w = w * frac[:, d]; idx.append(np.minimum(cell[:, d] + 1, res - 1))
                else:
                    w = w * (1 - frac[:, d]); idx.append(cell[:, d])
            rows = np.zeros(B, dtype=np.int64)
            for b in range(B):
                if hashed:
                    h = 0
                    for d in range(D):
                        h ^= (int(idx[d][b]) * primes[d]) & 0xFFFFFFFF
                    rows[b] = h % size
                else:
                    v, st = 0, 1
                    for d in range(nd):
                        v = (v + int(idx[d][b]) * st) & 0xFFFFFFFF; st = (st * res) & 0xFFFFFFFF
                    rows[b] = v % size
            acc = acc + w[:, None] * emb[torch.from_numpy(rows + int(offsets[l]))]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


@pytest.mark.parametrize("cfg", [dict(L=6, C=2, log2T=12, desired=512, gridtype=0, ac=False),
                                 dict(L=4, C=4, log2T=10, desired=64, gridtype=1, ac=False),
                                 dict(L=3, C=2, log2T=14, desired=40, gridtype=0, ac=True)])
def test_grid_forward_backward_vs_independent_torch(orc, cfg):
    rng = np.random.default_rng(3)
    offs, pls = orc.grid_layout(3, cfg["L"], cfg["C"], 2, 16, cfg["log2T"], cfg["desired"])
    emb = rng.uniform(-1, 1, (int(offs[-1]), cfg["C"])).astype(np.float32)
    x = rng.uniform(0, 1, (96, 3)).astype(np.float32)
    x[0] = [0, 0, 0]; x[1] = [1, 1, 1]; x[2] = [0.5, 0.5, 0.5]
    out, _ = orc.grid_encode_forward(x, emb, offs, pls, 16, False, cfg["gridtype"], cfg["ac"])
    t_emb = torch.from_numpy(emb).requires_grad_(True)
    ref = _torch_grid_reference(x, t_emb, offs, pls, 16, cfg["gridtype"], cfg["ac"])
    np.testing.assert_allclose(out, ref.detach().numpy(), rtol=0, atol=3e-6)
    g = rng.standard_normal(out.shape).astype(np.float32)
    ref.backward(torch.from_numpy(g))
    ge, _ = orc.grid_encode_backward(g, x, emb, offs, pls, 16, None, cfg["gridtype"], cfg["ac"])
    np.testing.assert_allclose(ge, t_emb.grad.numpy(), rtol=0, atol=1e-5)


def test_grid_reproduces_trilinear_field_on_dense_levels(orc):
    """A table filled with an affine function of the vertex position is reproduced exactly by D-linear blending."""
    offs, pls = orc.grid_layout(3, 3, 1, 2, 16, 19, 40)   # all three levels dense
    res = orc.level_resolutions(3, np.log2(pls), 16)
    emb = np.zeros((int(offs[-1]), 1), np.float32)
    a = np.array([0.3, -0.2, 0.7])
    for l, r in enumerate(res):
        assert r ** 3 <= offs[l + 1] - offs[l]
        ii = np.arange(r)
        vx, vy, vz = np.meshgrid(ii, ii, ii, indexing="ij")
        rows = vx + vy * r + vz * r * r
        emb[offs[l] + rows.reshape(-1), 0] = ((vx * a[0] + vy * a[1] + vz * a[2]) / r).reshape(-1)
    x = np.random.default_rng(5).uniform(0.1, 0.9, (200, 3)).astype(np.float32)
    out, dy_dx = orc.grid_encode_forward(x, emb, offs, pls, 16, True)
    for l, r in enumerate(res):
        want = ((x * r - 0.5) @ a) / r
        np.testing.assert_allclose(out[:, l], want, rtol=0, atol=2e-5)
    # input gradient of an affine field is its slope
    dd = dy_dx.reshape(200, 3, 3, 1)
    for l in range(3):
        np.testing.assert_allclose(dd[:, l, :, 0], np.broadcast_to(a, (200, 3)), rtol=0, atol=2e-3)


def test_grid_out_of_range_and_max_level(orc):
    offs, pls = orc.grid_layout(3, 4, 2, 2, 16, 12, 64)
    emb = np.random.default_rng(1).uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)
    x = np.array([[0.5, 0.5, 0.5], [1.5, 0.5, 0.5], [0.5, -0.1, 0.5]], np.float32)
    out, _ = orc.grid_encode_forward(x, emb, offs, pls, 16)
    assert np.all(out[1:] == 0) and np.any(out[0] != 0)            # gridencoder.cu:105-130
    out2, _ = orc.grid_encode_forward(x, emb, offs, pls, 16, max_level=2)
    assert np.array_equal(out2[:, :4], out[:, :4]) and np.all(out2[:, 4:] == 0)


def test_grid_fp16_table_is_fp32_arithmetic_on_rounded_values(orc):
    offs, pls = orc.grid_layout(3, 4, 2, 2, 16, 12, 64)
    emb = np.random.default_rng(2).uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)
    x = np.random.default_rng(3).uniform(0, 1, (64, 3)).astype(np.float32)
    a, _ = orc.grid_encode_forward(x, emb.astype(np.float16), offs, pls, 16)
    b, _ = orc.grid_encode_forward(x, emb.astype(np.float16).astype(np.float32), offs, pls, 16)
    assert np.array_equal(a, b)


def test_grid_tv_and_weight_decay(orc):
    offs, pls = orc.grid_layout(3, 3, 2, 2, 16, 19, 40)
    rng = np.random.default_rng(4)
    emb = rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)
    grad = np.zeros_like(emb)
    orc.grad_weight_decay(emb, grad, offs, 0.1)
    for l in range(3):
        sl = slice(offs[l], offs[l + 1])
        np.testing.assert_allclose(grad[sl], 2 * 0.1 * emb[sl] / (offs[l + 1] - offs[l]), rtol=1e-6)
    # TV: a constant table has zero differences -> zero gradient; a random one touches only visited cells
    const = np.ones_like(emb)
    g2 = np.zeros_like(emb)
    x = rng.uniform(0, 1, (500, 3)).astype(np.float32)
    orc.grad_total_variation(x, const, g2, offs, 1e-3, pls, 16)
    assert np.all(g2 == 0)
    g3 = np.zeros_like(emb)
    orc.grad_total_variation(x, emb, g3, offs, 1e-3, pls, 16)
    assert 0 < (np.abs(g3).sum(-1) > 0).sum() <= 500 * 3 and np.isfinite(g3).all()


def _sphere_quadrature(n_theta=64, n_phi=128):
    xs, ws = np.polynomial.legendre.leggauss(n_theta)           # nodes in cos(theta)
    phi = (np.arange(n_phi) + 0.5) * 2 * np.pi / n_phi
    ct, ph = np.meshgrid(xs, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    d = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    w = np.repeat(ws, n_phi) * (2 * np.pi / n_phi)
    return d.astype(np.float32), w


def test_sh_orthonormal_and_known_values(orc):
    d, w = _sphere_quadrature()
    y, _ = orc.sh_encode_forward(d, 8)
    gram = (y.astype(np.float64) * w[:, None]).T @ y.astype(np.float64)
    np.testing.assert_allclose(gram, np.eye(64), rtol=0, atol=2e-5)
    y0, _ = orc.sh_encode_forward(np.array([[0.0, 0.0, 1.0]], np.float32), 4)
    assert abs(y0[0, 0] - 0.28209479) < 1e-7 and abs(y0[0, 2] - 0.48860251) < 1e-7
    assert abs(y0[0, 6] - (0.94617470 - 0.31539157)) < 1e-6
    for deg in range(1, 9):   # lower degrees are prefixes
        yd, _ = orc.sh_encode_forward(d[:50], deg)
        assert np.array_equal(yd, y[:50, :deg * deg])


def test_sh_gradients_vs_finite_differences(orc):
    rng = np.random.default_rng(6)
    d = rng.standard_normal((40, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    _, dy = orc.sh_encode_forward(d, 8, True)
    dy = dy.reshape(40, 3, 64)
    h = 1e-3
    for k in range(3):
        dp, dm = d.copy(), d.copy()
        dp[:, k] += h; dm[:, k] -= h
        yp, _ = orc.sh_encode_forward(dp, 8); ym, _ = orc.sh_encode_forward(dm, 8)
        fd = (yp.astype(np.float64) - ym.astype(np.float64)) / (2 * h)
        np.testing.assert_allclose(dy[:, k], fd, rtol=0, atol=5e-3)
    # the closed forms (oracle/sh_grad.inc: derived symbolically from the forward polynomials and checked coefficient by
    # coefficient against shencoder.cu:125-353 by tools/gen_oracle_sh_grad.py) against the 4th-order fp64 difference the
    # oracle used before: agreement to fp32 round-off, every degree
    for deg in range(1, 9):
        _, dyc = orc.sh_encode_forward(d, deg, True)
        np.testing.assert_allclose(dyc, orc.sh_dy_dx_fd(d, deg), rtol=5e-6, atol=1e-5)
    g = rng.standard_normal((40, 64)).astype(np.float32)
    gi = orc.sh_encode_backward(g, d, 8, dy.reshape(40, -1))
    np.testing.assert_allclose(gi, np.einsum("bc,bdc->bd", g, dy), rtol=1e-5, atol=1e-5)


def test_freq_layout_and_backward(orc):
    x = np.random.default_rng(7).uniform(-1, 1, (33, 3)).astype(np.float32)
    y = orc.freq_encode_forward(x, 5)
    assert y.shape == (33, 33) and np.array_equal(y[:, :3], x)
    np.testing.assert_allclose(y[:, 3:6], np.sin(x), atol=1e-7)      # sin f0
    np.testing.assert_allclose(y[:, 6:9], np.cos(x), atol=1e-7)      # cos f0
    np.testing.assert_allclose(y[:, 27:30], np.sin(16 * x), atol=2e-6)
    g = np.random.default_rng(8).standard_normal(y.shape).astype(np.float32)
    gi = orc.freq_encode_backward(g, y, 3, 5)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    parts = [xt] + [fn(xt * 2.0 ** f) for f in range(5) for fn in (torch.sin, torch.cos)]
    torch.cat(parts, -1).backward(torch.from_numpy(g).double())
    np.testing.assert_allclose(gi, xt.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_weights_are_a_partition_with_opaque_last_sample(orc):
    rng = np.random.default_rng(9)
    rb = np.sort(rng.uniform(0.2, 50, (64, 33)), axis=1).astype(np.float32)
    sg = np.exp(rng.uniform(-4, 4, (64, 32))).astype(np.float32)
    w = orc.weights_from_sigma(rb, sg, True)
    np.testing.assert_allclose(w.sum(1), 1.0, atol=2e-6)             # renderer.py:313-315: sum of weights = 1
    w2 = orc.weights_from_sigma(rb, sg, False)
    assert np.all(w2.sum(1) <= 1 + 1e-6) and np.array_equal(w[:, :-1], w2[:, :-1])


def test_mlp_matches_torch_linear(orc):
    rng = np.random.default_rng(10)
    ws = [rng.standard_normal(s).astype(np.float32) * 0.2 for s in ((24, 12), (24, 24), (36, 5))]
    ws[2] = rng.standard_normal((5, 36)).astype(np.float32) * 0.2      # layer 2 is a skip layer: 24 + 12 inputs
    bs = [rng.standard_normal(s).astype(np.float32) * 0.1 for s in (24, 24, 5)]
    m = orc.make_mlp(ws, bs, "leaky", (2,), dim_in=12)
    x = rng.standard_normal((50, 12)).astype(np.float32)
    y = orc.mlp_forward(m, x)
    xt = torch.from_numpy(x)
    h = torch.nn.functional.leaky_relu(xt @ torch.from_numpy(ws[0]).T + torch.from_numpy(bs[0]))
    h = torch.nn.functional.leaky_relu(h @ torch.from_numpy(ws[1]).T + torch.from_numpy(bs[1]))
    h = torch.cat([h, xt], -1) @ torch.from_numpy(ws[2]).T + torch.from_numpy(bs[2])
    np.testing.assert_allclose(y, h.numpy(), rtol=1e-5, atol=1e-5)


def test_sh_gradient_include_is_what_the_generator_emits(tmp_path, monkeypatch):
    """oracle/sh_grad.inc is generated (tools/gen_oracle_sh_grad.py: symbolic partials of the oracle's own SH polynomials); the
    committed file must be exactly what the generator produces from the committed oracle.c."""
    sympy = pytest.importorskip("sympy")
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_oracle_sh_grad", os.path.join(root, "tools", "gen_oracle_sh_grad.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(root, "oracle", "sh_grad.inc")).read()
    real_open = open

    def fake_open(path, mode="r", *a, **k):          # redirect the generator's output file, leave everything else alone
        if str(path).endswith("sh_grad.inc") and "w" in mode:
            return real_open(tmp_path / "sh_grad.inc", mode, *a, **k)
        return real_open(path, mode, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    table = gen.emit(gen.oracle_polys())
    monkeypatch.undo()
    assert len(table) == 192
    assert (tmp_path / "sh_grad.inc").read_text() == committed


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference's source text (build container only)")
def test_oracle_matches_the_reference_source_text():
    """tools/check_reference_text.py: the reference's CUDA cannot run here, so its TEXT is the pin -- the hash / dense index
    functions of gridencoder.cu:45-79 are transliterated mechanically and executed against the oracle's grid_row, and the 64
    forward SH polynomials of shencoder.cu:50-120 are compared coefficient by coefficient with the oracle's and the device's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_reference_text", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_reference_text.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check_grid_index(trials=20000) == 20000
    worst = mod.check_sh_forward()
    assert worst["oracle"] < 1e-12 and worst["device"] < 2e-7
