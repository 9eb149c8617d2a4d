"""GPU tests of the grid encoder beyond tests/test_gpu_ops.py: every (D, C) instantiation of the reference, the binned gradient scatter at its extremes (split bins, one row, large batches beyond the slab bound)."""
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


@pytest.mark.parametrize("D", [2, 3, 4, 5])
@pytest.mark.parametrize("C", [1, 2, 4, 8, 16, 32])
def test_every_reference_grid_instantiation(gpu, orc, D, C):
    """gridencoder.cu:385-411 instantiates D in {2..5} x C in {1,2,4,8,16,32}; all 24 exist here (rounds 1-3 had 13).  Forward bit-equal to
    the oracle (same fmaf chain), dy_dx / input gradient and the table gradient (atomic kernel; binned kernels where D <= 3) within tolerance."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.gridencoder import grid_encode
    rng = np.random.default_rng(100 * D + C)
    L, B = 3, 777
    offs, pls = orc.grid_layout(D, L, C, 2, 4, 11, 40)
    emb = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0; x[1] = 1; x[2] = 0.5; x[3, 0] = 1.25; x[4, 1] = -0.01
    want, want_dd = orc.grid_encode_forward(x, emb, offs, pls, 4, True)
    g = rng.standard_normal(want.shape).astype(np.float32)
    ge, gi = orc.grid_encode_backward(g, x, emb, offs, pls, 4, want_dd)
    old = ops.GRID_BACKWARD_MODE
    try:
        for mode in ("atomic", "binned"):
            ops.GRID_BACKWARD_MODE = mode
            xt, et = T(x, gpu).requires_grad_(mode == "atomic"), T(emb, gpu).requires_grad_(True)
            got = grid_encode(xt, et, T(offs, gpu), pls, 4, mode == "atomic")
            assert np.array_equal(got.detach().cpu().numpy(), want), (mode, "forward")
            got.backward(T(g, gpu))
            scale = np.abs(ge).max()
            np.testing.assert_allclose(et.grad.cpu().numpy(), ge, rtol=1e-4, atol=1e-5 * scale, err_msg=mode)
            if mode == "atomic":
                np.testing.assert_allclose(xt.grad.cpu().numpy(), gi, rtol=1e-4, atol=1e-4 * np.abs(gi).max())
    finally:
        ops.GRID_BACKWARD_MODE = old


@pytest.mark.parametrize("D,C,L,log2T,desired,B,concentrate", [(3, 8, 16, 19, 512, 40000, 0.0), (3, 2, 16, 19, 4096, 300000, 0.0), (3, 2, 5, 17, 128, 200000, 0.9),
                                                               (2, 4, 6, 12, 256, 50000, 0.5), (3, 1, 4, 8, 32, 20000, 1.0), (3, 32, 3, 10, 64, 3000, 0.0),
                                                               (2, 16, 2, 6, 16, 100000, 0.95)])
def test_binned_grid_backward_split_bins_and_extremes(gpu, orc, D, C, L, log2T, desired, B, concentrate):
    """The binned scatter (grid_binned.hip) against the atomic kernel and the oracle where the bin sizing is stressed: a fraction of the
    samples piled onto one spot (bins far over the per-item capacity are split into several work items whose partial sums meet in slabs),
    levels of a few rows (one row per bin, 256 threads per row), C = 32 (504 entries per item), D = 2, out-of-range samples."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.gridencoder import grid_encode
    rng = np.random.default_rng(7 * D + C + L)
    offs, pls = orc.grid_layout(D, L, C, 2, 16 if desired >= 32 else 4, log2T, desired)
    base = 16 if desired >= 32 else 4
    emb = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    k = int(B * concentrate)
    x[:k] = np.clip(rng.uniform(0.3, 0.7, (1, D)) + rng.normal(0, 2e-3, (k, D)), 0, 1).astype(np.float32)
    x[-3:] = 1.5                                                            # out of range: no gradient (gridencoder.cu:290)
    g = rng.standard_normal((B, L * C)).astype(np.float32)
    res = {}
    old = ops.GRID_BACKWARD_MODE
    try:
        for mode in ("binned", "atomic"):
            ops.GRID_BACKWARD_MODE = mode
            et = T(emb, gpu).requires_grad_(True)
            grid_encode(T(x, gpu), et, T(offs, gpu), pls, base, False).backward(T(g, gpu))
            res[mode] = et.grad.clone()
    finally:
        ops.GRID_BACKWARD_MODE = old
    ref = res["atomic"].double()
    rel = float((res["binned"].double() - ref).norm() / ref.norm())
    assert rel < (2e-6 if concentrate == 0.0 else 5e-5), rel      # (1e5 fp32 addends on one row: either summation order is ~1e-5 off the exact sum)
    assert torch.equal(res["binned"].abs().sum(-1) > 0, res["atomic"].abs().sum(-1) > 0)
    if B * L <= 400000:
        want, _ = orc.grid_encode_backward(g, x, emb, offs, pls, base)
        np.testing.assert_allclose(res["binned"].cpu().numpy(), want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


@pytest.mark.parametrize("C,B,concentrate", [(8, 1 << 21, 0.0), (8, 1 << 21, 0.5), (2, 3 << 20, 0.0)])
def test_binned_grid_backward_large_batches(gpu, orc, C, B, concentrate):
    """grid_binned.hip at B >= 2 M samples of the heads' grid (L = 16, T = 2^19, desired 512: network.py:104).  The 4096-bins-per-level
    cap raises the rows per bin there, split bins outnumber the round-4 bound (2 x the table) and k_bin_accum wrote past the slab region
    (advisor, round 4).  The region is now sized from the proven worst case (bin_geometry: entries / E_CAP + min(bins, entries / E_CAP)
    slabs per level); binned == atomic within the summation-order tolerance, same set of touched rows, first element not poisoned."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.gridencoder import grid_encode
    L = 16
    offs, pls = orc.grid_layout(3, L, C, 2, 16, 19, 512)
    gen = torch.Generator(device=gpu).manual_seed(11 + C)
    emb = (torch.rand(int(offs[-1]), C, device=gpu, generator=gen) * 2 - 1)
    x = torch.rand(B, 3, device=gpu, generator=gen)
    k = int(B * concentrate)
    if k:
        x[:k] = (torch.tensor([[0.41, 0.57, 0.33]], device=gpu) + 2e-3 * torch.randn(k, 3, device=gpu, generator=gen)).clamp_(0, 1)
    g = torch.randn(B, L * C, device=gpu, generator=gen)
    res = {}
    old = ops.GRID_BACKWARD_MODE
    try:
        for mode in ("binned", "atomic"):
            ops.GRID_BACKWARD_MODE = mode
            et = emb.clone().requires_grad_(True)
            grid_encode(x, et, T(offs, gpu), pls, 16, False).backward(g)
            res[mode] = et.grad
            del et
    finally:
        ops.GRID_BACKWARD_MODE = old
    assert bool(torch.isfinite(res["binned"]).all())
    ref = res["atomic"].double()
    rel = float((res["binned"].double() - ref).norm() / ref.norm())
    assert rel < (2e-6 if concentrate == 0.0 else 5e-5), rel
    assert torch.equal(res["binned"].abs().sum(-1) > 0, res["atomic"].abs().sum(-1) > 0)
