"""GPU tests added in round 4: workgroups as four consecutive 8x8 wave tiles (any band shape), the hand-written grid-backward sort,
the remaining (D, C) grid instantiations, kernel choice through sn_render_cfg fields."""
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
    monkeypatch.setenv("SN_FINAL_SP_MAX", "0")       # (small linear-order batches would take the several-lanes-per-ray kernels: per-sample form)
    monkeypatch.setenv("SN_PROP_SP_MAX", "0")
    plain_t = {k: v.clone() for k, v in rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=W).items()}      # the default (linear-tail) kernel
    plain_l = rm.render_rays(plan, T(ro, gpu), T(rd, gpu), tile_w=0, out={})
    for k in plain_t:
        assert torch.equal(plain_t[k], plain_l[k]), k
    want = orc.render(oracle_cfg(orc, params, steps, table_f16=f16), ro, rd, debug=True)
    for k in range(1, len(steps)):
        assert np.array_equal(tiled[f"inds{k}"].cpu().numpy(), want[f"inds{k}"])
    np.testing.assert_allclose(plain_t["image"].cpu().numpy(), want["image"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(plain_t["depth"].cpu().numpy(), want["depth"], rtol=1e-5, atol=1e-5)


def test_wave_tile_workgroups_feature_stage_and_compaction(gpu, orc):
    """The feature stage and the compacting final stage share the lane -> ray mapping with the stages in front of them (scratch columns):
    odd band shape, tile order == linear order."""
    from sanerf_hq_amd import raymarching as rm
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
