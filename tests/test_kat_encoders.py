"""Known-answer test of the hash-grid and SH encoders against vectors that were NOT produced by oracle/oracle.c.

The reference ships these two encoders only as CUDA, so the reference-derived fixtures (tools/gen_golden.py) had to run
the imported reference on the oracle's own encoders -- circular for the encoder arithmetic.  tests/golden/kat_encoders.npz
comes from tools/gen_kat_encoders.py: Python-integer index arithmetic + float64 blending written from
gridencoder.cu:45-201,264-348, and SH from the associated-Legendre definition.  A wrong prime, stride walk, clamp,
resolution rounding or SH sign shared by oracle.c and the HIP kernels fails here.
  * CPU: the oracle against the vectors;  * -m gpu: the HIP kernels (forward, atomic and binned backward) against them.
Limit of the pin: the vectors restate the reference's algorithm from its source text; nothing the reference EXECUTES on
this machine produced them (no nvcc, no GPU in the build container)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLD

KAT = np.load(os.path.join(GOLD, "kat_encoders.npz"))
GRIDS = ["main", "head", "prop1", "tiled_ac", "small_ac"]
FWD_TOL = 3e-6      # fp32 fmaf chain vs float64 blend of O(1) table values
BWD_TOL = 2e-5


def _case(name):
    from sanerf_hq_amd import synth
    D, L, C, log2T, base, desired, gridtype, ac, seed = [int(v) for v in KAT[f"{name}.cfg"]]
    offs = KAT[f"{name}.offsets"]
    table = synth.make_param(dict(name=name, shape=[int(offs[-1]), C], seed=seed, lo=-1.0, hi=1.0))
    return dict(D=D, L=L, C=C, log2T=log2T, base=base, desired=desired, gridtype=gridtype, ac=bool(ac), offs=offs,
                table=table, scale=float(KAT[f"{name}.scale"][0]), x=KAT[f"{name}.x"], y=KAT[f"{name}.y"],
                grad=KAT[f"{name}.grad"], rows=KAT[f"{name}.grad_rows"], gvals=KAT[f"{name}.grad_vals"], res=KAT[f"{name}.res"])


def _check_backward(ge, c):
    touched = np.zeros(ge.shape[0], bool); touched[c["rows"]] = True
    np.testing.assert_allclose(ge[c["rows"]], c["gvals"], rtol=0, atol=BWD_TOL)
    assert not ge[~touched].any(), "gradient landed in rows the independent index computation never touches"


@pytest.mark.parametrize("name", GRIDS)
def test_oracle_grid_matches_independent_vectors(orc, name):
    c = _case(name)
    offs, pls = orc.grid_layout(c["D"], c["L"], c["C"], 2, c["base"], c["log2T"], c["desired"])
    assert np.array_equal(np.asarray(offs), c["offs"]) and abs(pls - c["scale"]) < 1e-15
    assert orc.level_resolutions(c["L"], np.log2(pls), c["base"]) == [int(r) for r in c["res"]]
    y, _ = orc.grid_encode_forward(c["x"], c["table"], offs, pls, c["base"], False, c["gridtype"], c["ac"])
    np.testing.assert_allclose(y, c["y"], rtol=0, atol=FWD_TOL)
    ge, _ = orc.grid_encode_backward(c["grad"], c["x"], c["table"], offs, pls, c["base"], None, c["gridtype"], c["ac"])
    _check_backward(ge, c)


@pytest.mark.parametrize("deg", [4, 8])
def test_oracle_sh_matches_the_definition(orc, deg):
    d = KAT["sh.dirs"]
    y, _ = orc.sh_encode_forward(d, deg)
    np.testing.assert_allclose(y, KAT[f"sh.y{deg}"], rtol=0, atol=4e-6 if deg == 4 else 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRIDS)
def test_hip_grid_matches_independent_vectors(gpu, name):
    """Forward, atomic backward and sorted backward of the HIP encoder against the oracle-free vectors: main grid incl.
    level 15 (kernel resolution 4096 vs allocation resolution 4097), the F=8 head grid's levels whose kernel resolution is
    one below the allocation's, tiled + align_corners."""
    from sanerf_hq_amd import ops
    from sanerf_hq_amd.gridencoder import grid_encode
    c = _case(name)
    dev = gpu
    x = torch.from_numpy(c["x"]).to(dev)
    offs = torch.from_numpy(c["offs"].astype(np.int32)).to(dev)
    for mode in ("atomic", "binned"):
        if mode == "binned" and c["D"] not in (2, 3):
            continue
        emb = torch.from_numpy(c["table"]).to(dev).requires_grad_(True)
        old = ops.GRID_BACKWARD_MODE
        ops.GRID_BACKWARD_MODE = mode
        try:
            y = grid_encode(x, emb, offs, c["scale"], c["base"], False, c["gridtype"], c["ac"], 0)
            np.testing.assert_allclose(y.detach().cpu().numpy(), c["y"], rtol=0, atol=FWD_TOL)
            y.backward(torch.from_numpy(c["grad"]).to(dev))
        finally:
            ops.GRID_BACKWARD_MODE = old
        _check_backward(emb.grad.cpu().numpy(), c)


@pytest.mark.gpu
@pytest.mark.parametrize("deg", [4, 8])
def test_hip_sh_matches_the_definition(gpu, deg):
    from sanerf_hq_amd.shencoder import sh_encode
    d = torch.from_numpy(KAT["sh.dirs"]).to(gpu)
    y = sh_encode(d, deg, False)
    np.testing.assert_allclose(y.cpu().numpy(), KAT[f"sh.y{deg}"], rtol=0, atol=4e-6 if deg == 4 else 3e-5)


@pytest.mark.gpu
def test_hip_fused_render_uses_the_same_indices(gpu):
    """The fused renderer inlines its own copy of the index arithmetic (FinalLv fast path: 24-bit hash multiplies, pair
    rows, level offsets OR-ed into the masked term).  Its geo features through an identity-like probe: render positions
    chosen by the camera, compare xyz -> stand-alone grid_encode (pinned above) features fed through the same MLP."""
    from helpers import product_model, synthetic_params
    from sanerf_hq_amd import raymarching as rm, synth
    steps = [48]
    params = synthetic_params(steps, seed=91)
    model = product_model(params, steps, False, gpu)
    H = W = 64
    ro, rd = rm.generate_rays(synth.orbit_pose(1.1, 25.0, 70.0), synth.pinhole_intrinsics(H, W), H, W, device=gpu)
    plan = rm.RenderPlan(model, steps)
    out = rm.render_rays(plan, ro, rd, tile_w=W, want=("xyzs_last", "geo_feat_last", "sigmas"), out={})
    with torch.no_grad():
        feat = model.grid(out["xyzs_last"].reshape(-1, 3), bound=model.bound)          # stand-alone encoder on the same positions
        raw = model.grid_mlp(feat)
    np.testing.assert_allclose(out["geo_feat_last"].reshape(-1, 15).cpu().numpy(), raw[:, 1:].cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out["sigmas0"].reshape(-1).cpu().numpy(), torch.exp(raw[:, 0]).cpu().numpy(), rtol=5e-4, atol=1e-6)
