"""Shared test plumbing: golden fixtures, parameter regeneration, oracle / product model builders."""
from __future__ import annotations

import json
import os
import sys
from typing import Dict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from sanerf_hq_amd.synth import GRIDS, make_opt, product_model, synthetic_params  # noqa: E402,F401  (builders live in the package)


def golden(name: str):
    return np.load(os.path.join(GOLD, name + ".npz"))


def spec_of(g) -> list:
    return json.loads(str(g["param_spec"]))


def params_from_spec(spec, tables_f16: bool = False) -> Dict[str, np.ndarray]:
    """tables_f16: the fixture was generated with every hash table rounded to fp16 VALUES (tools/gen_golden.py fx_render(tables_f16=True))."""
    from sanerf_hq_amd import synth
    out = {s["name"]: synth.make_param(s) for s in spec}
    if tables_f16:
        out = {k: (v.astype(np.float16).astype(np.float32) if k.endswith("embeddings") else v) for k, v in out.items()}
    return out


def oracle_cfg(orc, params: Dict[str, np.ndarray], num_steps, heads: bool = False, table_f16: bool = False, aabb=None):
    """orc_render_cfg for the reference network shape from a name->array dict."""
    keep = orc._Keep()
    cfg = orc.OrcRenderCfg()
    S = len(num_steps)
    cfg.num_stages = S
    for k, t in enumerate(num_steps):
        cfg.num_steps[k] = int(t)

    def grid(name):
        g = GRIDS[name]
        offs, pls = orc.grid_layout(3, g["num_levels"], g["level_dim"], 2, 16, g["log2_hashmap_size"], g["desired_resolution"])
        emb = params[name + ".embeddings"]
        if table_f16:
            emb = emb.astype(np.float16)
        return orc.make_grid(emb, offs, pls, 16, keep=keep)

    def mlp(prefix, n, act="relu", skip=(), bias=False, dim_in=None):
        ws = [params[f"{prefix}.{i}.weight"] for i in range(n)]
        bs = [params[f"{prefix}.{i}.bias"] for i in range(n)] if bias else None
        return orc.make_mlp(ws, bs, act, skip, keep=keep, dim_in=dim_in)

    for k in range(S - 1):
        cfg.prop_grid[k] = grid(f"prop_encoders.{k}")
        cfg.prop_mlp[k] = mlp(f"prop_mlp.{k}.net", 2)
    cfg.grid = grid("grid")
    cfg.grid_mlp = mlp("grid_mlp.net", 3)
    cfg.view_mlp = mlp("view_mlp.net", 3)
    cfg.sh_degree = 4
    for i, v in enumerate(aabb if aabb is not None else [-128.0] * 3 + [128.0] * 3):
        cfg.aabb[i] = v
    cfg.min_near, cfg.bound, cfg.contract, cfg.last_sample_opaque, cfg.bg_color = 0.2, 2.0, 1, 1, 1.0
    if heads:
        cfg.with_sam = 1
        cfg.s_grid = grid("s_grid")
        cfg.samvit_mlp = mlp("samvit_mlp.0.net", 5, "leaky", (2,), bias=True, dim_in=163)
        lw = keep.hold(np.ascontiguousarray(params["samvit_mlp.1.weight"]))
        lb = keep.hold(np.ascontiguousarray(params["samvit_mlp.1.bias"]))
        cfg.ln_weight, cfg.ln_bias, cfg.ln_eps = lw.ctypes.data, lb.ctypes.data, 1e-5
        cfg.with_mask = 1
        cfg.m_grid = grid("m_grid")
        cfg.mask_mlp = mlp("mask_mlp.0.net", 3, "leaky", (), dim_in=143)
    cfg._keep = keep
    return cfg


def camera_rays(orc, H: int, W: int, radius=1.0, elev=20.0, azim=30.0):
    from sanerf_hq_amd import synth
    pose = synth.orbit_pose(radius, elev, azim)
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    ro, rd = orc.generate_rays(pose, fx, fy, cx, cy, H, W)
    return pose, (fx, fy, cx, cy), ro, rd
