"""Shared test plumbing: golden fixtures, parameter regeneration, oracle / product model builders."""
from __future__ import annotations

import json
import os
import types
from typing import Dict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# hash-grid constants of the reference network (nerf/network.py:93-143)
GRIDS = {
    "grid": dict(num_levels=16, level_dim=2, log2_hashmap_size=19, desired_resolution=4096),
    "prop_encoders.0": dict(num_levels=5, level_dim=2, log2_hashmap_size=17, desired_resolution=128),
    "prop_encoders.1": dict(num_levels=5, level_dim=2, log2_hashmap_size=17, desired_resolution=256),
    "s_grid": dict(num_levels=16, level_dim=8, log2_hashmap_size=19, desired_resolution=512),
    "m_grid": dict(num_levels=16, level_dim=8, log2_hashmap_size=19, desired_resolution=512),
}


def golden(name: str):
    return np.load(os.path.join(GOLD, name + ".npz"))


def spec_of(g) -> list:
    return json.loads(str(g["param_spec"]))


def params_from_spec(spec) -> Dict[str, np.ndarray]:
    from sanerf_hq_amd import synth
    return {s["name"]: synth.make_param(s) for s in spec}


def make_opt(**kw):
    opt = types.SimpleNamespace(
        bound=128, contract=True, min_near=0.2, density_thresh=10, render_mesh=False,
        num_steps=[128, 64, 32], with_mask=False, with_sam=False, n_inst=2, mask_mlp_type="default",
        background="last_sample", lambda_proposal=0.0, lambda_distort=0.0, max_ray_batch=16384,
        sam_use_view_direction=True, epsilon=1e-6, num_rays=4096)
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


def oracle_cfg(orc, params: Dict[str, np.ndarray], num_steps, heads: bool = False, table_f16: bool = False):
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
    for i, v in enumerate([-128.0] * 3 + [128.0] * 3):
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


def product_model(params: Dict[str, np.ndarray], num_steps, heads: bool, device):
    """The product's NeRFNetwork with parameters loaded by name (state_dict compatibility is part of the test)."""
    import torch
    from sanerf_hq_amd.nerf import NeRFNetwork
    opt = make_opt(num_steps=list(num_steps), with_sam=heads, with_mask=heads)
    model = NeRFNetwork(opt)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith("offsets") or m.startswith("aabb") for m in missing), missing
    return model.to(device).eval()


def synthetic_params(num_steps, heads: bool = False, seed: int = 7, table_amp: float = 1.0, gain: float = 4.0,
                     decay: float = 0.7) -> Dict[str, np.ndarray]:
    """Fresh deterministic parameters for the reference network shape (no fixture needed)."""
    import zlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    from sanerf_hq_amd import synth
    out = {}

    def tab(name):
        g = GRIDS[name]
        offs, _ = orc.grid_layout(3, g["num_levels"], g["level_dim"], 2, 16, g["log2_hashmap_size"], g["desired_resolution"])
        s = dict(name=name, shape=[int(offs[-1]), g["level_dim"]], seed=(zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF,
                 lo=-table_amp, hi=table_amp, offsets=[int(o) for o in offs],
                 level_scale=[decay ** l for l in range(len(offs) - 1)])
        out[name + ".embeddings"] = synth.make_param(s)

    def lin(name, o, i, g=gain):
        out[name] = synth.linear_weight(o, i, (zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF, g)

    tab("grid")
    for i, (o, k) in enumerate(((64, 32), (64, 64), (16, 64))):
        lin(f"grid_mlp.net.{i}.weight", o, k)
    for i, (o, k) in enumerate(((32, 31), (32, 32), (3, 32))):
        lin(f"view_mlp.net.{i}.weight", o, k)
    for p in range(2):   # NeRFNetwork always owns both proposal nets (network.py:131-143)
        tab(f"prop_encoders.{p}")
        lin(f"prop_mlp.{p}.net.0.weight", 16, 10)
        lin(f"prop_mlp.{p}.net.1.weight", 1, 16)
    if heads:
        tab("s_grid"); tab("m_grid")
        for i, (o, k) in enumerate(((256, 163), (256, 256), (256, 419), (256, 256), (256, 256))):
            lin(f"samvit_mlp.0.net.{i}.weight", o, k, 2.0)
            out[f"samvit_mlp.0.net.{i}.bias"] = synth.hash_uniform((o,), 1000 + i + seed, -0.1, 0.1)
        out["samvit_mlp.1.weight"] = synth.hash_uniform((256,), 2000 + seed, 0.5, 1.5)
        out["samvit_mlp.1.bias"] = synth.hash_uniform((256,), 2001 + seed, -0.1, 0.1)
        for i, (o, k) in enumerate(((256, 143), (256, 256), (2, 256))):
            lin(f"mask_mlp.0.net.{i}.weight", o, k, 2.0)
    return out


def camera_rays(orc, H: int, W: int, radius=1.0, elev=20.0, azim=30.0):
    from sanerf_hq_amd import synth
    pose = synth.orbit_pose(radius, elev, azim)
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    ro, rd = orc.generate_rays(pose, fx, fy, cx, cy, H, W)
    return pose, (fx, fy, cx, cy), ro, rd
