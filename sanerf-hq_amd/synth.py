"""Deterministic synthetic inputs for the hot path (bench.py, smoke(), tests, fixture generation).

There are no datasets or checkpoints on the GPU box, so every workload is built
from (a) a counter-based hash -> float32 generator that gives identical bits on
any machine and any numpy version, and (b) the pinhole / orbit camera the
reference itself uses for synthetic views (nerf/provider.py:936-938 intrinsics,
nerf/utils.py:269-287 ray convention).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_u01(n: int, seed: int, offset: int = 0) -> np.ndarray:
    """n float32 values in [0,1), value i = splitmix64(seed, offset+i) >> 40 scaled by 2^-24."""
    with np.errstate(over="ignore"):
        x = np.arange(offset, offset + n, dtype=np.uint64)
        x = x + np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def hash_uniform(shape, seed: int, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape))
    u = hash_u01(n, seed)
    return (u * np.float32(hi - lo) + np.float32(lo)).astype(np.float32).reshape(shape)


def linear_weight(fan_out: int, fan_in: int, seed: int, gain: float = 1.0) -> np.ndarray:
    """[out,in] matrix, U(-b,b) with b = gain/sqrt(fan_in) (nn.Linear's default bound when gain=1)."""
    b = gain / math.sqrt(fan_in)
    return hash_uniform((fan_out, fan_in), seed, -b, b)


def orbit_pose(radius: float = 1.0, elevation_deg: float = 20.0, azimuth_deg: float = 30.0) -> np.ndarray:
    """cam2world 4x4 (OpenGL: x right, y up, camera looks down -z) looking at the origin."""
    el, az = math.radians(elevation_deg), math.radians(azimuth_deg)
    eye = np.array([radius * math.cos(el) * math.sin(az), radius * math.sin(el), radius * math.cos(el) * math.cos(az)])
    fwd = -eye / np.linalg.norm(eye)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    up2 = np.cross(right, fwd)
    pose = np.eye(4, dtype=np.float64)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up2, -fwd, eye
    return pose.astype(np.float32)


def pinhole_intrinsics(H: int, W: int, fovy_deg: float = 60.0) -> Tuple[float, float, float, float]:
    """fx, fy, cx, cy as nerf/provider.py:936-938 builds them for synthetic views."""
    f = H / (2.0 * math.tan(math.radians(fovy_deg) / 2.0))
    return float(f), float(f), W / 2.0, H / 2.0


def make_param(spec: dict) -> np.ndarray:
    """Materialise one tensor from a fixture `param_spec` entry.

    spec = {name, shape, seed, lo, hi[, offsets, level_scale]}.  For hash tables the
    rows of level l are additionally multiplied by level_scale[l] (fine levels get
    small amplitudes, as in a trained field, which keeps the rendered image
    well-conditioned w.r.t. 1-ulp differences in sample positions)."""
    a = hash_uniform(spec["shape"], spec["seed"], spec["lo"], spec["hi"])
    if "level_scale" in spec:
        offs = spec["offsets"]
        for l, s in enumerate(spec["level_scale"]):
            a[offs[l]:offs[l + 1]] *= np.float32(s)
    return a


# ---------------------------------------------------------------------------------------------
# synthetic model of the reference network shape (bench.py, smoke(), tests)
# ---------------------------------------------------------------------------------------------
# hash-grid constants of the reference network (nerf/network.py:93-143; contract=True -> bound 2 -> desired 4096)
GRIDS = {
    "grid": dict(num_levels=16, level_dim=2, log2_hashmap_size=19, desired_resolution=4096),
    "prop_encoders.0": dict(num_levels=5, level_dim=2, log2_hashmap_size=17, desired_resolution=128),
    "prop_encoders.1": dict(num_levels=5, level_dim=2, log2_hashmap_size=17, desired_resolution=256),
    "s_grid": dict(num_levels=16, level_dim=8, log2_hashmap_size=19, desired_resolution=512),
    "m_grid": dict(num_levels=16, level_dim=8, log2_hashmap_size=19, desired_resolution=512),
}


def make_opt(**kw):
    """The `opt` namespace NeRFNetwork reads (main.py defaults; main.py:217-221 forces bound=128, contract=True)."""
    import types
    opt = types.SimpleNamespace(
        bound=128, contract=True, min_near=0.2, density_thresh=10, render_mesh=False,
        num_steps=[128, 64, 32], with_mask=False, with_sam=False, n_inst=2, mask_mlp_type="default",
        background="last_sample", lambda_proposal=0.0, lambda_distort=0.0, max_ray_batch=16384,
        sam_use_view_direction=True, epsilon=1e-6, num_rays=4096)
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


def grid_offsets(name: str, base_resolution: int = 16, input_dim: int = 3) -> np.ndarray:
    """Level offsets of one of the network's grids from the product's own layout rule (ops.grid_level_offsets)."""
    from .ops import grid_level_offsets
    g = GRIDS[name]
    scale = np.exp2(np.log2(g["desired_resolution"] / base_resolution) / (g["num_levels"] - 1))     # grid.py:107-108
    return grid_level_offsets(input_dim, g["num_levels"], scale, base_resolution, g["log2_hashmap_size"])


def synthetic_params(num_steps, heads: bool = False, seed: int = 7, table_amp: float = 1.0, gain: float = 4.0,
                     decay: float = 0.7) -> dict:
    """Fresh deterministic parameters (name -> float32 array, the reference's state_dict names) for the reference
    network shape: tables U(-amp, amp) x decay^level, nn.Linear-style weights with `gain`."""
    import zlib
    out = {}

    def tab(name):
        g = GRIDS[name]
        offs = grid_offsets(name)
        s = dict(name=name, shape=[int(offs[-1]), g["level_dim"]], seed=(zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF,
                 lo=-table_amp, hi=table_amp, offsets=[int(o) for o in offs],
                 level_scale=[decay ** l for l in range(len(offs) - 1)])
        out[name + ".embeddings"] = make_param(s)

    def lin(name, o, i, g=gain):
        out[name] = linear_weight(o, i, (zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF, g)

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
            out[f"samvit_mlp.0.net.{i}.bias"] = hash_uniform((o,), 1000 + i + seed, -0.1, 0.1)
        out["samvit_mlp.1.weight"] = hash_uniform((256,), 2000 + seed, 0.5, 1.5)
        out["samvit_mlp.1.bias"] = hash_uniform((256,), 2001 + seed, -0.1, 0.1)
        for i, (o, k) in enumerate(((256, 143), (256, 256), (2, 256))):
            lin(f"mask_mlp.0.net.{i}.weight", o, k, 2.0)
    return out


def product_model(params: dict, num_steps, heads: bool, device):
    """NeRFNetwork with parameters loaded by name (state_dict compatibility with the reference is part of the point)."""
    import torch
    from .nerf import NeRFNetwork
    opt = make_opt(num_steps=list(num_steps), with_sam=heads, with_mask=heads)
    model = NeRFNetwork(opt)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith("offsets") or m.startswith("aabb") for m in missing), missing
    return model.to(device).eval()
