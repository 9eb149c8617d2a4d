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
