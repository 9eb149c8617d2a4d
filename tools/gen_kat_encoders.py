#!/usr/bin/env python3
"""Known-answer vectors for the hash-grid and SH encoders that do NOT come from oracle/oracle.c.

tools/gen_golden.py has to inject the oracle's own grid / SH into the imported reference (the reference ships them
only as CUDA), so those fixtures compare oracle arithmetic with itself.  This generator is a second, independent
statement of the same two algorithms, written from the reference's sources with different machinery and committed
together with its output (tests/golden/kat_encoders.npz):

  hash grid   gridencoder/src/gridencoder.cu:45-79 (index), :94-201 (forward), :264-348 (backward), grid.py:121-136
              (layout): Python integers masked to 32 bits for the index, numpy float64 for the blend.
  SH          the DEFINITION of real spherical harmonics (associated Legendre recurrence, float64), not the
              polynomial literals of shencoder.cu:50-120 -- the same ordering / sign convention results.

Nothing here imports oracle/ or the package's encoders; `synth` is used only as the deterministic table generator
the tests use too.  Run: python tools/gen_kat_encoders.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sanerf_hq_amd import synth  # noqa: E402  (data generator only)

PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037)     # gridencoder.cu:49
M32 = 0xFFFFFFFF


def layout(D, L, log2T, base, desired):
    """grid.py:107-108, 121-136: per_level_scale, row offsets (float64 resolution for the allocation)."""
    scale = np.exp2(np.log2(desired / base) / (L - 1))
    offs = [0]
    for l in range(L):
        res = int(np.ceil(base * scale ** l))
        offs.append(offs[-1] + int(np.ceil(min(2 ** log2T, res ** D) / 8) * 8))
    return float(scale), offs


def kernel_res(l, scale, base):
    """gridencoder.cu:132-133 (+ grid.py:38): fp32 `ceil(exp2f(level * S) * H)`, S = log2(scale) rounded to fp32."""
    S = np.float32(np.log2(scale))
    v = np.float32(np.exp2(np.float64(np.float32(l) * S))) * np.float32(base)      # exp2 in fp64, rounded: >= 650 ulp from any integer where inexact
    return int(np.ceil(v))


def vertex_row(p, res, size, gridtype):
    """gridencoder.cu:55-79 with Python integers: dense walk while the stride fits, else XOR of coordinate * prime."""
    stride, idx, d = 1, 0, 0
    while d < len(p) and stride <= size:
        idx = (idx + p[d] * stride) & M32
        stride *= res                                   # (uint32 in the CUDA source; never wraps at these sizes)
        d += 1
    if gridtype == 0 and stride > size:
        idx = 0
        for k in range(len(p)):
            idx ^= (p[k] * PRIMES[k]) & M32
    return idx % size


def grid_forward(x01, table, offs, scale, base, gridtype=0, align_corners=False):
    """[B,D] in [0,1] -> [B, L*C] (float64 blend); also returns, per (b, l), the 2^D (row, weight) pairs for the backward."""
    B, D = x01.shape
    L, C = len(offs) - 1, table.shape[1]
    out = np.zeros((B, L * C))
    pairs = []
    for b in range(B):
        x = x01[b].astype(np.float32)
        oob = bool(np.any(x < 0) or np.any(x > 1))                                   # gridencoder.cu:105-130
        for l in range(L):
            res = kernel_res(l, scale, base)
            size = offs[l + 1] - offs[l]
            if oob:
                continue
            if align_corners:                                                         # gridencoder.cu:141-147
                pos = x * np.float32(res - 1)
                cell = np.minimum(np.floor(pos).astype(np.int64), res - 2)
            else:
                pos_f = (x.astype(np.float64) * res - 0.5).astype(np.float32)        # x * res - 0.5 contracted by nvcc into one fma: one rounding
                pos = np.minimum(np.maximum(pos_f, np.float32(0)), np.float32(res - 1))
                cell = np.floor(pos).astype(np.int64)
            frac = (pos - cell.astype(np.float32)).astype(np.float64)
            for corner in range(1 << D):
                w, p = 1.0, []
                for d in range(D):
                    if (corner >> d) & 1:
                        w *= frac[d]; p.append(min(int(cell[d]) + 1, res - 1))       # gridencoder.cu:182
                    else:
                        w *= 1.0 - frac[d]; p.append(int(cell[d]))
                row = offs[l] + vertex_row(p, res, size, gridtype)
                out[b, l * C:(l + 1) * C] += w * table[row].astype(np.float64)
                pairs.append((b, l, row, w))
    return out, pairs


def grid_backward(grad, pairs, rows_total, C):
    """gridencoder.cu:264-348: grad_embeddings[row] += w * grad[b, l*C:(l+1)*C] (float64 scatter)."""
    g = {}
    for b, l, row, w in pairs:
        g.setdefault(row, np.zeros(C))
        g[row] += w * grad[b, l * C:(l + 1) * C]
    rows = np.array(sorted(g), dtype=np.int64)
    return rows, np.stack([g[r] for r in rows])


def sh_basis(dirs, degree):
    """Real spherical harmonics Y_l^m of unit vectors, l < degree, ordered (l, m = -l..l) like shencoder.cu:50-120
    (which writes them as polynomials in x, y, z).  From the definition: associated Legendre functions P_l^m(z) by the
    standard recurrences, K_l^m normalisation, sqrt(2) * cos / sin of m * phi; Condon-Shortley phase as in the CUDA
    literals (Y_1^{-1} = -0.4886 y, Y_1^0 = 0.4886 z, Y_1^1 = -0.4886 x)."""
    out = np.zeros((dirs.shape[0], degree * degree))
    for n, (x, y, z) in enumerate(dirs.astype(np.float64)):
        phi = math.atan2(y, x)
        s = math.sqrt(max(0.0, 1.0 - z * z))
        P = {}
        for m in range(degree):
            pmm = 1.0
            for k in range(1, m + 1):
                pmm *= -(2 * k - 1) * s                                               # Condon-Shortley: (-1)^m (2m-1)!! s^m
            P[(m, m)] = pmm
            if m + 1 < degree:
                P[(m + 1, m)] = z * (2 * m + 1) * pmm
            for l in range(m + 2, degree):
                P[(l, m)] = ((2 * l - 1) * z * P[(l - 1, m)] - (l + m - 1) * P[(l - 2, m)]) / (l - m)
        for l in range(degree):
            for m in range(-l, l + 1):
                am = abs(m)
                K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - am) / math.factorial(l + am))
                if m == 0:
                    v = K * P[(l, 0)]
                elif m > 0:
                    v = math.sqrt(2.0) * K * math.cos(m * phi) * P[(l, m)]
                else:
                    v = math.sqrt(2.0) * K * math.sin(am * phi) * P[(l, am)]
                out[n, l * l + l + m] = v
    return out


def sample_points(rng, n, res_list):
    """Random points plus the special ones: cube corners, centre, exact vertex / half-cell positions of several levels."""
    pts = [rng.uniform(0, 1, (n, 3))]
    pts.append(np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.5], [0, 1, 1], [0.999999, 1e-7, 0.25]]))
    for r in res_list:
        i = rng.integers(0, r, (2, 3))
        pts.append((i + 0.5) / r)                 # exact vertex positions (frac = 0)
        pts.append((i + 1.0) / r)                 # cell centres shifted by half a cell
        pts.append(np.array([[(r - 0.5) / r, (r - 0.25) / r, 0.25 / r]]))      # inside the clamped border cells
    return np.clip(np.concatenate(pts), 0, 1).astype(np.float32)


def main():
    rng = np.random.default_rng(20260929)
    out = {}
    cases = {
        # name: (D, L, C, log2T, base, desired, gridtype, align_corners)
        "main": (3, 16, 2, 19, 16, 4096, 0, False),        # network.py:93 -- level 15: kernel res 4096, allocation res 4097
        "head": (3, 16, 8, 19, 16, 512, 0, False),         # network.py:103,120 -- levels 6, 9, 12, 15: kernel res one below the allocation's
        "prop1": (3, 5, 2, 17, 16, 256, 0, False),         # network.py:140
        "tiled_ac": (3, 4, 4, 10, 16, 64, 1, True),        # gridtype 'tiled' + align_corners
        "small_ac": (3, 3, 2, 14, 16, 40, 0, True),
    }
    for name, (D, L, C, log2T, base, desired, gridtype, ac) in cases.items():
        scale, offs = layout(D, L, log2T, base, desired)
        spec = dict(name=name, shape=[offs[-1], C], seed=4000 + len(name) * 17 + L, lo=-1.0, hi=1.0)
        table = synth.make_param(spec)
        res_list = [kernel_res(l, scale, base) for l in range(L)]
        x = sample_points(rng, 24 if L == 16 else 40, res_list[::3])
        if name == "main":
            x = np.concatenate([x, np.array([[-0.01, 0.5, 0.5], [0.5, 1.001, 0.5]], np.float32)])      # out of range -> zeros
        y, pairs = grid_forward(x, table, offs, scale, base, gridtype, ac)
        g = rng.standard_normal(y.shape)
        rows, grows = grid_backward(g, pairs, offs[-1], C)
        out.update({f"{name}.x": x, f"{name}.y": y, f"{name}.grad": g.astype(np.float32), f"{name}.grad_rows": rows,
                    f"{name}.grad_vals": grows, f"{name}.offsets": np.array(offs, np.int64), f"{name}.res": np.array(res_list, np.int64),
                    f"{name}.cfg": np.array([D, L, C, log2T, base, desired, gridtype, int(ac), spec["seed"]], np.int64),
                    f"{name}.scale": np.array([scale])})
        print(f"{name}: {x.shape[0]} points, {len(rows)} touched rows, res {res_list}")
    d = rng.standard_normal((48, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.concatenate([d, np.eye(3), -np.eye(3)]).astype(np.float32)
    d64 = d.astype(np.float64); d64 /= np.linalg.norm(d64, axis=1, keepdims=True)
    out["sh.dirs"] = d
    for deg in (4, 8):
        out[f"sh.y{deg}"] = sh_basis(d64, deg)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kat_encoders.npz"), **out)
    print("wrote tests/golden/kat_encoders.npz")


if __name__ == "__main__":
    main()
