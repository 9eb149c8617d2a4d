"""Encoder factory: `get_encoder(name, ...) -> (encoder, output_dim)`.

Keeps the reference's names and return convention (encoding.py:47-79): 'None', 'frequency_torch',
'frequency', 'sh', 'hashgrid', 'tiledgrid'.  Built as a small registry of constructors; the HIP-backed
encoders live in `ops.py`.
"""
from typing import Callable, Dict

import torch
import torch.nn as nn


class FreqEncoder_torch(nn.Module):
    """Pure-torch positional encoding behind the 'frequency_torch' key: [x, sin(f0 x), cos(f0 x), sin(f1 x), ...]
    with log-spaced bands 2^0 .. 2^max_freq_log2 (same signature and output order as reference encoding.py:6-44)."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True,
                 periodic_fns=(torch.sin, torch.cos)):
        super().__init__()
        self.input_dim, self.include_input, self.periodic_fns = input_dim, include_input, tuple(periodic_fns)
        if log_sampling:
            bands = torch.logspace(0.0, float(max_freq_log2), N_freqs, base=2.0)
        else:
            bands = torch.linspace(1.0, 2.0 ** max_freq_log2, N_freqs)
        self.freq_bands = [float(b) for b in bands]
        self.output_dim = input_dim * (int(include_input) + N_freqs * len(self.periodic_fns))

    def forward(self, input, **kwargs):
        terms = [fn(input * band) for band in self.freq_bands for fn in self.periodic_fns]
        return torch.cat(([input] if self.include_input else []) + terms, dim=-1)


def _make_grid(gridtype: str):
    def build(input_dim, num_levels, level_dim, base_resolution, log2_hashmap_size, desired_resolution,
              align_corners, interpolation, **_):
        from .ops import GridEncoder
        return GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim,
                           base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                           desired_resolution=desired_resolution, gridtype=gridtype,
                           align_corners=align_corners, interpolation=interpolation)
    return build


def _make_sh(input_dim, degree, **_):
    from .ops import SHEncoder
    return SHEncoder(input_dim=input_dim, degree=degree)


def _make_freq(input_dim, multires, **_):
    from .ops import FreqEncoder
    return FreqEncoder(input_dim=input_dim, degree=multires)


def _make_freq_torch(input_dim, multires, **_):
    return FreqEncoder_torch(input_dim=input_dim, max_freq_log2=multires - 1, N_freqs=multires, log_sampling=True)


_REGISTRY: Dict[str, Callable] = {
    "frequency_torch": _make_freq_torch,
    "frequency": _make_freq,
    "sh": _make_sh,
    "hashgrid": _make_grid("hash"),
    "tiledgrid": _make_grid("tiled"),
}


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, interpolation="linear", **kwargs):
    if encoding == "None":                       # identity "encoder": a plain callable, not a module
        return (lambda x, **kw: x), input_dim
    try:
        build = _REGISTRY[encoding]
    except KeyError:
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency, sh, hashgrid, tiledgrid]") from None
    encoder = build(input_dim=input_dim, multires=multires, degree=degree, num_levels=num_levels, level_dim=level_dim,
                    base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                    desired_resolution=desired_resolution, align_corners=align_corners, interpolation=interpolation)
    return encoder, encoder.output_dim
