"""Radiance field + heads with the reference's module layout (nerf/network.py:85-231).

Sub-module names and tensor shapes are the reference's, so its checkpoints load unchanged:
`grid.{embeddings,offsets}`, `grid_mlp.net.{0,1,2}.weight`, `view_mlp.net.*`, `prop_encoders.{0,1}.*`,
`prop_mlp.{0,1}.net.*`, `s_grid.*`, `samvit_mlp.0.net.*`, `samvit_mlp.1.*`, `m_grid.*`, `mask_mlp.0.net.*`,
buffers `aabb_train` / `aabb_infer` (checked against the reference's own state_dict in the tests).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..activation import trunc_exp
from ..encoding import get_encoder
from ..ops import (ACT_LEAKY, ACT_NONE, ACT_RELU, SMALL_ACT_SIGMOID_BG, SMALL_ACT_TRUNC_EXP0, small_linear, small_mlp_fusable, small_mlp_train,
                   wide_mlp_fusable, wide_mlp_train)
from .renderer import NeRFRenderer

# constants the reference hard-codes in NeRFNetwork.__init__ (network.py:90-143)
GEOM_FEAT_DIM = 15
MAIN_GRID = dict(level_dim=2, num_levels=16, log2_hashmap_size=19)               # desired_resolution = 2048 * bound
HEAD_GRID = dict(num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=19, desired_resolution=512)
LIGHT_MASK_GRID = dict(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=10, desired_resolution=256)
PROPOSAL_RESOLUTIONS = (128, 256)
PROPOSAL_GRID = dict(level_dim=2, num_levels=5, log2_hashmap_size=17)


def _chain(widths: Sequence[int], bias: bool) -> nn.ModuleList:
    return nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(widths[:-1], widths[1:]))


class MLP(nn.Module):
    """ReLU perceptron: `num_layers` linear maps, activation between them (network.py:9-29)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = _chain([dim_in] + [dim_hidden] * (num_layers - 1) + [dim_out], bias)

    def forward(self, x):
        if small_mlp_fusable(x, list(self.net)):                 # training, one of the reference network's shapes: one kernel per direction
            return small_mlp_train(x, list(self.net))[0]
        *hidden, last = self.net
        for layer in hidden:
            x = small_linear(x, layer, ACT_RELU)
        return small_linear(x, last)

    def forward_trunc_exp(self, x):
        """(raw, trunc_exp(raw[..., 0])) -- network.py:155,179; folded into the kernel when the fused training route applies."""
        if small_mlp_fusable(x, list(self.net)):
            return small_mlp_train(x, list(self.net), SMALL_ACT_TRUNC_EXP0)
        raw = self.forward(x)
        return raw, trunc_exp(raw[..., 0])

    def forward_sigmoid_bg(self, x, weights_sum, bg: float):
        """sigmoid(self(x)) + (1 - weights_sum)[..., None] * bg -- renderer.py:349-353; one kernel when the fused training route applies."""
        if small_mlp_fusable(x, list(self.net)) and self.net[-1].weight.shape[0] <= 4:
            return small_mlp_train(x, list(self.net), SMALL_ACT_SIGMOID_BG, weights_sum, float(bg))[1]
        return torch.sigmoid(self.forward(x)) + (1 - weights_sum).unsqueeze(-1) * bg


class SkipConnMLP(nn.Module):
    """LeakyReLU perceptron; layers listed in `skip_layers` see cat([h, x_in]) (network.py:31-66)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, skip_layers=[], bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.skip_layers = skip_layers
        fan_in = [dim_in if i == 0 else dim_hidden + (dim_in if i in skip_layers else 0) for i in range(num_layers)]
        fan_out = [dim_hidden] * (num_layers - 1) + [dim_out]
        self.net = nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(fan_in, fan_out))

    def forward(self, x):
        if wide_mlp_fusable(x, list(self.net), self.skip_layers):           # training: one kernel for the whole backward data path
            return wide_mlp_train(x, list(self.net), leaky=True)
        h = x
        for i, layer in enumerate(self.net):
            if i in self.skip_layers:
                h = torch.cat([h, x], dim=-1)
            h = small_linear(h, layer, ACT_LEAKY if i != self.num_layers - 1 else ACT_NONE)
        return h


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt):
        super().__init__(opt)
        self.geom_feat_dim = GEOM_FEAT_DIM
        grid = lambda **kw: get_encoder("hashgrid", input_dim=3, **kw)          # noqa: E731

        self.grid, self.grid_in_dim = grid(desired_resolution=2048 * self.bound, **MAIN_GRID)
        self.grid_mlp = MLP(self.grid_in_dim, 1 + GEOM_FEAT_DIM, 64, 3, bias=False)
        self.view_encoder, self.view_in_dim = get_encoder("sh", input_dim=3, degree=4)
        self.view_mlp = MLP(GEOM_FEAT_DIM + self.view_in_dim, 3, 32, 3, bias=False)

        if opt.with_sam:       # SAM-feature head: 128 grid features + colour features + rgb + depth -> 256 (network.py:101-116)
            self.s_grid, self.s_dim = grid(**HEAD_GRID)
            self.samvit_mlp_input_dim = self.s_dim + GEOM_FEAT_DIM + 4 + (self.view_in_dim if opt.sam_use_view_direction else 0)
            width = 256
            self.samvit_mlp = nn.Sequential(
                SkipConnMLP(self.s_dim + GEOM_FEAT_DIM + self.view_in_dim + 4, width, width, 5, skip_layers=[2], bias=True),
                nn.LayerNorm(width))

        if opt.with_mask:      # per-sample instance logits (network.py:118-128)
            if opt.mask_mlp_type == "default":
                self.m_grid, self.m_dim = grid(**HEAD_GRID)
                self.mask_mlp = nn.Sequential(SkipConnMLP(self.m_dim + GEOM_FEAT_DIM, opt.n_inst, 256, 3, skip_layers=[], bias=False))
            elif opt.mask_mlp_type == "lightweight_mask":
                self.m_grid, self.m_dim = grid(**LIGHT_MASK_GRID)
                self.mask_mlp = MLP(GEOM_FEAT_DIM + self.view_in_dim + 4, opt.n_inst, 64, 3, bias=False)

        self.prop_encoders = nn.ModuleList()   # two proposal networks (network.py:131-143)
        self.prop_mlp = nn.ModuleList()
        for res in PROPOSAL_RESOLUTIONS:
            enc, enc_dim = grid(desired_resolution=res, **PROPOSAL_GRID)
            self.prop_encoders.append(enc)
            self.prop_mlp.append(MLP(enc_dim, 1, 16, 2, bias=False))

    # ---- field queries (network.py:146-186) ----
    def common_forward(self, x):
        grid_output = self.grid(x, bound=self.bound)
        raw, sigma = self.grid_mlp.forward_trunc_exp(grid_output)
        return sigma, raw[..., 1:], grid_output

    # ---- the same queries on unit-cube coordinates (raymarching.sample_positions(grid_bound=self.bound)): what the renderer's training
    # ---- path calls when this class's own forward() / density() are in effect (NeRFRenderer._unit_field_ok) ----
    def field_unit(self, x01):
        """(sigma [..], raw [.., 16]) of network.py:146-170 without the colour concatenation: raw = [sigma_raw | geo_feat]."""
        raw, sigma = self.grid_mlp.forward_trunc_exp(self.grid.forward_unit(x01))
        return sigma, raw

    def density_unit(self, x01, proposal):
        raw, sigma = self.prop_mlp[proposal].forward_trunc_exp(self.prop_encoders[proposal].forward_unit(x01))
        return sigma

    def forward(self, x, d, **kwargs) -> Dict[str, torch.Tensor]:
        sigma, geo, grid_output = self.common_forward(x)
        return dict(sigma=sigma, geo_feat=geo, color=torch.cat([geo, self.view_encoder(d)], dim=-1), grid_output=grid_output)

    def density(self, x, proposal=-1):
        if 0 <= proposal < len(self.prop_encoders):
            raw, sigma = self.prop_mlp[proposal].forward_trunc_exp(self.prop_encoders[proposal](x, bound=self.bound))
            return dict(sigma=sigma, geo_feat=None)
        sigma, geo, _ = self.common_forward(x)
        return dict(sigma=sigma, geo_feat=geo)

    # ---- regularisers act on the grid being trained in the current mode (network.py:189-203) ----
    def _trained_grid(self):
        return self.s_grid if self.opt.with_sam else (self.m_grid if self.opt.with_mask else self.grid)

    def apply_total_variation(self, w):
        self._trained_grid().grad_total_variation(w)

    def apply_weight_decay(self, w):
        self._trained_grid().grad_weight_decay(w)

    def get_params(self, lr):
        owners = [self.grid, self.grid_mlp, self.view_mlp, self.prop_encoders, self.prop_mlp]
        owners += [self.s_grid, self.samvit_mlp] if self.opt.with_sam else []
        owners += [self.m_grid, self.mask_mlp] if self.opt.with_mask else []
        return [dict(params=o.parameters(), lr=lr) for o in owners]
