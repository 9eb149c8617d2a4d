"""Radiance field + heads with the reference's module layout (nerf/network.py:85-231).

Sub-module names and shapes follow the reference exactly so its checkpoints load
(state_dict keys `grid.embeddings`, `grid.offsets`, `grid_mlp.net.{0,1,2}.weight`,
`view_mlp.net.*`, `prop_encoders.{0,1}.*`, `prop_mlp.{0,1}.net.*`, `s_grid.*`,
`samvit_mlp.0.net.*`, `samvit_mlp.1.*`, `m_grid.*`, `mask_mlp.0.net.*`, `aabb_*`).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..activation import trunc_exp
from ..encoding import get_encoder
from .renderer import NeRFRenderer


class MLP(nn.Module):
    """Linear -> ReLU -> ... -> Linear (network.py:9-29)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        widths = [dim_in] + [dim_hidden] * (num_layers - 1) + [dim_out]
        self.net = nn.ModuleList(nn.Linear(widths[i], widths[i + 1], bias=bias) for i in range(num_layers))

    def forward(self, x):
        for i, layer in enumerate(self.net):
            x = layer(x)
            if i + 1 < self.num_layers:
                x = F.relu(x, inplace=True)
        return x


class SkipConnMLP(nn.Module):
    """LeakyReLU MLP whose `skip_layers` take cat([h, x_in]) (network.py:31-66)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, skip_layers=[], bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.skip_layers = skip_layers
        layers = []
        for i in range(num_layers):
            fan_in = dim_in if i == 0 else dim_hidden + (dim_in if i in skip_layers else 0)
            fan_out = dim_out if i == num_layers - 1 else dim_hidden
            layers.append(nn.Linear(fan_in, fan_out, bias=bias))
        self.net = nn.ModuleList(layers)

    def forward(self, x):
        x_in = x
        for i, layer in enumerate(self.net):
            if i in self.skip_layers:
                x = torch.cat([x, x_in], dim=-1)
            x = layer(x)
            if i + 1 < self.num_layers:
                x = F.leaky_relu(x, inplace=True)
        return x


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt):
        super().__init__(opt)
        self.geom_feat_dim = 15
        # main field (network.py:93-94)
        self.grid, self.grid_in_dim = get_encoder("hashgrid", input_dim=3, level_dim=2, num_levels=16,
                                                  log2_hashmap_size=19, desired_resolution=2048 * self.bound)
        self.grid_mlp = MLP(self.grid_in_dim, 1 + self.geom_feat_dim, 64, 3, bias=False)
        # view dependence (network.py:97-98)
        self.view_encoder, self.view_in_dim = get_encoder("sh", input_dim=3, degree=4)
        self.view_mlp = MLP(self.geom_feat_dim + self.view_in_dim, 3, 32, 3, bias=False)
        # SAM feature head (network.py:101-116)
        if opt.with_sam:
            self.s_grid, self.s_dim = get_encoder("hashgrid", input_dim=3, num_levels=16, level_dim=8, base_resolution=16,
                                                  log2_hashmap_size=19, desired_resolution=512)
            self.samvit_mlp_input_dim = self.s_dim + self.geom_feat_dim + 4
            if opt.sam_use_view_direction:
                self.samvit_mlp_input_dim += self.view_in_dim
            out_dim = 256
            self.samvit_mlp = nn.Sequential(
                SkipConnMLP(self.s_dim + self.geom_feat_dim + self.view_in_dim + 4, out_dim, out_dim, 5, skip_layers=[2], bias=True),
                nn.LayerNorm(out_dim),
            )
        # mask head (network.py:118-128)
        if opt.with_mask:
            if opt.mask_mlp_type == "default":
                self.m_grid, self.m_dim = get_encoder("hashgrid", input_dim=3, num_levels=16, level_dim=8, base_resolution=16,
                                                      log2_hashmap_size=19, desired_resolution=512)
                self.mask_mlp = nn.Sequential(SkipConnMLP(self.m_dim + self.geom_feat_dim, opt.n_inst, 256, 3,
                                                          skip_layers=[], bias=False))
            elif opt.mask_mlp_type == "lightweight_mask":
                self.m_grid, self.m_dim = get_encoder("hashgrid", input_dim=3, num_levels=16, level_dim=2, base_resolution=16,
                                                      log2_hashmap_size=10, desired_resolution=256)
                self.mask_mlp = MLP(self.geom_feat_dim + self.view_in_dim + 4, opt.n_inst, 64, 3, bias=False)
        # two proposal networks (network.py:131-143)
        self.prop_encoders = nn.ModuleList()
        self.prop_mlp = nn.ModuleList()
        for res in (128, 256):
            enc, enc_dim = get_encoder("hashgrid", input_dim=3, level_dim=2, num_levels=5, log2_hashmap_size=17,
                                       desired_resolution=res)
            self.prop_encoders.append(enc)
            self.prop_mlp.append(MLP(enc_dim, 1, 16, 2, bias=False))

    def common_forward(self, x):
        grid_output = self.grid(x, bound=self.bound)
        f = self.grid_mlp(grid_output)
        return trunc_exp(f[..., 0]), f[..., 1:], grid_output

    def forward(self, x, d, **kwargs):
        sigma, feat, grid_output = self.common_forward(x)
        return {"sigma": sigma, "geo_feat": feat, "color": torch.cat([feat, self.view_encoder(d)], dim=-1),
                "grid_output": grid_output}

    def density(self, x, proposal=-1):
        if 0 <= proposal < len(self.prop_encoders):
            raw = self.prop_mlp[proposal](self.prop_encoders[proposal](x, bound=self.bound))
            return {"sigma": trunc_exp(raw.squeeze(-1)), "geo_feat": None}
        sigma, feat, _ = self.common_forward(x)
        return {"sigma": sigma, "geo_feat": feat}

    def _reg_grid(self):
        if self.opt.with_sam:
            return self.s_grid
        if self.opt.with_mask:
            return self.m_grid
        return self.grid

    def apply_total_variation(self, w):
        self._reg_grid().grad_total_variation(w)

    def apply_weight_decay(self, w):
        self._reg_grid().grad_weight_decay(w)

    def get_params(self, lr):
        groups = [self.grid, self.grid_mlp, self.view_mlp, self.prop_encoders, self.prop_mlp]
        if self.opt.with_sam:
            groups += [self.s_grid, self.samvit_mlp]
        if self.opt.with_mask:
            groups += [self.m_grid, self.mask_mlp]
        return [{"params": g.parameters(), "lr": lr} for g in groups]
