"""Real spherical-harmonics encoder (degree <= 8) on MI355X.

Operator surface of the reference's shencoder/sphere_harmonics.py (`sh_encode`,
`SHEncoder.forward(inputs, size=1)`), implemented over libsanerf_hip.so.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


class _sh_encoder(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.float().contiguous()          # forced fp32 (sphere_harmonics.py:16)
        B, input_dim = inputs.shape
        out_dim = degree ** 2
        outputs = torch.empty(B, out_dim, dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * out_dim, dtype=torch.float32, device=inputs.device) if calc_grad_inputs else None
        _lib.check(_lib.lib().sn_sh_encode_forward(_lib.dev(inputs, "inputs"), _lib.dev(outputs, "outputs"), B, input_dim,
                                                   degree, _lib.dev(dy_dx, "dy_dx"), _lib.stream()), "sh_encode_forward")
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (B, input_dim, degree)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, input_dim, degree = ctx.dims
        grad = grad.contiguous().float()
        grad_inputs = torch.zeros_like(inputs)
        _lib.check(_lib.lib().sn_sh_encode_backward(_lib.dev(grad, "grad"), _lib.dev(inputs, "inputs"), B, input_dim, degree,
                                                    _lib.dev(dy_dx, "dy_dx"), _lib.dev(grad_inputs, "grad_inputs"),
                                                    _lib.stream()), "sh_encode_backward")
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        inputs = inputs / torch.norm(inputs, dim=-1, keepdim=True)   # sphere_harmonics.py:82
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        out = sh_encode(flat, self.degree, flat.requires_grad)
        return out.reshape(lead + [self.output_dim])
