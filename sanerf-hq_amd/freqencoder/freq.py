"""NeRF positional (frequency) encoding on MI355X.

Operator surface of the reference's freqencoder/freq.py (`freq_encode`, `FreqEncoder`);
output ordering x | sin f0 | cos f0 | sin f1 | ... (freqencoder.cu:30-58).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


class _freq_encoder(Function):
    @staticmethod
    def forward(ctx, inputs, degree, output_dim):
        inputs = inputs.float().contiguous()
        B, input_dim = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=torch.float32, device=inputs.device)
        _lib.check(_lib.lib().sn_freq_encode_forward(_lib.dev(inputs, "inputs"), B, input_dim, degree, output_dim,
                                                     _lib.dev(outputs, "outputs"), _lib.stream()), "freq_encode_forward")
        ctx.save_for_backward(outputs)
        ctx.dims = (B, input_dim, degree, output_dim)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        (outputs,) = ctx.saved_tensors
        B, input_dim, degree, output_dim = ctx.dims
        grad = grad.contiguous().float()
        grad_inputs = torch.zeros(B, input_dim, dtype=torch.float32, device=grad.device)
        _lib.check(_lib.lib().sn_freq_encode_backward(_lib.dev(grad, "grad"), _lib.dev(outputs, "outputs"), B, input_dim,
                                                      degree, output_dim, _lib.dev(grad_inputs, "grad_inputs"),
                                                      _lib.stream()), "freq_encode_backward")
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        out = freq_encode(flat, self.degree, self.output_dim)
        return out.reshape(lead + [self.output_dim])
