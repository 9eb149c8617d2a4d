"""trunc_exp with the reference's contract (activation.py:5-17): exp forward in fp32,
backward g * exp(clamp(x, -15, 15))."""
import torch
from torch.autograd import Function


class _trunc_exp(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply
