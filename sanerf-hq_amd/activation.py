"""`trunc_exp`: the density activation of the radiance field.

Contract taken from the reference (activation.py:5-17): the forward pass is a plain fp32 `exp`
(inputs are up-cast), the backward pass multiplies the incoming gradient by `exp` of the input
clamped to [-15, 15] so that a large pre-activation cannot blow the gradient up.
"""
import torch

_GRAD_CLAMP = 15.0


class TruncatedExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre_activation):
        x32 = pre_activation.to(torch.float32)
        ctx.save_for_backward(x32)
        return x32.exp()

    @staticmethod
    def backward(ctx, grad_output):
        (x32,) = ctx.saved_tensors
        return grad_output * torch.clamp(x32, -_GRAD_CLAMP, _GRAD_CLAMP).exp()


def trunc_exp(x):
    return TruncatedExp.apply(x)
