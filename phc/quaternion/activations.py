"""Activation lookup shared by the PHC stack (reference:
phc/quaternion/activations.py:13-35,134-147).  Only the module-returning
helper and Swish are needed by phc.hypercomplex; the QTensor split activations
belong to the quaternion stack, which is out of scope for this engine."""
import torch
import torch.nn as nn


class Swish(torch.autograd.Function):
    """x * sigmoid(x) with a hand-written derivative (ref :13-29)."""

    @staticmethod
    def forward(ctx, i):
        ctx.save_for_backward(i)
        return i * torch.sigmoid(i)

    @staticmethod
    def backward(ctx, grad_output):
        (i,) = ctx.saved_tensors
        s = torch.sigmoid(i)
        return grad_output * (s * (1 + i * (1 - s)))


swish = Swish.apply


class Swish_module(nn.Module):
    def forward(self, x):
        return swish(x)


swish_func = Swish_module()

_MODULES = {"relu": nn.ReLU, "lrelu": nn.LeakyReLU, "elu": nn.ELU, "selu": nn.SELU, "identity": nn.Identity}


def get_module_activation(activation: str):
    name = activation.lower()
    if name == "swish":
        return swish_func
    if name in _MODULES:
        return _MODULES[name]()
    return None
