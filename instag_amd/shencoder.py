"""Drop-in for the reference's ``shencoder`` package on MI355X.

Mirrors /root/reference/shencoder/sphere_harmonics.py: ``_sh_encoder`` (:14-54), ``sh_encode`` (:58),
``SHEncoder`` (:61-107), with the native calls replaced by libinstag_hip.so's C ABI.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr


class _sh_encoder(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        # inputs: [B, input_dim] float in [-1, 1] -> [B, degree^2]; always fp32 (custom_fwd cast, :16)
        inputs = inputs.float().contiguous()
        if not inputs.is_cuda:
            raise RuntimeError("inputs must be a CUDA tensor")
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device) \
            if calc_grad_inputs else None
        check(_lib.lib().instag_sh_encode_forward(ptr(inputs), ptr(outputs), B, input_dim, degree, ptr(dy_dx),
                                                  _lib.current_stream()), "sh_encode_forward")
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        grad = grad.contiguous().float()
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        check(_lib.lib().instag_sh_encode_backward(ptr(grad), ptr(inputs), B, input_dim, degree, ptr(dy_dx),
                                                   ptr(grad_inputs), _lib.current_stream()), "sh_encode_backward")
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
