"""CPU oracle: torch.nn.Module / autograd wrapper around oracle/grid_ref.py.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Same constructor, attributes and state_dict
keys as /root/reference/gridencoder/grid.py:96-161 so that the reference's motion networks (and
the build's) can run on the CPU with it injected as the ``gridencoder`` module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import grid_ref


class _GridEncodeCPU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, S, H, calc_grad_inputs, gridtype, align_corners, interp):
        x = inputs.detach().contiguous().float().numpy()
        emb = embeddings.detach().contiguous().float().numpy()
        offs = offsets.numpy()
        out, dy_dx = grid_ref.grid_encode_forward(x, emb, offs, S, H, calc_grad_inputs, gridtype, align_corners, interp)
        ctx.meta = (x, emb, offs, S, H, dy_dx, gridtype, align_corners, interp)
        L, B, C = out.shape
        return torch.from_numpy(out).permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        x, emb, offs, S, H, dy_dx, gridtype, align_corners, interp = ctx.meta
        B = x.shape[0]
        L, C = len(offs) - 1, emb.shape[1]
        g = grad.detach().contiguous().view(B, L, C).permute(1, 0, 2).contiguous().numpy()
        ge, gi = grid_ref.grid_encode_backward(g, x, emb, offs, S, H, dy_dx, gridtype, align_corners, interp)
        return (None if gi is None else torch.from_numpy(gi)), torch.from_numpy(ge), None, None, None, None, None, None, None


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = grid_ref.per_level_scale(base_resolution, desired_resolution, num_levels)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype_id = {'hash': 0, 'tiled': 1}[gridtype]
        self.interp_id = {'linear': 0, 'smoothstep': 1}[interpolation]
        self.align_corners = align_corners
        offs = grid_ref.make_offsets(input_dim, num_levels, base_resolution, log2_hashmap_size, per_level_scale,
                                     align_corners)
        self.register_buffer('offsets', torch.from_numpy(offs))
        self.embeddings = nn.Parameter(torch.empty(int(offs[-1]), level_dim).uniform_(-1e-4, 1e-4))

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = _GridEncodeCPU.apply(inputs, self.embeddings, self.offsets, float(np.log2(self.per_level_scale)),
                                   self.base_resolution, inputs.requires_grad, self.gridtype_id, self.align_corners,
                                   self.interp_id)
        return out.view(prefix + [self.output_dim])
