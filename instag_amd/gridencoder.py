"""Drop-in for the reference's ``gridencoder`` package on MI355X.

Mirrors /root/reference/gridencoder/grid.py: ``_grid_encode`` (:24-89), ``grid_encode`` (:93),
``GridEncoder`` (:96-185) -- same constructor arguments, attributes (``embeddings``, ``offsets``,
``output_dim``, ...), state_dict keys and forward / grad_total_variation semantics, with the
native calls replaced by libinstag_hip.so's C ABI (include/instag_hip.h).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from ._lib import check, ptr

_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


def _check_inputs(**tensors):
    for name, t in tensors.items():
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous tensor")


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                gridtype=0, align_corners=False, interpolation=0):
        # inputs: [B, D] float in [0, 1]; embeddings: [sO, C]; offsets: [L + 1] int32 -> [B, L * C]
        inputs = inputs.contiguous().float()
        embeddings = embeddings.contiguous().float()    # fp32 only (InsTaG never enables autocast, grid.py:43-44)
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        if offsets.dtype != torch.int32:
            raise RuntimeError("offsets must be an int tensor")
        _check_inputs(inputs=inputs, embeddings=embeddings, offsets=offsets)

        outputs = torch.empty(L, B, C, device=inputs.device, dtype=torch.float32)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=torch.float32) if calc_grad_inputs else None
        check(_lib.lib().instag_grid_encode_forward(ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs),
                                                    B, D, C, L, S, H, ptr(dy_dx), gridtype, int(align_corners),
                                                    interpolation, _lib.current_stream()), "grid_encode_forward")
        outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous().float()   # [L, B, C]
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs) if dy_dx is not None else None
        check(_lib.lib().instag_grid_encode_backward(ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets),
                                                     ptr(grad_embeddings), B, D, C, L, S, H, ptr(dy_dx),
                                                     ptr(grad_inputs), gridtype, int(ctx.align_corners),
                                                     interpolation, _lib.current_stream()), "grid_encode_backward")
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners

        offsets, offset = [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> "
                f"{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} "
                f"gridtype={self.gridtype} align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)          # map to [0, 1]
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id)
        return outputs.view(prefix_shape + [self.output_dim])

    @torch.no_grad()
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        D = self.input_dim
        C = self.embeddings.shape[1]
        L = self.offsets.shape[0] - 1
        S = float(np.log2(self.per_level_scale))
        H = self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError('grad is None, should be called after loss.backward() and before optimizer.step()!')
        inputs = inputs.contiguous().float()
        lib = _lib.lib()
        total = int(self.embeddings.shape[0])
        ws = torch.empty(lib.instag_grid_total_variation_workspace_bytes(total, C), dtype=torch.uint8,
                         device=self.embeddings.device)
        check(lib.instag_grid_total_variation(ptr(inputs), ptr(self.embeddings), ptr(self.embeddings.grad),
                                              ptr(self.offsets), float(weight), B, D, C, L, S, H,
                                              self.gridtype_id, int(self.align_corners), total, ptr(ws), ws.numel(),
                                              _lib.current_stream()), "grad_total_variation")


class _tri_plane_encode(Function):
    """cat(enc_xy(xy), enc_yz(yz), enc_xz(xz)) of three identically configured 2-D, C=1 encoders in one kernel
    (csrc/grid.hip: triplane_*).  Equivalent to scene/motion_net.py:244-258 of the reference.  Optional ``shift``
    [N, >=3]: the encoders are evaluated at xyz + shift_scale * shift[:, :3].

    ``passthrough``: also return ``xyz`` (and ``shift``) themselves.  A caller whose position has further consumers
    hands THOSE the returned tensors: their gradients then arrive here and are summed into d/dxyz (d/dshift) by the
    backward kernel, instead of by one autograd add launch per extra consumer behind it (C3: four launches, each with a
    cross-queue wait, on the tail of the step)."""

    @staticmethod
    def forward(ctx, xyz, emb_xy, emb_yz, emb_xz, offsets, S, H, bound, shift, shift_scale, passthrough=False):
        xyz = xyz.contiguous().float()
        tabs = [e.contiguous().float() for e in (emb_xy, emb_yz, emb_xz)]
        _check_inputs(xyz=xyz, offsets=offsets, emb_xy=tabs[0], emb_yz=tabs[1], emb_xz=tabs[2])
        shift = None if shift is None else shift.contiguous().float()
        N = xyz.shape[0]
        L = offsets.shape[0] - 1
        T = tabs[0].shape[0]
        out = torch.empty(N, 3 * L, device=xyz.device, dtype=torch.float32)
        check(_lib.lib().instag_triplane_forward(ptr(xyz), ptr(tabs[0]), ptr(tabs[1]), ptr(tabs[2]), ptr(offsets),
                                                 ptr(out), ptr(shift), 0 if shift is None else shift.shape[1],
                                                 float(shift_scale), N, L, S, H, float(bound), T,
                                                 _lib.current_stream()),
              "triplane_forward")
        ctx.save_for_backward(xyz, tabs[0], tabs[1], tabs[2], offsets, *([shift] if shift is not None else []))
        ctx.meta = (N, L, S, H, float(bound), T, float(shift_scale))
        if passthrough:
            return (out, xyz, shift) if shift is not None else (out, xyz)
        return out

    @staticmethod
    def backward(ctx, grad, g_xyz=None, g_shift=None):
        saved = ctx.saved_tensors
        xyz, t0, t1, t2, offsets = saved[:5]
        shift = saved[5] if len(saved) > 5 else None
        N, L, S, H, bound, T, shift_scale = ctx.meta
        grad = grad.contiguous().float()
        L_ = _lib.lib()
        want_shift = shift is not None and ctx.needs_input_grad[8]
        dxyz = torch.empty_like(xyz) if (ctx.needs_input_grad[0] or want_shift) else None
        dshift = torch.empty_like(shift) if want_shift else None
        g_xyz = g_xyz.contiguous().float() if (g_xyz is not None and dxyz is not None) else None
        g_shift = g_shift.contiguous().float() if (g_shift is not None and dshift is not None) else None
        dt = torch.empty(3, T, 1, device=xyz.device, dtype=torch.float32)
        # A field's encoder backward is the last kernel of that field: the weight gradients its MLPs queued
        # (instag_amd/deferred.py) start on the side stream BESIDE it -- the flush comes first, so that the side stream
        # waits for the MLPs' backward only, not for this kernel.
        from . import deferred
        if FLUSH_BEFORE_ENCODER_BACKWARD or shift is None:
            deferred.flush_async(xyz.device)
        ws_bytes = L_.instag_triplane_backward_workspace_bytes(N, T)
        ws = torch.empty(max(1, ws_bytes // 4), device=xyz.device, dtype=torch.float32)
        check(L_.instag_triplane_backward(ptr(grad), ptr(xyz), ptr(t0), ptr(t1), ptr(t2), ptr(offsets),
                                          ptr(dxyz), ptr(dt[0]), ptr(dt[1]), ptr(dt[2]), ptr(ws), ws_bytes,
                                          ptr(shift), 0 if shift is None else shift.shape[1], shift_scale, ptr(dshift),
                                          N, L, S, H, bound, T, ptr(g_xyz), ptr(g_shift), _lib.current_stream()),
              "triplane_backward")
        return ((dxyz if ctx.needs_input_grad[0] else None), dt[0], dt[1], dt[2], None, None, None, None, dshift, None,
                None)


def tri_plane_supported(enc_xy, enc_yz, enc_xz) -> bool:
    encs = (enc_xy, enc_yz, enc_xz)
    if not all(isinstance(e, GridEncoder) for e in encs):
        return False
    e0 = encs[0]
    same = all(e.input_dim == 2 and e.level_dim == 1 and e.num_levels == e0.num_levels
               and e.base_resolution == e0.base_resolution and e.per_level_scale == e0.per_level_scale
               and e.gridtype_id == 0 and not e.align_corners and e.interp_id == 0
               and e.embeddings.shape == e0.embeddings.shape and e.embeddings.is_cuda for e in encs)
    # (tables of up to 13,312 entries per plane are staged in LDS; larger ones -- the mouth field's 46,600 -- are read
    # in place by the triplane_global_* / triplane_level_* kernels behind the same two entry points)
    if not (same and e0.num_levels <= 16):
        return False
    # the fused kernels index every level densely (x + y*(res+1)): no level may be hashed
    for i in range(e0.num_levels):
        res = int(np.ceil(e0.base_resolution * e0.per_level_scale ** i))
        if (res + 1) ** 2 > e0.max_params:
            return False
    return True


_PASS = None
# Also flush before the encoder backward of a field evaluated at a shifted position (the universal one, whose backward
# another field's chain follows)?  Measured on the C3 step: 1.049 ms with, 1.029 without -- the GEMMs then run beside
# the universal field's encoder backward, which is on the critical path, instead of beside the personalised field's.
FLUSH_BEFORE_ENCODER_BACKWARD = False


class passthrough:
    """``with passthrough(carrier):`` -- a tri-plane encode inside the block also leaves its position (and shift) in
    ``carrier["xyz"]`` (``carrier["shift"]``) as outputs of the encode (see _tri_plane_encode): the caller hands those to
    the position's further consumers.  Values are the inputs'; only the route of their gradients changes."""

    def __init__(self, carrier):
        self.carrier = carrier

    def __enter__(self):
        global _PASS
        self.prev, _PASS = _PASS, self.carrier
        return self.carrier

    def __exit__(self, *exc):
        global _PASS
        _PASS = self.prev
        return False


def tri_plane_encode(xyz, enc_xy, enc_yz, enc_xz, bound, shift=None, shift_scale=1.0):
    """xyz [N,3] (+ shift_scale * shift[:, :3]) -> [N, 3*L]; the three encoders must satisfy tri_plane_supported()."""
    e0 = enc_xy
    args = (xyz, enc_xy.embeddings, enc_yz.embeddings, enc_xz.embeddings, e0.offsets,
            float(np.log2(e0.per_level_scale)), int(e0.base_resolution), bound, shift, shift_scale)
    if _PASS is not None and torch.is_grad_enabled() and xyz.requires_grad \
            and xyz.is_contiguous() and xyz.dtype == torch.float32 \
            and (shift is None or (shift.is_contiguous() and shift.dtype == torch.float32)):
        res = _tri_plane_encode.apply(*args, True)
        _PASS["xyz"] = res[1]
        if shift is not None:
            _PASS["shift"] = res[2]
        return res[0]
    return _tri_plane_encode.apply(*args)
