"""Fused per-Gaussian glue operators (HIP, csrc/glue.hip) used by the motion networks, the render
composition and the loss block.  Each replaces a chain of eager elementwise ops of the reference:
  motion_glue      scene/motion_net.py:291-306 / :679-692
  deform_activate  gaussian_renderer/__init__.py:200-235 (render_motion, personalized=False, align=True)
  motion_l1_reg    train_face.py:510-514
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr


def _c(t):
    return t.contiguous().float()


class _MotionGlue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_x, aud, eye_pre, enc_a, enc_e, frame_stream=None):
        L = _lib.lib()
        ctx.frame_stream = frame_stream
        ctx.set_materialize_grads(False)
        enc_x, aud, eye_pre, enc_a, enc_e = _c(enc_x), _c(aud), _c(eye_pre), _c(enc_a), _c(enc_e)
        N, KX = enc_x.shape
        KA, KE = aud.shape[1], eye_pre.shape[1]
        h_in = torch.empty(N, KX + KA + KE, dtype=torch.float32, device=enc_x.device)
        amb = torch.empty(N, 3, dtype=torch.float32, device=enc_x.device)
        check(L.instag_motion_glue_forward(ptr(enc_x), ptr(aud), ptr(eye_pre), ptr(enc_a), ptr(enc_e), ptr(h_in),
                                           ptr(amb), N, KX, KA, KE, _lib.current_stream()), "motion_glue_forward")
        ctx.save_for_backward(aud, eye_pre, enc_a, enc_e, amb)
        ctx.dims = (N, KX, KA, KE)
        ctx.mark_non_differentiable()
        return h_in, amb

    @staticmethod
    def backward(ctx, d_h_in, d_amb):
        L = _lib.lib()
        if d_amb is not None:
            # amb doubles as the attention colours of the rasterizer's auxiliary image, whose backward may still be
            # running on its side stream (diff_gauss.DEFER_AUX_JOIN)
            from . import diff_gauss
            diff_gauss.join_pending_aux()
        aud, eye_pre, enc_a, enc_e, amb = ctx.saved_tensors
        N, KX, KA, KE = ctx.dims
        dev = aud.device
        if d_h_in is None:
            d_h_in = torch.zeros(N, KX + KA + KE, dtype=torch.float32, device=dev)
        d_h_in = _c(d_h_in)
        d_amb = None if d_amb is None else _c(d_amb)
        d_enc_x = torch.empty(N, KX, dtype=torch.float32, device=dev)
        d_aud = torch.empty(N, KA, dtype=torch.float32, device=dev)
        d_eye = torch.empty(N, KE, dtype=torch.float32, device=dev)
        parts = torch.empty(L.instag_motion_glue_backward_num_partials(N, KX, KA, KE), KA + KE,
                            dtype=torch.float32, device=dev)
        check(L.instag_motion_glue_backward(ptr(d_h_in), ptr(d_amb), ptr(aud), ptr(eye_pre), ptr(enc_a), ptr(enc_e),
                                            ptr(amb), ptr(d_enc_x), ptr(d_aud), ptr(d_eye), ptr(parts),
                                            N, KX, KA, KE, _lib.current_stream()),
              "motion_glue_backward")
        side = ctx.frame_stream
        from . import deferred
        if side is not None and deferred.active() and _lib.may_fork(dev):
            # the column sums only feed the per-frame branch, whose backward runs on `side`: summing there keeps the launch
            # (and its cross-queue hand-over) out of the per-Gaussian chain that continues on this stream
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                d_vec = parts.sum(dim=0)
            from . import _keepalive
            _keepalive.cross_stream(parts, side)
            deferred.join_at_exit(side)       # (the per-frame branch may be frozen: then nobody else joins `side`)
        else:
            d_vec = parts.sum(dim=0)      # fixed-order column sums of the per-workgroup partials
        return d_enc_x, d_aud, d_eye, d_vec[:KA], d_vec[KA:], None


def motion_glue(enc_x, aud, eye_pre, enc_a, enc_e, frame_stream=None):
    """-> (h_in [N, KX+KA+KE], amb [N,3] = (||aud||, ||relu(eye_pre)||, 0)); enc_a [KA], enc_e [KE] per-frame vectors."""
    return _MotionGlue.apply(enc_x, aud, eye_pre, enc_a.reshape(-1), enc_e.reshape(-1), frame_stream)


def motion_glue_supported(enc_x, aud, eye_pre) -> bool:
    return (enc_x.is_cuda and aud.shape[1] <= 32 and eye_pre.shape[1] <= 8
            and enc_x.shape[1] + aud.shape[1] + eye_pre.shape[1] <= 256)


class _DeformActivate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, h, p, reg_weight):
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        xyz, scaling, rotation, opacity, h, p = (_c(t) for t in (xyz, scaling, rotation, opacity, h, p))
        N = xyz.shape[0]
        dev = xyz.device
        means3D = torch.empty(N, 3, dtype=torch.float32, device=dev)
        scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        rots = torch.empty(N, 4, dtype=torch.float32, device=dev)
        opac = torch.empty(N, 1, dtype=torch.float32, device=dev)
        reg = None
        if reg_weight is not None:
            reg = torch.empty(L.instag_deform_activate_num_reg_partials(N), dtype=torch.float32, device=dev)
        check(L.instag_deform_activate_forward(ptr(xyz), ptr(scaling), ptr(rotation), ptr(opacity), ptr(h), ptr(p),
                                               ptr(means3D), ptr(scales), ptr(rots), ptr(opac), ptr(reg),
                                               0.0 if reg_weight is None else float(reg_weight), N,
                                               _lib.current_stream()), "deform_activate_forward")
        ctx.save_for_backward(scaling, rotation, opacity, h, p)
        ctx.reg_weight = reg_weight
        if reg is None:
            return means3D, scales, rots, opac
        return means3D, scales, rots, opac, reg

    @staticmethod
    def backward(ctx, g_means, g_scales, g_rots, g_opac, g_reg=None):
        L = _lib.lib()
        scaling, rotation, opacity, h, p = ctx.saved_tensors
        N = scaling.shape[0]
        dev = scaling.device
        gs = [None if g is None else _c(g) for g in (g_means, g_scales, g_rots, g_opac)]
        # every partial sum of the regulariser feeds the same scalar: its upstream gradient is one number
        g_reg1 = None if g_reg is None else g_reg.reshape(-1)[:1].contiguous().float()
        d_xyz = torch.empty(N, 3, dtype=torch.float32, device=dev)
        d_scaling = torch.empty(N, 3, dtype=torch.float32, device=dev)
        d_rot = torch.empty(N, 4, dtype=torch.float32, device=dev)
        d_op = torch.empty(N, 1, dtype=torch.float32, device=dev)
        d_h = torch.empty(N, 11, dtype=torch.float32, device=dev)
        d_p = torch.empty(N, 6, dtype=torch.float32, device=dev)
        check(L.instag_deform_activate_backward(ptr(scaling), ptr(rotation), ptr(opacity), ptr(h), ptr(p), ptr(gs[0]),
                                                ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(d_xyz), ptr(d_scaling),
                                                ptr(d_rot), ptr(d_op), ptr(d_h), ptr(d_p), ptr(g_reg1),
                                                0.0 if ctx.reg_weight is None else float(ctx.reg_weight), N,
                                                _lib.current_stream()),
              "deform_activate_backward")
        return d_xyz, d_scaling, d_rot, d_op, d_h, d_p, None


def deform_activate(xyz, scaling, rotation, opacity, h, p, reg_weight=None):
    """means3D, scales, rotations, opacity for render_motion(personalized=False, align=True).  With ``reg_weight`` a
    fifth output holds per-workgroup partial sums of reg_weight * motion_l1_reg(h, p) (train_face.py:510-514): feed it
    to ``losses.face_loss(extra=..., w_extra=1.0)``; its gradient comes back through this operator's backward."""
    return _DeformActivate.apply(xyz, scaling, rotation, opacity, h, p, reg_weight)


class _MotionL1Reg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, p):
        L = _lib.lib()
        h, p = _c(h), _c(p)
        N = h.shape[0]
        parts = torch.empty(L.instag_motion_l1_reg_num_partials(N), dtype=torch.float32, device=h.device)
        check(L.instag_motion_l1_reg_forward(ptr(h), ptr(p), ptr(parts), N, _lib.current_stream()),
              "motion_l1_reg_forward")
        ctx.save_for_backward(h, p)
        return parts.sum()

    @staticmethod
    def backward(ctx, g):
        h, p = ctx.saved_tensors
        N = h.shape[0]
        g = _c(g)
        d_h = torch.empty_like(h)
        d_p = torch.empty_like(p)
        check(_lib.lib().instag_motion_l1_reg_backward(ptr(h), ptr(p), ptr(g), ptr(d_h), ptr(d_p), N,
                                                       _lib.current_stream()), "motion_l1_reg_backward")
        return d_h, d_p


def motion_l1_reg(h, p):
    """mean|d_xyz| + mean|d_rot| + mean|d_opa| + mean|d_scale| + mean|p_xyz| from the raw head outputs."""
    return _MotionL1Reg.apply(h, p)


@torch.no_grad()
def densify_stats(viewspace_grad, radii, max_radii2D, xyz_gradient_accum, denom):
    """In-place densification statistics of one step (train_face.py:626-629) as one launch."""
    N = radii.shape[0]
    assert viewspace_grad.is_contiguous() and max_radii2D.is_contiguous() and xyz_gradient_accum.is_contiguous() \
        and denom.is_contiguous() and radii.dtype == torch.int32 and max_radii2D.dtype == torch.float32
    check(_lib.lib().instag_densify_stats(ptr(viewspace_grad), ptr(radii.contiguous()), ptr(max_radii2D),
                                          ptr(xyz_gradient_accum), ptr(denom), N, _lib.current_stream()),
          "densify_stats")
