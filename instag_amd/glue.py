"""Fused per-Gaussian glue operators (HIP, csrc/glue.hip) used by the motion networks, the render
composition and the loss block.  Each replaces a chain of eager elementwise ops of the reference:
  motion_glue      scene/motion_net.py:291-306 / :679-692
  deform_activate  gaussian_renderer/__init__.py:200-235 (render_motion, personalized=False, align=True)
  motion_l1_reg    train_face.py:510-514
"""
from __future__ import annotations

import os

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
        d_vec = _column_sums(parts, ctx.frame_stream, dev)
        return d_enc_x, d_aud, d_eye, d_vec[:KA], d_vec[KA:], None


# sigma_net's input rows [N,74] are not written by the forward (30 MB of its 85 MB at 100k Gaussians); =0 stores them
VIRTUAL_INPUT = os.environ.get("INSTAG_GLUE_VIRTUAL_INPUT", "1") == "1"


class _GlueSigma(torch.autograd.Function):
    """motion_glue followed by sigma_net (scene/motion_net.py:291-306) as ONE autograd node whose backward is one
    kernel: sigma_net's backward writes d_enc_x / d_aud / d_eye_pre from its accumulators and keeps the per-frame
    vectors' column sums in registers (csrc/mlp.hip: mlp_backward_kernel<..., GLUE>), instead of storing the [N,74]
    input gradient for motion_glue_backward to read back (35 us + a column-sum launch on the backward's critical path
    at 100k rows).  The forward forms sigma_net's input rows in the registers that feed its first layer (GLUE variant
    of mlp_forward_kernel) instead of in a kernel of its own, and (VIRTUAL_INPUT) does not store them: their only later
    reader is the first layer's weight gradient, which assembles them again from the same three tensors."""

    @staticmethod
    def forward(ctx, enc_x, aud, eye_pre, enc_a, enc_e, w1, w2, w3, frame_stream):
        from . import mlp as _mlp
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        enc_x, aud, eye_pre, enc_a, enc_e = _c(enc_x), _c(aud), _c(eye_pre), _c(enc_a), _c(enc_e)
        w1c, w2c, w3c = _c(w1), _c(w2), _c(w3)
        N, KX = enc_x.shape
        KA, KE = aud.shape[1], eye_pre.shape[1]
        K0, H, O = KX + KA + KE, w1c.shape[0], w3c.shape[0]
        dev = enc_x.device
        stream = _lib.current_stream()
        # (forward only -- no input needs a gradient: inference, the mouth branch's jaw feature -- nothing is kept)
        keep = any(ctx.needs_input_grad)
        h_in = torch.empty(N, K0, dtype=torch.float32, device=dev) if keep and not VIRTUAL_INPUT else None
        amb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        y = torch.empty(N, O, dtype=torch.float32, device=dev)
        a1 = torch.empty(N, H, dtype=torch.float32, device=dev) if keep else None
        a2 = torch.empty(N, H, dtype=torch.float32, device=dev) if keep else None
        check(L.instag_mlp_forward_glue(ptr(enc_x), ptr(aud), ptr(eye_pre), ptr(enc_a), ptr(enc_e), ptr(w1c), ptr(w2c),
                                        ptr(w3c), ptr(y), ptr(a1), ptr(a2), ptr(h_in), ptr(amb), N, H, O, stream),
              "mlp_forward_glue")
        _mlp.STATS["fwd_flops"] += 2 * N * (K0 * H + H * H + H * O)
        ctx.save_for_backward(aud, eye_pre, enc_a, enc_e, amb, h_in if h_in is not None or not keep else enc_x, w1c,
                              w2c, w3c, a1, a2)
        ctx.virtual = keep and h_in is None
        ctx.weights = (w1, w2, w3)
        ctx.dims = (N, KX, KA, KE, H, O)
        ctx.frame_stream = frame_stream
        return y, amb

    @staticmethod
    def backward(ctx, dy, d_amb):
        from . import deferred, mlp as _mlp
        L = _lib.lib()
        if d_amb is not None:
            from . import diff_gauss
            diff_gauss.join_pending_aux()        # (see _MotionGlue.backward)
        aud, eye_pre, enc_a, enc_e, amb, h_in, w1, w2, w3, a1, a2 = ctx.saved_tensors
        N, KX, KA, KE, H, O = ctx.dims
        dev = aud.device
        dy = torch.zeros(N, O, dtype=torch.float32, device=dev) if dy is None else _c(dy)
        d_amb = None if d_amb is None else _c(d_amb)
        dz1 = torch.empty(N, H, dtype=torch.float32, device=dev)
        dz2 = torch.empty(N, H, dtype=torch.float32, device=dev)
        d_enc_x = torch.empty(N, KX, dtype=torch.float32, device=dev)
        d_aud = torch.empty(N, KA, dtype=torch.float32, device=dev)
        d_eye = torch.empty(N, KE, dtype=torch.float32, device=dev)
        parts = torch.empty(L.instag_mlp_backward_glue_num_partials(N), KA + KE, dtype=torch.float32, device=dev)
        check(L.instag_mlp_backward_glue(ptr(dy), ptr(a1), ptr(a2), ptr(w1), ptr(w2), ptr(w3), ptr(dz1), ptr(dz2),
                                         ptr(aud), ptr(eye_pre), ptr(enc_a), ptr(enc_e), ptr(amb), ptr(d_amb),
                                         ptr(d_enc_x), ptr(d_aud), ptr(d_eye), ptr(parts), N, H, O,
                                         _lib.current_stream()), "mlp_backward_glue")
        _mlp.STATS["bwd_flops"] += 2 * N * (H * O + H * H + (KX + KA + KE) * H)
        deferred.milestone("sigma_backward")
        d_vec = _column_sums(parts, ctx.frame_stream, dev)
        virt = (aud, eye_pre, enc_a, enc_e) if ctx.virtual else None
        grads = [None, None, None]
        queue = deferred.active() and all(w.is_leaf for w in ctx.weights)
        if virt is not None and not queue and ctx.needs_input_grad[5]:
            # (h_in holds enc_x here; outside a deferred block the single-job entry point wants the rows in memory)
            h_in = torch.cat([h_in, aud * enc_a, torch.relu(eye_pre) * enc_e], dim=1)
        jobs = [(dz1, h_in, 0), (dz2, a1, 1), (dy, a2, 2)]
        if queue:
            for dz, inp, idx in jobs:
                if ctx.needs_input_grad[5 + idx]:
                    deferred.defer_weight_grad(dz, inp, ctx.weights[idx], virt if idx == 0 else None)
        else:
            for dz, inp, idx in jobs:
                if ctx.needs_input_grad[5 + idx]:
                    dw = torch.empty(dz.shape[1], inp.shape[1], dtype=torch.float32, device=dev)
                    ws = torch.empty(L.instag_linear_weight_grad_workspace_bytes(N, dz.shape[1], inp.shape[1]),
                                     dtype=torch.uint8, device=dev)
                    _mlp._weight_grad(L, dz, inp, dw, ws)
                    grads[idx] = dw
        return d_enc_x, d_aud, d_eye, d_vec[:KA], d_vec[KA:], grads[0], grads[1], grads[2], None


def _column_sums(parts, side, dev):
    """Fixed-order column sums of the per-workgroup partials; on the per-frame branch's stream when there is one (they
    only feed that branch, whose backward runs there: the launch and its cross-queue hand-over stay out of the
    per-Gaussian chain that continues on the current stream)."""
    from . import deferred
    if side is not None and deferred.active() and _lib.may_fork(dev):
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            d_vec = parts.sum(dim=0)
        from . import _keepalive
        _keepalive.cross_stream(parts, side)
        deferred.join_at_exit(side)           # (the per-frame branch may be frozen: then nobody else joins `side`)
        return d_vec
    return parts.sum(dim=0)


def glue_sigma_supported(enc_x, aud, eye_pre, sigma_net) -> bool:
    if not (motion_glue_supported(enc_x, aud, eye_pre) and sigma_net.num_layers == 3):
        return False
    ws = [layer.weight for layer in sigma_net.net]
    if any(layer.bias is not None for layer in sigma_net.net) or not all(w.is_cuda and w.dtype == torch.float32 for w in ws):
        return False
    KX, KA, KE = enc_x.shape[1], aud.shape[1], eye_pre.shape[1]
    return bool(_lib.lib().instag_mlp_backward_glue_supported(KX + KA + KE, ws[0].shape[0], ws[2].shape[0], KX, KA, KE)) \
        and ws[0].shape[1] == KX + KA + KE


def glue_sigma(enc_x, aud, eye_pre, enc_a, enc_e, sigma_net, frame_stream=None):
    """-> (sigma_net(cat(enc_x, enc_a * aud, enc_e * relu(eye_pre))) [N, out], amb [N,3]); see _GlueSigma."""
    w = [layer.weight for layer in sigma_net.net]
    return _GlueSigma.apply(enc_x, aud, eye_pre, enc_a.reshape(-1), enc_e.reshape(-1), w[0], w[1], w[2], frame_stream)


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


class _MouthActivate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, scaling, rotation, opacity, h, hs, xyz_scale):
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        xyz, scaling, rotation, opacity, h, hs = (_c(t) for t in (xyz, scaling, rotation, opacity, h, hs))
        N, dev = xyz.shape[0], xyz.device
        means3D = torch.empty(N, 3, dtype=torch.float32, device=dev)
        scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        rots = torch.empty(N, 4, dtype=torch.float32, device=dev)
        opac = torch.empty(N, 1, dtype=torch.float32, device=dev)
        sx, sy, sz = xyz_scale
        check(L.instag_mouth_activate_forward(ptr(xyz), ptr(scaling), ptr(rotation), ptr(opacity), ptr(h), ptr(hs), sx, sy,
                                              sz, ptr(means3D), ptr(scales), ptr(rots), ptr(opac), N,
                                              _lib.current_stream()), "mouth_activate_forward")
        ctx.save_for_backward(scaling, rotation, opacity, h, hs)
        ctx.xyz_scale = (sx, sy, sz)
        return means3D, scales, rots, opac

    @staticmethod
    def backward(ctx, g_means, g_scales, g_rots, g_opac):
        L = _lib.lib()
        scaling, rotation, opacity, h, hs = ctx.saved_tensors
        N, dev = scaling.shape[0], scaling.device
        gs = [None if g is None else _c(g) for g in (g_means, g_scales, g_rots, g_opac)]
        d_xyz = torch.empty(N, 3, dtype=torch.float32, device=dev)
        d_scaling = torch.empty(N, 3, dtype=torch.float32, device=dev)
        d_rot = torch.empty(N, 4, dtype=torch.float32, device=dev)
        d_op = torch.empty(N, 1, dtype=torch.float32, device=dev)
        d_h = torch.empty(N, 7, dtype=torch.float32, device=dev)
        d_hs = torch.empty(N, 1, dtype=torch.float32, device=dev)
        sx, sy, sz = ctx.xyz_scale
        check(L.instag_mouth_activate_backward(ptr(scaling), ptr(rotation), ptr(opacity), ptr(h), ptr(hs), sx, sy, sz,
                                               ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(d_xyz), ptr(d_scaling),
                                               ptr(d_rot), ptr(d_op), ptr(d_h), ptr(d_hs), N, _lib.current_stream()),
              "mouth_activate_backward")
        return d_xyz, d_scaling, d_rot, d_op, d_h, d_hs, None


def mouth_activate(xyz, scaling, rotation, opacity, h, hs, xyz_scale=(1e-2 / 5, 1e-2, 1e-2 / 5)):
    """means3D, scales, rotations, opacity of the mouth render (gaussian_renderer/__init__.py:404-420): the mouth
    field's gated displacement (scene/motion_net.py:446-452) added to the positions, and the three activations."""
    return _MouthActivate.apply(xyz, scaling, rotation, opacity, h, hs, tuple(float(v) for v in xyz_scale))


class _AbsMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ncols, scale):
        L = _lib.lib()
        x = _c(x)
        N, stride = x.shape
        part = torch.empty(L.instag_abs_mean_num_partials(N), dtype=torch.float32, device=x.device)
        check(L.instag_abs_mean_forward(ptr(x), N, stride, int(ncols), float(scale), ptr(part), _lib.current_stream()),
              "abs_mean_forward")
        ctx.save_for_backward(x)
        ctx.meta = (int(ncols), float(scale))
        return part

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        (x,) = ctx.saved_tensors
        ncols, scale = ctx.meta
        N, stride = x.shape
        # every partial sum feeds the same scalar: its upstream gradient is one number
        g1 = g.reshape(-1)[:1].contiguous().float()
        dx = torch.empty_like(x)
        check(L.instag_abs_mean_backward(ptr(x), ptr(g1), N, stride, ncols, scale, ptr(dx), _lib.current_stream()),
              "abs_mean_backward")
        return dx, None, None


def abs_mean_partials(x, ncols, scale=1.0):
    """Partial sums of mean|x[:, :ncols] * scale| (x [N, C] on the device): sum them, or hand them to
    ``losses.face_loss / mouth_loss_fused`` as the ``extra`` array (train_mouth.py:203 with x = the alignment head's raw
    output and scale = 1e-2)."""
    return _AbsMean.apply(x, ncols, scale)


class _MouthGlue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_x, enc_a, move):
        L = _lib.lib()
        enc_x, enc_a, move = _c(enc_x), _c(enc_a).reshape(-1), _c(move).reshape(-1)
        N, KX = enc_x.shape
        KA, KM = enc_a.numel(), move.numel()
        in_sigma = torch.empty(N, KX + KA + KM, dtype=torch.float32, device=enc_x.device)
        in_scaler = torch.empty(N, KX + KM, dtype=torch.float32, device=enc_x.device)
        check(L.instag_mouth_glue_forward(ptr(enc_x), ptr(enc_a), ptr(move), ptr(in_sigma), ptr(in_scaler), N, KX, KA, KM,
                                          _lib.current_stream()), "mouth_glue_forward")
        ctx.dims = (N, KX, KA, KM)
        return in_sigma, in_scaler

    @staticmethod
    def backward(ctx, d_sigma, d_scaler):
        L = _lib.lib()
        N, KX, KA, KM = ctx.dims
        dev = (d_sigma if d_sigma is not None else d_scaler).device
        d_sigma = None if d_sigma is None else _c(d_sigma)
        d_scaler = None if d_scaler is None else _c(d_scaler)
        d_enc_x = torch.empty(N, KX, dtype=torch.float32, device=dev)
        parts = torch.empty(L.instag_mouth_glue_backward_num_partials(N), KA, dtype=torch.float32, device=dev)
        check(L.instag_mouth_glue_backward(ptr(d_sigma), ptr(d_scaler), ptr(d_enc_x), ptr(parts), N, KX, KA, KM,
                                           _lib.current_stream()), "mouth_glue_backward")
        d_enc_a = parts.sum(0, keepdim=True) if ctx.needs_input_grad[1] else None
        return d_enc_x, d_enc_a, None


def mouth_glue(enc_x, enc_a, move):
    """(in_sigma, in_scaler) = (cat[enc_x, enc_a.repeat, move.repeat], cat[enc_x, move.repeat]) of the mouth field
    (scene/motion_net.py:437-444) in one launch per pass; enc_a [1,KA<=32], move [1,KM] (no gradient)."""
    return _MouthGlue.apply(enc_x, enc_a, move)


def mouth_glue_supported(enc_x, enc_a, move) -> bool:
    return (enc_x.is_cuda and enc_x.dim() == 2 and enc_x.dtype == torch.float32 and enc_a.numel() <= 32
            and enc_a.dtype == torch.float32 and move.dtype == torch.float32 and not move.requires_grad)


class _FuseCompose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, face, a_face, mouth, a_mouth, bg, scene):
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        face, a_face, mouth, a_mouth, bg = (_c(t) for t in (face, a_face, mouth, a_mouth, bg))
        scene = None if scene is None else _c(scene)
        _, H, W = face.shape
        image, mouth_image = torch.empty_like(face), torch.empty_like(face)
        check(L.instag_fuse_compose_forward(ptr(face), ptr(a_face), ptr(mouth), ptr(a_mouth), ptr(bg), ptr(scene),
                                            ptr(image), ptr(mouth_image), H, W, _lib.current_stream()),
              "fuse_compose_forward")
        ctx.save_for_backward(a_face, mouth_image, bg, *([scene] if scene is not None else []))
        ctx.shapes = (tuple(a_face.shape), tuple(a_mouth.shape))
        return image, mouth_image

    @staticmethod
    def backward(ctx, g_image, g_mouth_image):
        L = _lib.lib()
        saved = ctx.saved_tensors
        a_face, mouth_image, bg = saved[:3]
        scene = saved[3] if len(saved) > 3 else None
        _, H, W = mouth_image.shape
        g_image = None if g_image is None else _c(g_image)
        g_mouth_image = None if g_mouth_image is None else _c(g_mouth_image)
        d_face, d_mouth = torch.empty_like(mouth_image), torch.empty_like(mouth_image)
        d_af = torch.empty(ctx.shapes[0], dtype=torch.float32, device=mouth_image.device)
        d_am = torch.empty(ctx.shapes[1], dtype=torch.float32, device=mouth_image.device)
        check(L.instag_fuse_compose_backward(ptr(g_image), ptr(g_mouth_image), ptr(a_face), ptr(mouth_image), ptr(bg),
                                             ptr(scene), ptr(d_face), ptr(d_af), ptr(d_mouth), ptr(d_am), H, W,
                                             _lib.current_stream()), "fuse_compose_backward")
        return d_face, d_af, d_mouth, d_am, None, None


def fuse_compose(face, a_face, mouth, a_mouth, bg, scene=None):
    """(image, mouth_image) of the fuse stage (train_fuse_con.py:102-121) in one launch per pass; face / mouth [3,H,W],
    the alphas [1,H,W], bg [3], scene [3,H,W] or None (black)."""
    return _FuseCompose.apply(face, a_face, mouth, a_mouth, bg, scene)


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
def densify_stats(viewspace_grad, radii, max_radii2D, xyz_gradient_accum, denom, grad_add=None):
    """In-place densification statistics of one step (train_face.py:626-629) as one launch.  ``grad_add`` [N,3]: a
    second producer's share of the screen-space gradient, added into ``viewspace_grad`` by the same launch first."""
    N = radii.shape[0]
    assert viewspace_grad.is_contiguous() and max_radii2D.is_contiguous() and xyz_gradient_accum.is_contiguous() \
        and denom.is_contiguous() and radii.dtype == torch.int32 and max_radii2D.dtype == torch.float32
    if grad_add is not None:
        assert grad_add.is_contiguous() and grad_add.shape == viewspace_grad.shape and grad_add.dtype == torch.float32
        check(_lib.lib().instag_densify_stats_add(ptr(viewspace_grad), ptr(grad_add), ptr(radii.contiguous()),
                                                  ptr(max_radii2D), ptr(xyz_gradient_accum), ptr(denom), N,
                                                  _lib.current_stream()), "densify_stats_add")
        return
    check(_lib.lib().instag_densify_stats(ptr(viewspace_grad), ptr(radii.contiguous()), ptr(max_radii2D),
                                          ptr(xyz_gradient_accum), ptr(denom), N, _lib.current_stream()),
          "densify_stats")
