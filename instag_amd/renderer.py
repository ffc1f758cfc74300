"""Render composition on the MI355X operators: ``render`` and ``render_motion``.

Counterpart of /root/reference/gaussian_renderer/__init__.py: render :37-133, render_motion :151-298
(same argument meaning and returned dictionary keys).  ``viewpoint_camera`` needs the attributes the
reference reads: FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
camera_center, and ``talking_dict`` with ``auds`` [8,29,16] and ``au_exp`` [6] for render_motion.
"""
from __future__ import annotations

import math

import torch

from . import _keepalive
from .diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
from .motion_net import MotionNetwork as _MotionNetwork

CONCURRENT_PASSES = True      # fork the attention raster pass(es) onto a second stream
SHARED_ATTN_PASS = True       # attention map as an auxiliary colour set of the main raster pass
# Fuse stage (training): the mouth pass on a second stream beside the face pass (1.45 -> 1.26 ms per captured step in
# round 2).  Round 2 shipped it switched off behind a segmentation fault inside hipGraphLaunch of a LATER, unrelated
# face-step graph that only a soak of the stage tests showed.  Round 3 removed the two ownership defects the bisect
# pointed at: side streams drawn from torch's round-robin pool could be the SAME HIP stream
# under two names once a process had asked for more than 32 (the "fuse" stream is created late, so it was the one that
# could alias an operator's lane or a capture's warm-up stream: _lib.side_stream now guarantees distinct handles), and
# captured steps kept their loss -- hence their autograd graph and every parameter's gradient accumulator -- alive, bound
# to a per-capture stream that later backward passes then hopped to inside an open capture (one capture stream, detached
# outputs).  The GPU suite passes with the switch on; INSTAG_CONCURRENT_FUSE=0 switches it off.
import os as _os
CONCURRENT_FUSE_PASSES = _os.environ.get("INSTAG_CONCURRENT_FUSE", "1") == "1"


def _side_stream(device):
    """Second HIP stream per device: independent work (the attention raster pass) is forked onto it so that it
    overlaps the main pass -- both blend kernels are bound by their longest tile and leave most CUs idle.
    ``device`` may be a (device, tag) pair: a separate stream per purpose (all from _lib.side_stream's registry)."""
    from . import _lib
    tag = "attn"
    if isinstance(device, tuple):
        device, tag = device
    return _lib.side_stream(device, ("renderer", tag))


def _settings(cam, pc, bg_color, scaling_modifier, debug=False):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=debug)


_ZEROS = {}


def prepare_screenspace(pc):
    """Allocate the zeros behind _screenspace_points for this Gaussian count now (a trainer calls it in front of a stream
    capture: inside one, a first use could only allocate from the capture's pool and would be replayed as a fill)."""
    xyz = pc.get_xyz
    key = (xyz.device, xyz.shape[0], xyz.dtype)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) >= 8:
            _ZEROS.clear()               # (Gaussian counts of past density-control events)
        z = _ZEROS[key] = torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device)
    return z


def _screenspace_points(pc):
    # gradient carrier of the screen-space means (gaussian_renderer/__init__.py:47-52 builds it as zeros + 0 with
    # retain_grad(); a leaf receives the same .grad and saves a launch).  Nothing reads or writes its VALUE -- the
    # rasterizer takes it for the gradient's sake only -- so on the GPU every step's leaf is a fresh view of one zeroed
    # buffer per Gaussian count instead of a fill launch at the head of the step (4 us + a launch gap on the chain)
    xyz = pc.get_xyz
    if not xyz.is_cuda:
        return torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True)
    key = (xyz.device, xyz.shape[0], xyz.dtype)
    z = _ZEROS.get(key)
    if z is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True)
        z = prepare_screenspace(pc)
    return z.detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe=None, bg_color=None, scaling_modifier=1.0, override_color=None):
    """Static render (no motion fields)."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, pc, bg_color, scaling_modifier,
                                              getattr(pipe, "debug", False)))
    opacity = pc.get_opacity
    shs, colors = (pc.get_features, None) if override_color is None else (None, override_color)
    image, depth, normal, alpha, radii, extra = rasterizer(
        means3D=pc.get_xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors, opacities=opacity,
        scales=pc.get_scaling, rotations=pc.get_rotation, cov3Ds_precomp=None,
        extra_attrs=torch.ones_like(opacity))
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "depth": depth, "alpha": alpha, "normal": normal, "radii": radii}


class _BackwardCut(torch.autograd.Function):
    """Identity on a set of tensors, as ONE autograd node: the place where FaceTrainer's three-segment data-parallel
    step cuts the backward pass.  ``torch.autograd.grad(loss, cut_outputs)`` stops AT this node -- it runs the loss
    block, the rasterizer and the deform operator and nothing of the motion fields (without the node the capture
    points would sit on the encoders' own nodes, and the engine would have to run every producer of those nodes'
    other inputs -- the whole sigma-net / attention-head backward -- before it could hand the gradients out);
    ``torch.autograd.backward(cut_outputs, grads)`` continues from here."""

    @staticmethod
    def forward(ctx, *tensors):
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        return grads


MARK_BACKWARD_CUT = False     # set by FaceTrainer._forward_backward_cut around its render_motion call


_ONES = {}


_ZEROS = {}


def _zeros_const(like):
    """Read-only zero tensor of ``like``'s shape (the neutral expression the mouth branch feeds the face field), created
    once per (device, shape, dtype) outside any graph capture."""
    key = (like.device, tuple(like.shape), like.dtype)
    t = _ZEROS.get(key)
    if t is None:
        if like.is_cuda and torch.cuda.is_current_stream_capturing():
            return torch.zeros_like(like)
        t = _ZEROS[key] = torch.zeros_like(like)
    return t


def _ones(like):
    """Constant extra_attrs column, created once per (device, N) outside any graph capture."""
    key = (like.device, tuple(like.shape))
    t = _ONES.get(key)
    if t is None:
        if like.is_cuda and torch.cuda.is_current_stream_capturing():
            return torch.ones_like(like)
        t = _ONES[key] = torch.ones_like(like)
    return t


def render_motion(viewpoint_camera, pc, motion_net, pipe=None, bg_color=None, scaling_modifier=1.0, frame_idx=None,
                  return_attn=False, personalized=False, align=False, detach_motion=False, motion_reg_weight=None):
    """Render with the universal (motion_net) and personalised (pc.neural_motion_grid) motion fields.
    ``motion_reg_weight`` (extension): also return ``motion_reg`` = partial sums of weight * the motion regulariser of
    train_face.py:510-514, computed inside the fused deform operator (None when that operator is not used)."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, pc, bg_color, scaling_modifier,
                                              getattr(pipe, "debug", False)))
    dev = pc.get_xyz.device
    audio_feat = viewpoint_camera.talking_dict["auds"].to(dev, non_blocking=True)
    exp_feat = viewpoint_camera.talking_dict["au_exp"].to(dev, non_blocking=True)

    xyz = pc.get_xyz
    p_motion_preds = None
    cut_tensors = amb_cut = None
    if xyz.is_cuda and hasattr(motion_net, "start_audio"):
        # both networks' audio branches depend only on the frame: start them now, each on its own side stream
        motion_net.start_audio(audio_feat, 1, exp_feat)
        if personalized and hasattr(pc.neural_motion_grid, "start_audio"):
            # (with align only, the personalised field's deformation head is never read: motion_net.py)
            pc.neural_motion_grid.start_audio(audio_feat, 2, exp_feat)
    # The Gaussians' positions feed three operators (personalised field, universal field, deformation): on the device
    # each encode hands the position on as one of its outputs, so the three gradients are summed inside the encoders'
    # backward kernels instead of by autograd add launches on the tail of the step (gridencoder.passthrough)
    from . import gridencoder as _ge
    carrier = {}
    xyz_route = pc.get_xyz
    if personalized or align:
        with _ge.passthrough(carrier):
            p_motion_preds = pc.neural_motion_grid(pc.get_xyz, audio_feat, exp_feat)
        xyz_route = carrier.pop("xyz", xyz_route)
        xyz = xyz_route
    x_shift = None
    p_route = None
    if align:
        p_raw = p_motion_preds.get("_p")
        if p_raw is not None and xyz.is_cuda and isinstance(motion_net, _MotionNetwork):
            # xyz + p_xyz (= p[:, :3] * 1e-2) is formed inside the tri-plane kernel
            x_shift = (p_raw, 1e-2)
            if p_raw.requires_grad:
                # backward reaches p when the universal field is done and the personalised field's chain is about to
                # start: the weight gradients queued so far can run beside it (instag_amd/deferred.py)
                from . import deferred
                p_raw.register_hook(lambda g, dev=dev: deferred.flush_async(dev))
        else:
            xyz = xyz + p_motion_preds["p_xyz"]
    with _ge.passthrough(carrier):
        if x_shift is not None:
            motion_preds = motion_net(xyz, audio_feat, exp_feat, x_shift=x_shift)
        else:
            motion_preds = motion_net(xyz, audio_feat, exp_feat)
    if x_shift is not None or not align:
        xyz_route = carrier.pop("xyz", xyz_route)     # (align without the fused shift encodes xyz + p_xyz, not xyz)
    p_route = carrier.pop("shift", None)
    motion_reg = None

    fused = (align and not personalized and not detach_motion and pc.get_xyz.is_cuda
             and motion_preds.get("_h") is not None and dict.get(p_motion_preds, "_p") is not None
             and motion_preds["_h"].shape[-1] == 11)
    # personalized + align (synthesize_fuse.py:55 with --personalized): the two fields' head outputs add
    # before everything the deform operator does ((d_xyz + p.d_xyz) * p_scale, scaling + d_scale + p.d_scale, ...), so the
    # same operator serves with h + h_p -- one add launch instead of ~20 elementwise ones each way
    fused_pers = (align and personalized and not detach_motion and pc.get_xyz.is_cuda and motion_reg_weight is None
                  and motion_preds.get("_h") is not None and dict.get(p_motion_preds, "_p") is not None
                  and dict.get(p_motion_preds, "_h") is not None and motion_preds["_h"].shape[-1] == 11)
    if fused_pers:
        from .glue import deform_activate
        h_sum = motion_preds["_h"] + p_motion_preds["_h"]
        p_ = p_route if p_route is not None else p_motion_preds["_p"]
        means3D, scales, rotations, opacity = deform_activate(xyz_route, pc._scaling, pc._rotation, pc._opacity, h_sum, p_)
        # (the reference's in-place updates of the returned dictionary / motion_net.cache, :207-217, rebuilt on access)
        motion_preds["d_xyz"] = lambda: (h_sum[..., :3] * 1e-2) * (torch.tanh(p_[..., 3:] / 5) * 0.25 + 1)
        motion_preds["d_scale"], motion_preds["d_rot"] = h_sum[..., 8:11], h_sum[..., 3:7]
        if getattr(motion_net, "cache", None) is not None:
            hd, pd = h_sum.detach(), p_.detach()
            motion_net.cache["d_xyz"] = lambda: (hd[..., :3] * 1e-2) * (torch.tanh(pd[..., 3:] / 5) * 0.25 + 1)
            motion_net.cache["d_scale"], motion_net.cache["d_rot"] = hd[..., 8:11], hd[..., 3:7]
    elif fused:
        # deltas + softplus / normalize / sigmoid in one HIP kernel per pass (instag_amd/glue.py)
        from .glue import deform_activate
        p_in = p_route if p_route is not None else p_motion_preds["_p"]
        h_in, amb_cut = motion_preds["_h"], None
        if MARK_BACKWARD_CUT:
            # every tensor through which a gradient crosses from the rasterizer side (deform operator, attention
            # colours) to the motion fields goes through one identity node: the three-segment step's cut
            amb_in = motion_preds["_amb3"] if (return_attn and SHARED_ATTN_PASS) else None
            ins = [t for t in (xyz_route, h_in, p_in, amb_in) if t is not None]
            cut_tensors = list(_BackwardCut.apply(*ins))
            xyz_c, h_in, p_in = cut_tensors[0], cut_tensors[1], cut_tensors[2]
            amb_cut = cut_tensors[3] if amb_in is not None else None
        else:
            xyz_c = xyz_route
        outs_d = deform_activate(xyz_c, pc._scaling, pc._rotation, pc._opacity, h_in, p_in, motion_reg_weight)
        means3D, scales, rotations, opacity = outs_d[:4]
        motion_reg = outs_d[4] if motion_reg_weight is not None else None
        # The reference scales the universal field's displacement IN PLACE (d_xyz *= p_scale, :217): the dictionary it
        # returns under "motion" -- and motion_net.cache, the same object, which the mouth branch reads at inference
        # (:362-363) -- holds the scaled value.  The fused operator never materialises it; the entries are rebuilt on
        # first access (the regulariser inside deform_activate uses the scaled value too).
        h_, p_ = motion_preds["_h"], p_motion_preds["_p"]
        motion_preds["d_xyz"] = lambda: (h_[..., :3] * 1e-2) * (torch.tanh(p_[..., 3:] / 5) * 0.25 + 1)
        if getattr(motion_net, "cache", None) is not None:
            hd, pd = h_.detach(), p_.detach()
            motion_net.cache["d_xyz"] = lambda: (hd[..., :3] * 1e-2) * (torch.tanh(pd[..., 3:] / 5) * 0.25 + 1)
    else:
        d_xyz, d_scale, d_rot = motion_preds["d_xyz"], motion_preds["d_scale"], motion_preds["d_rot"]
        if personalized:
            d_xyz = d_xyz + p_motion_preds["d_xyz"]
            d_scale = d_scale + p_motion_preds["d_scale"]
            d_rot = d_rot + p_motion_preds["d_rot"]
        if align:
            d_xyz = d_xyz * p_motion_preds["p_scale"]
        # in-place semantics of the reference (d_xyz += ..., d_xyz *= p_scale mutate the entries of the returned
        # dictionary and of motion_net.cache, gaussian_renderer/__init__.py:207-217): the regularisers of
        # train_face.py:508-514 and the mouth branch's jaw feature at inference see the combined, scaled values
        if personalized or align:
            motion_preds["d_xyz"] = d_xyz
            if personalized:
                motion_preds["d_scale"], motion_preds["d_rot"] = d_scale, d_rot
            cache = getattr(motion_net, "cache", None)
            if cache is not None:
                cache["d_xyz"] = d_xyz.detach()
                if personalized:
                    cache["d_scale"], cache["d_rot"] = d_scale.detach(), d_rot.detach()
        if detach_motion:
            d_xyz, d_scale, d_rot = d_xyz.detach(), d_scale.detach(), d_rot.detach()
        means3D = xyz_route + d_xyz
        opacity = pc.get_opacity
        scales = pc.scaling_activation(pc._scaling + d_scale)
        rotations = pc.rotation_activation(pc._rotation + d_rot)
    ones = _ones(opacity)

    def attn_colors(preds):
        if preds.get("_amb3") is not None:
            return preds["_amb3"]
        eye = preds["ambient_eye"]
        return torch.cat([preds["ambient_aud"], eye, torch.zeros_like(eye)], dim=-1)

    def attn_pass(preds, carrier=None):
        out = rasterizer(means3D=means3D.detach(), means2D=screenspace_points if carrier is None else carrier, shs=None,
                         colors_precomp=attn_colors(preds), opacities=opacity.detach(), scales=scales.detach(),
                         rotations=rotations.detach(), cov3Ds_precomp=None, extra_attrs=ones)
        return out[0]

    rendered_attn = p_rendered_attn = None
    # The attention map is rendered over the same (detached) geometry as the image: on the device it rides along
    # the main pass as an auxiliary colour set (one preprocess / binning / sort instead of two).
    shared = return_attn and means3D.is_cuda and SHARED_ATTN_PASS
    from . import _lib
    fork = return_attn and means3D.is_cuda and CONCURRENT_PASSES and (personalized or not shared) \
        and _lib.may_fork(dev)
    if fork:
        # remaining attention pass(es) only share inputs with the main pass: run them on a second stream
        main_stream = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        # The screen-space gradient carrier is a leaf that the side-stream pass(es) AND the main pass write a gradient to.
        # A leaf's gradient accumulator runs on ONE stream (the one current at the leaf's first use): fed directly from
        # two streams, one of the producers always mismatches ("AccumulateGrad node's stream does not match").  The side
        # passes therefore get a view made HERE, on the main stream: their gradient reaches the leaf through the view's
        # backward node, which runs where the view was made.
        side_carrier = screenspace_points.view_as(screenspace_points)
        side.wait_stream(main_stream)
        with torch.cuda.stream(side):
            if not shared:
                rendered_attn = attn_pass(motion_preds, side_carrier)
            if personalized:
                p_rendered_attn = attn_pass(p_motion_preds, side_carrier)

    shs = pc.get_features_pair if (means3D.is_cuda and hasattr(pc, "get_features_pair")) else pc.get_features
    aux = attn_colors(motion_preds) if shared else None
    if cut_tensors is not None and shared:
        aux = amb_cut
    outs = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=opacity,
        scales=scales, rotations=rotations, cov3Ds_precomp=None, extra_attrs=ones,
        **({"aux_colors": aux} if shared else {}))
    image, depth, normal, alpha, radii, extra = outs[:6]
    if shared:
        rendered_attn = outs[6]

    if fork:
        main_stream.wait_stream(side)
        for t in (rendered_attn, p_rendered_attn):
            if t is not None:
                _keepalive.cross_stream(t, main_stream)
    elif return_attn:
        if not shared:
            rendered_attn = attn_pass(motion_preds)
        if personalized:
            p_rendered_attn = attn_pass(p_motion_preds)

    from .motion_net import LazyOutputs
    return LazyOutputs({"render": image, "viewspace_points": screenspace_points,
            "visibility_filter": lambda: radii > 0,      # one launch, only when somebody reads it
            "depth": depth, "alpha": alpha, "normal": normal, "radii": radii, "motion": motion_preds,
            "p_motion": p_motion_preds if personalized or align else None, "attn": rendered_attn,
            "p_attn": p_rendered_attn, "motion_reg": motion_reg,
            # (fused shared-pass path only; None otherwise)
            "_cut": cut_tensors if (cut_tensors is not None and shared and not fork) else None})


def _top_values(v, kmax, largest):
    """The kmax largest (smallest) values of a 1-D tensor, sorted, by row-wise top-k over 512-element rows until
    fewer than 10,000 candidates are left.  (torch.topk on a 1-D tensor of >= 10,000 elements takes a full-sort
    path on ROCm, which does not survive stream capture; the row-wise form selects the same values.)"""
    pad = float("-inf") if largest else float("inf")
    v = v.reshape(-1)
    while v.numel() >= 10000:
        rows = (v.numel() + 511) // 512
        v = torch.nn.functional.pad(v, (0, rows * 512 - v.numel()), value=pad).view(rows, 512)
        v = v.topk(min(kmax, 512), 1, largest, False).values.reshape(-1)
    return v.topk(min(kmax, v.numel()), 0, largest, True).values


def _extreme_values(v, k):
    """(the k largest values descending, the k smallest ascending) of a 1-D tensor: one HIP operator on the device
    (csrc/select.hip, k <= 64), torch.topk otherwise."""
    v = v.reshape(-1)
    k = min(int(k), v.numel())
    if v.is_cuda and v.dtype == torch.float32 and 1 <= k <= 64:
        from . import _lib
        L = _lib.lib()
        v = v.contiguous()
        top = torch.empty(k, dtype=torch.float32, device=v.device)
        bottom = torch.empty(k, dtype=torch.float32, device=v.device)
        ws = torch.empty(L.instag_extreme_values_workspace_bytes(v.numel(), k), dtype=torch.uint8, device=v.device)
        _lib.check(L.instag_extreme_values(_lib.ptr(v), v.numel(), k, _lib.ptr(top), _lib.ptr(bottom), _lib.ptr(ws),
                                           ws.numel(), _lib.current_stream()), "extreme_values")
        return top, bottom
    return _top_values(v, k, True), _top_values(v, k, False)


def _jaw_feature(h, column, scale, k):
    """[1,3] = [max, min, max - min] * 1e2 of the k-th largest / smallest value of scale * h[:, column] in one operator
    (csrc/select.hip instag_jaw_feature).  ``k``: int, or int64 tensor on the device (clamped to [1, 50])."""
    from . import _lib
    L = _lib.lib()
    N, stride = h.shape[0], h.shape[1]
    dev_k = torch.is_tensor(k)
    kmax = min(50, N) if dev_k else min(int(k), N)
    out = torch.empty(1, 3, dtype=torch.float32, device=h.device)
    ws = torch.empty(L.instag_jaw_feature_workspace_bytes(N, kmax), dtype=torch.uint8, device=h.device)
    _lib.check(L.instag_jaw_feature(_lib.ptr(h), N, stride, column, float(scale), kmax, _lib.ptr(k) if dev_k else None,
                                    0 if dev_k else kmax, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "jaw_feature")
    return out


def render_motion_mouth_con(viewpoint_camera, pc, motion_net, pc_face, motion_net_face, pipe=None, bg_color=None,
                            scaling_modifier=1.0, frame_idx=None, return_attn=False, personalized=False, align=False,
                            k=10, inference=False):
    """Mouth branch (gaussian_renderer/__init__.py:302-435): the mouth field is conditioned on a 3-element jaw
    movement feature derived (without gradient) from the k-th largest / smallest vertical displacement the FACE
    field predicts for the face Gaussians with a neutral expression."""
    screenspace_points = _screenspace_points(pc)
    rasterizer = GaussianRasterizer(_settings(viewpoint_camera, pc, bg_color, scaling_modifier,
                                              getattr(pipe, "debug", False)))
    dev = pc.get_xyz.device
    audio_feat = viewpoint_camera.talking_dict["auds"].to(dev, non_blocking=True)
    xyz = pc.get_xyz
    exp_feat = None
    if xyz.is_cuda:
        # the per-frame branches of both fields depend on the frame only: announce them now, each on its own side stream
        if hasattr(motion_net, "start_audio"):
            motion_net.start_audio(audio_feat, 2)
        if not inference and hasattr(motion_net_face, "start_audio"):
            exp_feat = _zeros_const(viewpoint_camera.talking_dict["au_exp"].to(dev, non_blocking=True))
            motion_net_face.start_audio(audio_feat, 1, exp_feat)
    p_motion_preds = None
    # (as in render_motion: the positions feed the personalised field, the mouth field and the activation operator; each
    # encode hands them on as an output, so their three gradients are summed inside the encoders' backward kernels
    # instead of by two autograd add launches -- gridencoder.passthrough)
    from . import gridencoder as _ge
    carrier = {}
    xyz_route = pc.get_xyz
    if personalized or align:
        with _ge.passthrough(carrier):
            p_motion_preds = pc.neural_motion_grid(pc.get_xyz, audio_feat)
        xyz_route = carrier.pop("xyz", xyz_route)
        xyz = xyz_route
    x_shift = None
    if align:
        p_raw = dict.get(p_motion_preds, "_p")
        if torch.is_tensor(p_raw) and xyz.is_cuda and getattr(motion_net, "XYZ_SCALE", None) is not None:
            x_shift = (p_raw, 1e-2)           # xyz + p_xyz (= p[:, :3] * 1e-2) is formed inside the tri-plane kernel
        else:
            xyz = xyz + p_motion_preds["p_xyz"]
    if not inference:
        if exp_feat is None:
            exp_feat = _zeros_const(viewpoint_camera.talking_dict["au_exp"].to(dev, non_blocking=True))
        # (gaussian_renderer/__init__.py:361 evaluates the face field with autograd on and then reads it under no_grad
        # only, :365-372: no gradient can reach it.  Evaluated without a graph here: same values, no activations saved,
        # and -- when this pass runs on a stream of its own beside the face pass -- no gradient accumulators of the
        # face field's parameters created on that stream)
        with torch.no_grad():
            motion_preds_face = motion_net_face(pc_face.get_xyz, audio_feat, exp_feat)
    else:
        motion_preds_face = motion_net_face.cache
    h_face = None if inference else dict.get(motion_preds_face, "_h")
    if (torch.is_tensor(h_face) and h_face.is_cuda and h_face.dim() == 2 and h_face.is_contiguous()
            and h_face.dtype == torch.float32 and h_face.shape[0] >= 1 and (torch.is_tensor(k) or 1 <= int(k) <= 64)
            and (not torch.is_tensor(k) or (k.is_cuda and k.dtype == torch.int64))):
        # d_xyz[:, 1] = h[:, 1] * 1e-2 (scene/motion_net.py:330): selection, gather and the three products in one operator
        with torch.no_grad():
            move_feat = _jaw_feature(h_face.detach(), 1, 1e-2, k)
    else:
        with torch.no_grad():
            dy = motion_preds_face["d_xyz"][..., 1]
            if torch.is_tensor(k):
                # k on the device (int64 [1], 1 <= k <= 50): a captured step draws a new k per replay without a host
                # round trip -- the 50 largest / smallest once, the k-th of them by index
                kmax = min(50, dy.shape[0])
                kidx = (k.reshape(1) - 1).clamp(0, kmax - 1)
                top, bottom = _extreme_values(dy, kmax)
                motion_max = top.gather(0, kidx)[0]
                motion_min = bottom.gather(0, kidx)[0]
            else:
                top, bottom = _extreme_values(dy, k)
                motion_max, motion_min = top[-1], bottom[-1]
            move_feat = torch.stack([motion_max, motion_min, motion_max - motion_min]).reshape(1, 3) * 1e2
    with _ge.passthrough(carrier):
        if x_shift is not None:
            motion_preds = motion_net(xyz, audio_feat, move_feat.detach(), x_shift=x_shift)
        else:
            motion_preds = motion_net(xyz, audio_feat, move_feat.detach())
    if x_shift is not None or not align:
        xyz_route = carrier.pop("xyz", xyz_route)     # (align without the fused shift encodes xyz + p_xyz, not xyz)
    carrier.pop("shift", None)
    h_raw, hs_raw = dict.get(motion_preds, "_h"), dict.get(motion_preds, "_hs")
    if (not personalized and torch.is_tensor(h_raw) and torch.is_tensor(hs_raw) and h_raw.is_cuda
            and h_raw.shape[-1] == 7):
        # gated displacement + softplus / normalize / sigmoid in one HIP kernel per pass (instag_amd/glue.py)
        from .glue import mouth_activate
        means3D, scales, rotations, opacity = mouth_activate(xyz_route, pc._scaling, pc._rotation, pc._opacity, h_raw,
                                                             hs_raw, getattr(motion_net, "XYZ_SCALE",
                                                                             (1e-2 / 5, 1e-2, 1e-2 / 5)))
    else:
        d_xyz = motion_preds["d_xyz"]
        if personalized:
            d_xyz = d_xyz + p_motion_preds["d_xyz"]
        means3D = xyz_route + d_xyz
        opacity = pc.get_opacity
        scales, rotations = pc.get_scaling, pc.rotation_activation(pc._rotation)
    # (dc and rest coefficients as the pair the model stores: no concatenation, no slice copies in backward)
    shs = pc.get_features_pair if (means3D.is_cuda and hasattr(pc, "get_features_pair")) else pc.get_features
    image, depth, normal, alpha, radii, extra = rasterizer(
        means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=opacity,
        scales=scales, rotations=rotations, cov3Ds_precomp=None, extra_attrs=_ones(opacity))
    from .motion_net import LazyOutputs
    return LazyOutputs({"render": image, "viewspace_points": screenspace_points,
                        "visibility_filter": lambda: radii > 0,          # built on first access
                        "depth": depth, "alpha": alpha, "radii": radii, "motion": motion_preds,
                        "p_motion": p_motion_preds if personalized or align else None})


def render_fuse(viewpoint_camera, pc, motion_net, pc_mouth, motion_net_mouth, pipe=None, bg_color=None,
                scene_background=None, personalized=False, inference=False, k=10):
    """Face + mouth composition of the fuse stage: train_fuse_con.py:102-121 (training: both passes carry gradients
    and were rendered over ``bg_color``, which is taken out again) / synthesize_fuse.py:46-66 (inference: the mouth
    field reads the face field's cached motion).  ``scene_background`` [3,H,W] in [0,1] is what shows through both.
    -> dict(image, face=<render_motion pkg>, mouth=<render_motion_mouth_con pkg>)."""
    from . import _lib
    dev = pc.get_xyz.device
    if CONCURRENT_FUSE_PASSES and dev.type == "cuda" and not inference and _lib.may_fork(dev):
        # In training the mouth pass re-evaluates the face field with a neutral expression (it does not read the face
        # pass's cache): the two passes share nothing but parameters, so the mouth pass -- forward here, and with it
        # its whole backward -- runs on a second stream beside the face pass.  Each pass alone is a chain of short
        # kernels that leaves most of the chip idle.
        main, side = torch.cuda.current_stream(dev), _side_stream((dev, "fuse"))
        _lib.leaf_stream(side)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            mouth = render_motion_mouth_con(viewpoint_camera, pc_mouth, motion_net_mouth, pc, motion_net, pipe, bg_color,
                                            personalized=personalized, align=True, k=k, inference=False)
        face = render_motion(viewpoint_camera, pc, motion_net, pipe, bg_color, personalized=personalized, align=True)
        main.wait_stream(side)
        for t in (mouth["render"], mouth["alpha"], mouth["depth"], mouth["radii"]):
            _keepalive.cross_stream(t, main)
    else:
        face = render_motion(viewpoint_camera, pc, motion_net, pipe, bg_color, personalized=personalized, align=True)
        mouth = render_motion_mouth_con(viewpoint_camera, pc_mouth, motion_net_mouth, pc, motion_net, pipe, bg_color,
                                        personalized=personalized, align=True, k=k, inference=inference)
    alpha, alpha_mouth = face["alpha"], mouth["alpha"]
    fr, mr = face["render"], mouth["render"]
    if (fr.is_cuda and fr.dim() == 3 and fr.shape[0] == 3 and fr.dtype == torch.float32 and mr.shape == fr.shape
            and alpha.numel() == fr.shape[1] * fr.shape[2] and alpha_mouth.numel() == alpha.numel()
            and (scene_background is None or scene_background.shape == fr.shape)):
        from .glue import fuse_compose          # ~13 broadcast launches forward, ~25 backward otherwise
        image, mouth_image = fuse_compose(fr, alpha, mr, alpha_mouth, bg_color, scene_background)
    else:
        bg3 = bg_color[:, None, None]
        if scene_background is None:
            scene_background = torch.zeros_like(fr)
        mouth_image = mr - bg3 * (1.0 - alpha_mouth) + scene_background * (1.0 - alpha_mouth)
        image = fr - bg3 * (1.0 - alpha) + mouth_image * (1.0 - alpha)
    return {"image": image, "mouth_image": mouth_image, "face": face, "mouth": mouth}

