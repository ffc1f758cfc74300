"""Drop-in for the reference's ``diff_gauss`` package (GaussianRasterizationSettings,
GaussianRasterizer) on MI355X, backed by libinstag_hip.so through its C ABI.

Reference call sites this mirrors (the package itself is an absent third-party submodule):
  /root/reference/gaussian_renderer/__init__.py:15      from diff_gauss import ...
  /root/reference/gaussian_renderer/__init__.py:58-73   settings (12 keyword fields) + rasterizer ctor
  /root/reference/gaussian_renderer/__init__.py:111-121 call with 9 keyword tensors -> 6-tuple
      (image[3,H,W], depth[1,H,W], normal[3,H,W], alpha[1,H,W], radii[N] int32, extra[E,H,W])
  /root/reference/scene/gaussian_model.py:684           means2D.grad[:, :2] consumer

No CPU path: tensors must live on the GPU and the HIP library must load.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import RasterArgs, check, ptr


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_cuda(**tensors):
    for name, t in tensors.items():
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA/HIP tensor (the MI355X rasterizer has no CPU path)")


# statistics of the most recent forward (bench.py reads the instance count R for the roofline accounting)
LAST_STATS = {"num_rendered": 0, "num_gaussians": 0}


class CapacityPlan:
    """Sync-free (hipGraph-capturable) mode of the rasterizer.

    The k-th rasterizer call of a step uses the fixed instance capacity ``capacities[k]`` instead of
    reading the instance count back to the host, and reports (needed R, overflow flag) into the
    static device tensor ``status[k]``.  Call ``begin_step()`` before every step (eager or capture)."""

    def __init__(self, capacities, device):
        self.capacities = [int(c) for c in capacities]
        self.status = [torch.zeros(2, dtype=torch.int32, device=device) for _ in self.capacities]
        self.index = 0

    def begin_step(self):
        self.index = 0

    def next_slot(self):
        if self.index >= len(self.capacities):
            raise RuntimeError("CapacityPlan: more rasterizer calls in a step than planned capacities")
        k = self.index
        self.index += 1
        return self.capacities[k], self.status[k]

    def overflowed(self):
        """Host check (synchronises): list of (slot, needed R) whose capacity was exceeded."""
        st = torch.stack(self.status).cpu()
        return [(k, int(st[k, 0])) for k in range(len(self.capacities)) if int(st[k, 1]) != 0]

    def needed(self):
        return [int(v) for v in torch.stack(self.status).cpu()[:, 0]]


_CAPACITY_PLAN = None


def set_capacity_plan(plan):
    """Install (or clear with None) the capacity plan used by every subsequent rasterizer forward."""
    global _CAPACITY_PLAN
    _CAPACITY_PLAN = plan


class _State:
    """Opaque device buffers kept between forward and backward (geom / binning / image)."""
    __slots__ = ("args", "keep", "geom", "binning", "image", "R", "radii", "N", "H", "W", "E", "M")


def _make_args(s: GaussianRasterizationSettings, means3D, shs, colors, opac, scales, rots, cov3D, extra):
    N = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    E = 0 if extra is None else (extra.shape[1] if extra.dim() > 1 else 1)
    bg, view, proj, campos = _f32c(s.bg), _f32c(s.viewmatrix), _f32c(s.projmatrix), _f32c(s.campos)
    _require_cuda(bg=bg, viewmatrix=view, projmatrix=proj, campos=campos)
    a = RasterArgs()
    a.N, a.M, a.sh_degree, a.E = N, M, int(s.sh_degree), E
    a.image_height, a.image_width = int(s.image_height), int(s.image_width)
    a.tanfovx, a.tanfovy, a.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
    a.prefiltered, a.debug = int(bool(s.prefiltered)), int(bool(s.debug))
    a.bg, a.viewmatrix, a.projmatrix, a.campos = ptr(bg), ptr(view), ptr(proj), ptr(campos)
    a.means3D, a.shs, a.colors_precomp, a.opacities = ptr(means3D), ptr(shs), ptr(colors), ptr(opac)
    a.scales, a.rotations, a.cov3Ds_precomp, a.extra_attrs = ptr(scales), ptr(rots), ptr(cov3D), ptr(extra)
    keep = (bg, view, proj, campos, means3D, shs, colors, opac, scales, rots, cov3D, extra)
    return a, keep, N, M, E


def rasterize_forward(settings, means3D, shs, colors, opac, scales, rots, cov3D, extra):
    """Run the forward (two-stage with one host round trip, or sync-free under a CapacityPlan).
    Returns (outputs, state)."""
    L = _lib.lib()
    dev = means3D.device
    a, keep, N, M, E = _make_args(settings, means3D, shs, colors, opac, scales, rots, cov3D, extra)
    H, W = a.image_height, a.image_width
    stream = _lib.current_stream()
    geom = torch.empty(L.instag_raster_geom_bytes(N), dtype=torch.uint8, device=dev)
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    image = torch.empty(L.instag_raster_image_bytes(H, W), dtype=torch.uint8, device=dev)
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    normal = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    extra_img = torch.empty(E, H, W, dtype=torch.float32, device=dev)
    if _CAPACITY_PLAN is not None:
        R, status = _CAPACITY_PLAN.next_slot()
        binning = torch.empty(L.instag_raster_binning_bytes(R), dtype=torch.uint8, device=dev)
        check(L.instag_raster_forward_capacity(C.byref(a), ptr(geom), geom.numel(), ptr(binning), binning.numel(),
                                               ptr(image), image.numel(), R, ptr(radii), ptr(status), ptr(color),
                                               ptr(depth), ptr(normal), ptr(alpha),
                                               ptr(extra_img) if E > 0 else None, stream), "rasterize_gaussians")
    else:
        Rc = C.c_int64(0)
        check(L.instag_raster_forward_stage1(C.byref(a), ptr(geom), geom.numel(), ptr(radii), C.byref(Rc), stream),
              "rasterize_gaussians")
        R = int(Rc.value)
        LAST_STATS["num_rendered"], LAST_STATS["num_gaussians"] = R, N
        binning = torch.empty(L.instag_raster_binning_bytes(R), dtype=torch.uint8, device=dev)
        check(L.instag_raster_forward_stage2(C.byref(a), ptr(geom), geom.numel(), ptr(binning), binning.numel(),
                                             ptr(image), image.numel(), R, ptr(color), ptr(depth), ptr(normal),
                                             ptr(alpha), ptr(extra_img) if E > 0 else None, stream),
              "rasterize_gaussians")
    st = _State()
    st.args, st.keep, st.geom, st.binning, st.image = a, keep, geom, binning, image
    st.R, st.radii, st.N, st.H, st.W, st.E, st.M = R, radii, N, H, W, E, M
    return (color, depth, normal, alpha, radii, extra_img), st


def rasterize_backward(st: _State, g_color, g_depth, g_normal, g_alpha, g_extra, want):
    """want: dict name -> bool for means3D, means2D, shs, colors, opacities, scales, rotations, cov3D, extra."""
    L = _lib.lib()
    dev = st.geom.device
    N, M = st.N, st.M
    stream = _lib.current_stream()

    def buf(flag, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=dev) if flag else None

    use_sh = st.args.shs is not None
    use_cov = st.args.cov3Ds_precomp is not None
    out = dict(
        means3D=buf(want["means3D"], N, 3), means2D=buf(want["means2D"], N, 3),
        shs=buf(want["shs"] and use_sh, N, M, 3), colors=buf(want["colors"] and not use_sh, N, 3),
        opacities=buf(want["opacities"], N, 1), scales=buf(want["scales"] and not use_cov, N, 3),
        rotations=buf(want["rotations"] and not use_cov, N, 4), cov3D=buf(want["cov3D"] and use_cov, N, 6),
        extra=buf(want["extra"] and st.E > 0, N, st.E),
    )
    ws = torch.empty(L.instag_raster_backward_workspace_bytes(N, st.R), dtype=torch.uint8, device=dev)
    gs = [None if g is None else _f32c(g) for g in (g_color, g_depth, g_normal, g_alpha, g_extra)]
    check(L.instag_raster_backward(C.byref(st.args), ptr(st.geom), st.geom.numel(), ptr(st.binning),
                                   st.binning.numel(), ptr(st.image), st.image.numel(), st.R, ptr(st.radii),
                                   ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]),
                                   ptr(gs[4]) if st.E > 0 else None, ptr(ws), ws.numel(),
                                   ptr(out["means3D"]), ptr(out["means2D"]), ptr(out["shs"]), ptr(out["colors"]),
                                   ptr(out["opacities"]), ptr(out["scales"]), ptr(out["rotations"]),
                                   ptr(out["cov3D"]), ptr(out["extra"]), stream),
          "rasterize_gaussians_backward")
    return out


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                extra_attrs, raster_settings):
        _require_cuda(means3D=means3D)
        ctx.set_materialize_grads(False)      # unused outputs (depth / normal / extra) stay NULL in backward
        m3, shs, col = _f32c(means3D), _f32c(sh), _f32c(colors_precomp)
        op, sc, ro = _f32c(opacities), _f32c(scales), _f32c(rotations)
        cov, ex = _f32c(cov3Ds_precomp), _f32c(extra_attrs)
        _require_cuda(shs=shs, colors_precomp=col, opacities=op, scales=sc, rotations=ro, cov3Ds_precomp=cov,
                      extra_attrs=ex)
        outs, st = rasterize_forward(raster_settings, m3, shs, col, op, sc, ro, cov, ex)
        ctx.state = st
        ctx.shapes = (opacities.shape, None if extra_attrs is None else extra_attrs.shape)
        color, depth, normal, alpha, radii, extra = outs
        ctx.mark_non_differentiable(radii)
        return color, depth, normal, alpha, radii, extra

    @staticmethod
    def backward(ctx, g_color, g_depth, g_normal, g_alpha, _g_radii, g_extra):
        st = ctx.state
        need = ctx.needs_input_grad
        want = dict(means3D=need[0], means2D=need[1], shs=need[2], colors=need[3], opacities=need[4],
                    scales=need[5], rotations=need[6], cov3D=need[7], extra=need[8])
        g = rasterize_backward(st, g_color, g_depth, g_normal, g_alpha, g_extra, want)
        op_shape, ex_shape = ctx.shapes
        g_op = None if g["opacities"] is None else g["opacities"].reshape(op_shape)
        g_ex = None if g["extra"] is None else g["extra"].reshape(ex_shape)
        return (g["means3D"], g["means2D"], g["shs"], g["colors"], g_op, g["scales"], g["rotations"],
                g["cov3D"], g_ex, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        extra_attrs, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, extra_attrs, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum test of the published rasterizer: view-space z > 0.2."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix
            z = positions[:, 0] * V[0, 2] + positions[:, 1] * V[1, 2] + positions[:, 2] * V[2, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None, extra_attrs=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or \
           ((scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if extra_attrs is not None and extra_attrs.dim() > 1 and extra_attrs.shape[1] > 1:
            raise RuntimeError("extra_attrs: only [N] / [N,1] is supported by the MI355X rasterizer")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3Ds_precomp, extra_attrs, self.raster_settings)


def debug_export(st: _State):
    """Integer / per-Gaussian state of a forward pass as torch tensors (bit-exact parity tests)."""
    L = _lib.lib()
    dev = st.geom.device
    N, R, H, W = st.N, st.R, st.H, st.W
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    out = dict(
        tiles_touched=torch.zeros(N, dtype=torch.int32, device=dev),
        point_offsets=torch.zeros(N, dtype=torch.int32, device=dev),
        keys=torch.zeros(R, dtype=torch.int64, device=dev),
        point_list=torch.zeros(R, dtype=torch.int32, device=dev),
        ranges=torch.zeros(tiles, 2, dtype=torch.int32, device=dev),
        n_contrib=torch.zeros(H, W, dtype=torch.int32, device=dev),
        final_T=torch.zeros(H, W, dtype=torch.float32, device=dev),
        rec2d=torch.zeros(N, 16, dtype=torch.float32, device=dev),
    )
    check(L.instag_raster_debug_export(ptr(st.geom), st.geom.numel(), ptr(st.binning), st.binning.numel(),
                                       ptr(st.image), st.image.numel(), N, R, H, W,
                                       ptr(out["tiles_touched"]), ptr(out["point_offsets"]), ptr(out["keys"]),
                                       ptr(out["point_list"]), ptr(out["ranges"]), ptr(out["n_contrib"]),
                                       ptr(out["final_T"]), ptr(out["rec2d"]), _lib.current_stream()),
          "debug_export")
    out["R"] = R
    out["radii"] = st.radii
    return out
