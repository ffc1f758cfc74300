"""CPU oracle: differentiable 3D-Gaussian-Splatting tile rasterizer (pure PyTorch, fp32).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PARITY UNPINNED against
the reference's ``diff_gauss`` (absent from /root/reference, .gitmodules:4-6).

What it restates
----------------
The operator the reference calls at
  /root/reference/gaussian_renderer/__init__.py:58-73   (settings, 12 fields)
  /root/reference/gaussian_renderer/__init__.py:111-121 (9 inputs -> 6 outputs)
with the semantics of the published 3DGS rasterizer (Kerbl et al. 2023):
preprocess (frustum cull z<=0.2, cov3D=R S S^T R^T, EWA cov2D with 1.3*tanfov
clamp and +0.3 low-pass, conic, radius=ceil(3 sqrt(lambda_max)), 16x16 tile
rect, SH->RGB +0.5 clamp>=0) -> per-tile lists (tiles of the rect that the alpha>=1/255
ellipse can reach: exact tile culling, results unchanged) sorted by (tile, depth) with a
stable sort -> front-to-back alpha blending (skip power>0, alpha=min(.99,o*G),
skip alpha<1/255, stop when T(1-alpha)<1e-4).  The fork's extra outputs are
constrained by the reference's call sites:
  depth  = sum_i w_i * z_view_i              (train_face.py:840, utils/normal_utils.py:9-24)
  alpha  = 1 - T_final                       (train_face.py:585)
  normal = sum_i w_i * n_i, n_i = shortest-axis direction of Gaussian i in view
           space, flipped to face the camera (train_face.py:469)  [build's own choice]
  extra  = sum_i w_i * extra_i               (gaussian_renderer/__init__.py:120)
  render = sum_i w_i * c_i + T_final * bg    (train_face.py:585, synthesize_fuse.py:70)
Pinned pieces: SH->RGB against utils/sh_utils.py:57-117 eval_sh, cov3D against
utils/general_utils.py:71-117, camera matrices against utils/graphics_utils.py:38-96
(tests/test_oracle_golden.py).

Backward = torch autograd of this forward, with the two places where the
published backward differs from the true derivative reproduced explicitly:
  * alpha=min(0.99, o*G): gradient passes straight through the clamp;
  * the 1.3*tanfov clamp of t.x/t.z: a clamped t.x is treated as a constant.
``means2D`` is the reference's gradient carrier (gaussian_renderer/__init__.py:47,
scene/gaussian_model.py:684): it is added to the NDC position, so its gradient
is dL/d(pixel) * 0.5*(W,H).

Every fp32 operation of the preprocess is written out in a fixed order (no
matmul, no FMA) so that the HIP kernels, compiled with -ffp-contract=off, give
bit-identical radii / tile rects / depths / sort keys.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

BLOCK_X = 16
BLOCK_Y = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)

F32 = np.float32


class RasterSettings(NamedTuple):
    """Same 12 fields as the reference passes (gaussian_renderer/__init__.py:58-71)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor   # [4,4], row-vector convention: p_view = [p,1] @ viewmatrix
    projmatrix: torch.Tensor   # [4,4], full projection, same convention
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


def _f(x) -> float:
    """Round a python number to fp32 and return it as a python float."""
    return float(F32(x))


def _sqrt(x: torch.Tensor) -> torch.Tensor:
    """Correctly rounded fp32 sqrt (torch's vectorised CPU sqrtf is 1 ulp off for ~0.7% of inputs;
    the HIP sqrtf is correctly rounded).  sqrt in fp64 then one rounding is exact for fp32 inputs."""
    return torch.sqrt(x.double()).to(x.dtype)


def eval_sh_rgb(deg: int, shs: torch.Tensor, means3D: torch.Tensor, campos: torch.Tensor):
    """SH -> RGB in the fixed evaluation order the HIP kernel uses.  shs: [N, M, 3]."""
    dx = means3D[:, 0] - campos[0]
    dy = means3D[:, 1] - campos[1]
    dz = means3D[:, 2] - campos[2]
    ln = _sqrt((dx * dx + dy * dy) + dz * dz)
    x = (dx / ln)[:, None]
    y = (dy / ln)[:, None]
    z = (dz / ln)[:, None]
    res = _f(SH_C0) * shs[:, 0]
    if deg > 0:
        res = res - _f(SH_C1) * y * shs[:, 1] + _f(SH_C1) * z * shs[:, 2] - _f(SH_C1) * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + _f(SH_C2[0]) * xy * shs[:, 4]
                   + _f(SH_C2[1]) * yz * shs[:, 5]
                   + _f(SH_C2[2]) * (2.0 * zz - xx - yy) * shs[:, 6]
                   + _f(SH_C2[3]) * xz * shs[:, 7]
                   + _f(SH_C2[4]) * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + _f(SH_C3[0]) * y * (3.0 * xx - yy) * shs[:, 9]
                       + _f(SH_C3[1]) * xy * z * shs[:, 10]
                       + _f(SH_C3[2]) * y * (4.0 * zz - xx - yy) * shs[:, 11]
                       + _f(SH_C3[3]) * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
                       + _f(SH_C3[4]) * x * (4.0 * zz - xx - yy) * shs[:, 13]
                       + _f(SH_C3[5]) * z * (xx - yy) * shs[:, 14]
                       + _f(SH_C3[6]) * x * (xx - 3.0 * yy) * shs[:, 15])
    res = res + 0.5
    clamped = res < 0
    return torch.clamp_min(res, 0.0), clamped


def build_cov3d(scales, rotations, scale_modifier):
    """cov3D = R S S^T R^T, upper-triangular 6 (xx,xy,xz,yy,yz,zz); also returns R columns.

    Quaternion (r,x,y,z) is used as given (the caller normalises it,
    scene/gaussian_model.py:51), like utils/general_utils.py:92-105 without the
    re-normalisation."""
    sx = scale_modifier * scales[:, 0]
    sy = scale_modifier * scales[:, 1]
    sz = scale_modifier * scales[:, 2]
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R00 = 1.0 - 2.0 * (y * y + z * z)
    R01 = 2.0 * (x * y - r * z)
    R02 = 2.0 * (x * z + r * y)
    R10 = 2.0 * (x * y + r * z)
    R11 = 1.0 - 2.0 * (x * x + z * z)
    R12 = 2.0 * (y * z - r * x)
    R20 = 2.0 * (x * z - r * y)
    R21 = 2.0 * (y * z + r * x)
    R22 = 1.0 - 2.0 * (x * x + y * y)
    M00, M01, M02 = R00 * sx, R01 * sy, R02 * sz
    M10, M11, M12 = R10 * sx, R11 * sy, R12 * sz
    M20, M21, M22 = R20 * sx, R21 * sy, R22 * sz
    S00 = (M00 * M00 + M01 * M01) + M02 * M02
    S01 = (M00 * M10 + M01 * M11) + M02 * M12
    S02 = (M00 * M20 + M01 * M21) + M02 * M22
    S11 = (M10 * M10 + M11 * M11) + M12 * M12
    S12 = (M10 * M20 + M11 * M21) + M12 * M22
    S22 = (M20 * M20 + M21 * M21) + M22 * M22
    cov = torch.stack([S00, S01, S02, S11, S12, S22], dim=1)
    Rm = ((R00, R01, R02), (R10, R11, R12), (R20, R21, R22))
    return cov, Rm, (sx, sy, sz)


def preprocess(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
               cov3Ds_precomp, extra_attrs, s: RasterSettings):
    """Per-Gaussian stage.  Returns a dict; float entries are differentiable."""
    N = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    V, Pm = s.viewmatrix, s.projmatrix
    tanfovx, tanfovy = _f(s.tanfovx), _f(s.tanfovy)
    focal_x = _f(F32(W) / (F32(2.0) * F32(tanfovx)))
    focal_y = _f(F32(H) / (F32(2.0) * F32(tanfovy)))
    grid_x = (W + BLOCK_X - 1) // BLOCK_X
    grid_y = (H + BLOCK_Y - 1) // BLOCK_Y

    px, py, pz = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    # view-space position (transformPoint4x3)
    tx = ((V[0, 0] * px + V[1, 0] * py) + V[2, 0] * pz) + V[3, 0]
    ty = ((V[0, 1] * px + V[1, 1] * py) + V[2, 1] * pz) + V[3, 1]
    tz = ((V[0, 2] * px + V[1, 2] * py) + V[2, 2] * pz) + V[3, 2]
    in_front = tz.detach() > _f(0.2)

    # clip-space position (transformPoint4x4)
    hx = ((Pm[0, 0] * px + Pm[1, 0] * py) + Pm[2, 0] * pz) + Pm[3, 0]
    hy = ((Pm[0, 1] * px + Pm[1, 1] * py) + Pm[2, 1] * pz) + Pm[3, 1]
    hw = ((Pm[0, 3] * px + Pm[1, 3] * py) + Pm[2, 3] * pz) + Pm[3, 3]
    p_w = 1.0 / (hw + _f(1e-7))
    ndc_x = hx * p_w
    ndc_y = hy * p_w
    if means2D is not None:            # gradient carrier, value is zero
        ndc_x = ndc_x + means2D[:, 0]
        ndc_y = ndc_y + means2D[:, 1]
    pix_x = ((ndc_x + 1.0) * float(W) - 1.0) * 0.5
    pix_y = ((ndc_y + 1.0) * float(H) - 1.0) * 0.5

    # 3D covariance
    if cov3Ds_precomp is not None:
        cov3 = cov3Ds_precomp
        normal_v = torch.zeros(N, 3, dtype=means3D.dtype)
    else:
        cov3, Rm, sc = build_cov3d(scales, rotations, _f(s.scale_modifier))
        sx, sy, sz = (c.detach() for c in sc)
        k0 = (sx <= sy) & (sx <= sz)
        k1 = (~k0) & (sy <= sz)
        # shortest axis = column k of R, in world space
        nwx = torch.where(k0, Rm[0][0], torch.where(k1, Rm[0][1], Rm[0][2]))
        nwy = torch.where(k0, Rm[1][0], torch.where(k1, Rm[1][1], Rm[1][2]))
        nwz = torch.where(k0, Rm[2][0], torch.where(k1, Rm[2][1], Rm[2][2]))
        nvx = (nwx * V[0, 0] + nwy * V[1, 0]) + nwz * V[2, 0]
        nvy = (nwx * V[0, 1] + nwy * V[1, 1]) + nwz * V[2, 1]
        nvz = (nwx * V[0, 2] + nwy * V[1, 2]) + nwz * V[2, 2]
        facing_away = (((nvx * tx + nvy * ty) + nvz * tz).detach() > 0)
        sign = torch.where(facing_away, -1.0, 1.0).to(means3D.dtype)
        normal_v = torch.stack([nvx * sign, nvy * sign, nvz * sign], dim=1)
    S00, S01, S02, S11, S12, S22 = (cov3[:, i] for i in range(6))

    # EWA 2D covariance
    limx = _f(F32(1.3) * F32(tanfovx))
    limy = _f(F32(1.3) * F32(tanfovy))
    txtz = tx / tz
    tytz = ty / tz
    okx = (txtz.detach() >= -limx) & (txtz.detach() <= limx)
    oky = (tytz.detach() >= -limy) & (tytz.detach() <= limy)
    txc = torch.where(okx, txtz * tz, (torch.clamp(txtz, -limx, limx) * tz).detach())
    tyc = torch.where(oky, tytz * tz, (torch.clamp(tytz, -limy, limy) * tz).detach())
    # NB: python_scalar / tensor is evaluated by torch as reciprocal * scalar (two roundings)
    J00 = torch.full_like(tz, focal_x) / tz
    J02 = -(focal_x * txc) / (tz * tz)
    J11 = torch.full_like(tz, focal_y) / tz
    J12 = -(focal_y * tyc) / (tz * tz)
    # T = J * W, W = world->view rotation = V[:3,:3]^T
    T00 = J00 * V[0, 0] + J02 * V[0, 2]
    T01 = J00 * V[1, 0] + J02 * V[1, 2]
    T02 = J00 * V[2, 0] + J02 * V[2, 2]
    T10 = J11 * V[0, 1] + J12 * V[0, 2]
    T11 = J11 * V[1, 1] + J12 * V[1, 2]
    T12 = J11 * V[2, 1] + J12 * V[2, 2]
    U00 = (S00 * T00 + S01 * T01) + S02 * T02
    U10 = (S01 * T00 + S11 * T01) + S12 * T02
    U20 = (S02 * T00 + S12 * T01) + S22 * T02
    U01 = (S00 * T10 + S01 * T11) + S02 * T12
    U11 = (S01 * T10 + S11 * T11) + S12 * T12
    U21 = (S02 * T10 + S12 * T11) + S22 * T12
    c00 = ((T00 * U00 + T01 * U10) + T02 * U20) + _f(0.3)
    c01 = (T00 * U01 + T01 * U11) + T02 * U21
    c11 = ((T10 * U01 + T11 * U11) + T12 * U21) + _f(0.3)

    det = c00 * c11 - c01 * c01
    det_ok = det.detach() != 0
    det_inv = 1.0 / det
    con_x = c11 * det_inv
    con_y = -c01 * det_inv
    con_z = c00 * det_inv
    with torch.no_grad():
        mid = 0.5 * (c00 + c11)
        sq = _sqrt(torch.clamp_min(mid * mid - det, _f(0.1)))
        lam = torch.maximum(mid + sq, mid - sq)
        radius_f = torch.ceil(3.0 * _sqrt(lam))
        finite = torch.isfinite(radius_f) & (radius_f > 0) & torch.isfinite(pix_x.detach()) & torch.isfinite(pix_y.detach())
        radius_f = torch.where(finite, radius_f, torch.zeros_like(radius_f))
        radius = radius_f.to(torch.int32)
        rf = radius.to(torch.float32)
        pxd, pyd = pix_x.detach(), pix_y.detach()
        # (int) casts truncate toward zero; clamp first so the cast is defined
        def _trunc(v):
            v = torch.where(torch.isfinite(v), v, torch.zeros_like(v))
            return torch.clamp(v, -1e9, 1e9).to(torch.int32)
        rminx = torch.clamp(_trunc((pxd - rf) / float(BLOCK_X)), 0, grid_x)
        rminy = torch.clamp(_trunc((pyd - rf) / float(BLOCK_Y)), 0, grid_y)
        rmaxx = torch.clamp(_trunc((((pxd + rf) + float(BLOCK_X)) - 1.0) / float(BLOCK_X)), 0, grid_x)
        rmaxy = torch.clamp(_trunc((((pyd + rf) + float(BLOCK_Y)) - 1.0) / float(BLOCK_Y)), 0, grid_y)
        tiles = (rmaxx - rminx) * (rmaxy - rminy)
        visible = in_front & det_ok & finite & (tiles > 0)
        radii = torch.where(visible, radius, torch.zeros_like(radius))
        tiles_touched = torch.where(visible, tiles, torch.zeros_like(tiles))

    if shs is not None:
        rgb, clamped = eval_sh_rgb(int(s.sh_degree), shs, means3D, s.campos)
    else:
        rgb, clamped = colors_precomp, torch.zeros(N, 3, dtype=torch.bool)

    return dict(
        xy=torch.stack([pix_x, pix_y], dim=1), depth=tz,
        conic=torch.stack([con_x, con_y, con_z], dim=1), opacity=opacities.reshape(N),
        rgb=rgb, clamped=clamped, normal=normal_v,
        extra=(extra_attrs if extra_attrs is not None else torch.zeros(N, 0, dtype=means3D.dtype)),
        cov3D=cov3, cov2D=torch.stack([c00, c01, c11], dim=1),
        radii=radii, tiles_touched=tiles_touched, visible=visible,
        rect=torch.stack([rminx, rminy, rmaxx, rmaxy], dim=1), grid=(grid_x, grid_y),
    )


def _det_ln(x: np.ndarray) -> np.ndarray:
    """ln(x) in fp64 from + - * / only, same expression tree as csrc/raster_preprocess.hip::det_ln, so that the
    culling threshold does not depend on any libm (numpy never contracts a*b+c into an FMA)."""
    x = np.asarray(x, dtype=np.float64)
    m, e = np.frexp(x)
    lo = m < 0.7071067811865476
    m = np.where(lo, m * 2.0, m)
    e = np.where(lo, e - 1, e)
    s = (m - 1.0) / (m + 1.0)
    t = s * s
    p = np.full_like(t, 1.0 / 27.0)
    for k in (25.0, 23.0, 21.0, 19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        p = p * t + 1.0 / k
    p = p * t + 1.0
    return e.astype(np.float64) * 0.6931471805599453 + (2.0 * s) * p


def tile_keep_mask(px, py, A, B, C, thr, tx, ty):
    """Exact tile culling (numpy fp32, same operation order as csrc/raster_preprocess.hip::tile_kept).

    A (tile, Gaussian) pair is kept iff the minimum of q(d) = A dx^2 + 2 B dx dy + C dy^2 over the rectangle of
    the tile's pixel centres is <= thr = 2 ln(255 o) * 1.001 + 0.001, i.e. iff some pixel of the tile can reach
    alpha >= 1/255 (conservatively): dropped pairs contribute nothing to any pixel."""
    f = np.float32
    x0 = (tx * BLOCK_X).astype(f)
    y0 = (ty * BLOCK_Y).astype(f)
    dxl, dxr = x0 - px, (x0 + f(BLOCK_X - 1)) - px
    dyl, dyr = y0 - py, (y0 + f(BLOCK_Y - 1)) - py
    inside = (dxl <= 0) & (dxr >= 0) & (dyl <= 0) & (dyr >= 0)
    B2 = f(2.0) * B
    with np.errstate(all="ignore"):
        ya = np.minimum(np.maximum(-(B * dxl) / C, dyl), dyr)
        yb = np.minimum(np.maximum(-(B * dxr) / C, dyl), dyr)
        xa = np.minimum(np.maximum(-(B * dyl) / A, dxl), dxr)
        xb = np.minimum(np.maximum(-(B * dyr) / A, dxl), dxr)
        e1 = ((A * dxl) * dxl + (B2 * dxl) * ya) + (C * ya) * ya
        e2 = ((A * dxr) * dxr + (B2 * dxr) * yb) + (C * yb) * yb
        e3 = ((A * xa) * xa + (B2 * xa) * dyl) + (C * dyl) * dyl
        e4 = ((A * xb) * xb + (B2 * xb) * dyr) + (C * dyr) * dyr
        qmin = np.minimum(np.minimum(e1, e2), np.minimum(e3, e4))
    return (thr >= 0) & (inside | (qmin <= thr))


def bin_and_sort(pre: dict, cull: str = "exact"):
    """Emit one instance per tile of each visible Gaussian's rectangle (row-major inside the rectangle),
    stable-sort by (tile<<32 | depth bits).

    cull="rect"  : the PUBLISHED binning -- every tile of the bounding rectangle gets an instance
                   (duplicateWithKeys of the published rasterizer; the contract the reference's call
                   sites gaussian_renderer/__init__.py:111-121 rely on).
    cull="exact" : only the tiles the alpha >= 1/255 ellipse can reach (tile_keep_mask) -- what the HIP
                   kernels emit.  Dropped instances never pass the blend loop's own alpha test, so
                   images, final_T and gradients are those of cull="rect"
                   (tests/test_oracle_golden.py::test_exact_tile_culling_changes_nothing)."""
    assert cull in ("exact", "rect")
    grid_x, grid_y = pre["grid"]
    depth, rect = pre["depth"], pre["rect"]
    area = pre["tiles_touched"].numpy().astype(np.int64)            # rectangle area of visible Gaussians
    rect_np = rect.numpy()
    n = len(area)
    cand_g = np.repeat(np.arange(n, dtype=np.int64), area)
    starts = np.repeat(np.cumsum(area) - area, area)
    local = np.arange(len(cand_g), dtype=np.int64) - starts
    rw = (rect_np[:, 2] - rect_np[:, 0]).astype(np.int64)[cand_g]
    ty = rect_np[cand_g, 1].astype(np.int64) + local // np.maximum(rw, 1)
    txx = rect_np[cand_g, 0].astype(np.int64) + local % np.maximum(rw, 1)
    xy = pre["xy"].detach().numpy().astype(np.float32)
    con = pre["conic"].detach().numpy().astype(np.float32)
    op64 = pre["opacity"].detach().numpy().astype(np.float64)
    with np.errstate(all="ignore"):
        thr = ((2.0 * _det_ln(255.0 * op64)) * 1.001 + 0.001).astype(np.float32)
    thr = np.where(np.isfinite(thr), thr, np.float32(-1.0))
    if cull == "rect":
        keep = np.ones(len(cand_g), dtype=bool)
    else:
        keep = tile_keep_mask(xy[cand_g, 0], xy[cand_g, 1], con[cand_g, 0], con[cand_g, 1], con[cand_g, 2],
                              thr[cand_g], txx, ty)
    gids = cand_g[keep]
    tt = np.bincount(gids, minlength=n).astype(np.int64)           # instances per Gaussian after culling
    offsets = np.cumsum(tt)                                         # inclusive scan, like the device scan
    R = int(offsets[-1]) if n else 0
    dbits = depth.detach().numpy().astype(np.float32).view(np.uint32).astype(np.uint64)
    tile = (ty[keep] * grid_x + txx[keep]).astype(np.uint64)
    keys = (tile << np.uint64(32)) | dbits[gids]
    order = np.argsort(keys, kind="stable")
    keys_sorted = keys[order]
    point_list = gids[order].astype(np.int32)
    n_tiles = grid_x * grid_y
    tile_sorted = (keys_sorted >> np.uint64(32)).astype(np.int64)
    starts_t = np.searchsorted(tile_sorted, np.arange(n_tiles), side="left")
    ends_t = np.searchsorted(tile_sorted, np.arange(n_tiles), side="right")
    ranges = np.stack([starts_t, ends_t], axis=1).astype(np.int32)
    ranges[starts_t == ends_t] = 0                 # tiles with no instance keep the zero-filled range
    return dict(R=R, offsets=offsets.astype(np.int64), keys_unsorted=keys, keys=keys_sorted,
                point_list=point_list, ranges=ranges, tiles_touched=tt, candidates=int(len(cand_g)),
                cand_gid=cand_g, cand_tile=(ty * grid_x + txx), cand_keep=keep)


ALPHA_MIN = float(F32(1.0) / F32(255.0))
T_MIN = _f(0.0001)


def blend(pre: dict, binning: dict, s: RasterSettings, chunk: int = 512, decide: Optional[dict] = None):
    """Front-to-back compositing, vectorised over the 256 pixels of each tile.

    The tile's depth-sorted list is consumed in chunks; a tile stops as soon as every pixel has
    terminated (T(1-alpha) < 1e-4), exactly like the per-batch early exit of the kernels.  Within and
    across chunks T is the plain sequential product T <- T*(1-alpha) (cumprod seeded with the carry).

    ``decide`` (fp64 mode): the fp32 per-Gaussian record.  The loop's three discrete decisions -- skip power > 0,
    skip alpha < 1/255, stop at T(1-alpha) < 1e-4 -- are then taken from an fp32 shadow of the recurrence (they are part
    of the specification), the values they gate are computed from ``pre`` in its own precision."""
    H, W = int(s.image_height), int(s.image_width)
    grid_x, grid_y = pre["grid"]
    E = pre["extra"].shape[1]
    feat = torch.cat([pre["rgb"], pre["depth"][:, None], pre["normal"], pre["extra"]], dim=1)
    CH = feat.shape[1]
    dtype = feat.dtype
    out = torch.zeros(CH, H, W, dtype=dtype)
    final_T = torch.ones(H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    point_list = torch.from_numpy(binning["point_list"].astype(np.int64))
    ranges = binning["ranges"]
    shadow = decide is not None
    if shadow:
        d_xy, d_con, d_op = decide["xy"].detach(), decide["conic"].detach(), decide["opacity"].detach()
        ddt = d_xy.dtype

    def masks(xy, con, op, pxf, pyf, T_cur, alive):
        """-> power, alpha (straight-through clamp), contrib, keep, T_excl, T_incl for one chunk."""
        dx = xy[:, 0:1] - pxf[None, :]
        dy = xy[:, 1:2] - pyf[None, :]
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        G = torch.exp(power)
        raw = op[:, None] * G
        alpha = raw - (raw - raw.clamp_max(_f(0.99))).detach()      # straight-through min(.99, .)
        contrib = (power.detach() <= 0) & (alpha.detach() >= ALPHA_MIN)
        a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
        T_all = torch.cumprod(torch.cat([T_cur[None, :], 1.0 - a_eff], dim=0), dim=0)
        keep = alive[None, :] & (T_all[1:].detach() >= T_MIN)   # prefix mask per pixel
        return alpha, contrib, keep

    pieces = []
    for t in range(grid_x * grid_y):
        a, b = int(ranges[t, 0]), int(ranges[t, 1])
        if b <= a:
            continue
        tx0, ty0 = (t % grid_x) * BLOCK_X, (t // grid_x) * BLOCK_Y
        w_ = min(BLOCK_X, W - tx0)
        h_ = min(BLOCK_Y, H - ty0)
        ys, xs = torch.meshgrid(torch.arange(ty0, ty0 + h_), torch.arange(tx0, tx0 + w_), indexing="ij")
        pxf = xs.reshape(-1).to(dtype)
        pyf = ys.reshape(-1).to(dtype)
        npix = pxf.numel()
        T_cur = torch.ones(npix, dtype=dtype)
        alive = torch.ones(npix, dtype=torch.bool)
        acc = torch.zeros(npix, CH, dtype=dtype)
        nc = torch.zeros(npix, dtype=torch.int32)
        if shadow:
            T_sh = torch.ones(npix, dtype=ddt)
        for c0 in range(a, b, chunk):
            ids = point_list[c0:min(b, c0 + chunk)]
            alpha, contrib, keep = masks(pre["xy"][ids], pre["conic"][ids], pre["opacity"][ids], pxf, pyf, T_cur, alive)
            if shadow:
                with torch.no_grad():
                    al_s, contrib, keep = masks(d_xy[ids], d_con[ids], d_op[ids], pxf.to(ddt), pyf.to(ddt), T_sh, alive)
                    T_s = torch.cumprod(torch.cat([T_sh[None, :], 1.0 - torch.where(contrib, al_s, torch.zeros_like(al_s))],
                                                  dim=0), dim=0)[1:]
                    nk = keep.sum(0)
                    T_sh = torch.where(nk > 0, T_s.gather(0, (nk - 1).clamp_min(0)[None, :])[0], T_sh)
            a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
            T_all = torch.cumprod(torch.cat([T_cur[None, :], 1.0 - a_eff], dim=0), dim=0)
            T_excl, T_incl = T_all[:-1], T_all[1:]
            wgt = a_eff * T_excl * keep
            acc = acc + wgt.t() @ feat[ids]
            n_keep = keep.sum(0)
            T_cur = torch.where(n_keep > 0, T_incl.gather(0, (n_keep - 1).clamp_min(0)[None, :])[0], T_cur)
            idx1 = torch.arange(c0 - a + 1, c0 - a + len(ids) + 1, dtype=torch.int32)[:, None]
            nc = torch.maximum(nc, (idx1 * (contrib & keep)).max(0).values)
            alive = keep[-1]
            if not bool(alive.any()):
                break
        pieces.append((ty0, tx0, h_, w_, acc, T_cur, nc))
    for (ty0, tx0, h_, w_, acc, Tf, nc) in pieces:
        out[:, ty0:ty0 + h_, tx0:tx0 + w_] = acc.t().reshape(CH, h_, w_)
        final_T[ty0:ty0 + h_, tx0:tx0 + w_] = Tf.reshape(h_, w_)
        n_contrib[ty0:ty0 + h_, tx0:tx0 + w_] = nc.reshape(h_, w_)
    image = out[0:3] + final_T[None] * s.bg.reshape(3, 1, 1)
    depth = out[3:4]
    normal = out[4:7]
    extra = out[7:7 + E]
    alpha_img = (1.0 - final_T)[None]
    return image, depth, normal, alpha_img, extra, final_T, n_contrib


def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
              cov3Ds_precomp, extra_attrs, settings: RasterSettings, return_aux: bool = False,
              cull: str = "exact", precision: str = "fp32"):
    """Oracle counterpart of ``GaussianRasterizer.forward`` -> 6-tuple (+aux).  ``cull``: see bin_and_sort.

    precision="fp64": every DISCRETE decision of the per-Gaussian stage (visibility, radii, tile rectangles, kept
    tiles, depth order) is taken from the fp32 evaluation -- it is part of the specification, bit for bit -- while the
    differentiable arithmetic (projection, conics, colours, blending and, through autograd, the whole backward) runs
    in double precision on the same values.  Gradient parity tests compare the fp32 kernels against this, so that
    the oracle's own fp32 rounding leaves the comparison."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3Ds_precomp is None) or \
       ((scales is not None or rotations is not None) and cov3Ds_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    assert precision in ("fp32", "fp64")
    if precision == "fp32":
        pre = preprocess(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                         cov3Ds_precomp, extra_attrs, settings)
        binning = bin_and_sort(pre, cull)
        radii = pre["radii"]
    else:
        def det(t):
            return None if t is None else t.detach()

        def dbl(t):
            return None if t is None else t.double()
        with torch.no_grad():
            pre32 = preprocess(det(means3D), det(means2D), det(shs), det(colors_precomp), det(opacities), det(scales),
                               det(rotations), det(cov3Ds_precomp), det(extra_attrs), settings)
        binning = bin_and_sort(pre32, cull)
        pre = preprocess(dbl(means3D), dbl(means2D), dbl(shs), dbl(colors_precomp), dbl(opacities), dbl(scales),
                         dbl(rotations), dbl(cov3Ds_precomp), dbl(extra_attrs), settings)
        # branch decisions that may differ between the two evaluations (SH clamp at exactly 0, ...): reported
        binning["fp64_branch_flips"] = int((pre["clamped"] != pre32["clamped"]).sum())
        for k in ("radii", "tiles_touched", "visible", "rect"):
            pre[k] = pre32[k]
        radii = pre32["radii"]
    image, depth, normal, alpha, extra, final_T, n_contrib = blend(pre, binning, settings,
                                                                   decide=pre32 if precision == "fp64" else None)
    outs = (image, depth, normal, alpha, radii, extra)
    if return_aux:
        aux = dict(pre=pre, binning=binning, final_T=final_T, n_contrib=n_contrib)
        return outs, aux
    return outs
