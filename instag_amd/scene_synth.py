"""Synthetic scenes and cameras for tests and benchmarks.

Cameras restate the reference chain
  scene/dataset_readers.py:206-213  (NeRF c2w -> COLMAP axes -> w2c -> R, T)
  utils/graphics_utils.py:38-96     (getWorld2View2, getProjectionMatrix, focal2fov)
  scene/cameras.py:55-64            (znear .01, zfar 100, transposed matrices, camera centre)
on poses stored in ``instag_amd/data/toy_cameras.json`` (a 16-frame extract of the
reference's camera fixture, produced by tests/golden/make_golden.py).
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(__file__), "data", "toy_cameras.json")


@dataclass
class Camera:
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4] = w2c^T
    full_proj_transform: torch.Tensor    # [4,4] = w2c^T @ P^T
    camera_center: torch.Tensor          # [3]

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)

    def to(self, device):
        return Camera(self.image_height, self.image_width, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device))


def projection_matrix(znear, zfar, fovX, fovY):
    """OpenGL-style perspective with z_sign=+1 (utils/graphics_utils.py:67-90)."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_from_c2w(c2w, focal, width, height, znear=0.01, zfar=100.0) -> Camera:
    c2w = np.array(c2w, dtype=np.float64).copy()
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    # getWorld2View2 with translate=0, scale=1 inverts twice; keep that rounding path
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = w2c[:3, :3]
    Rt[:3, 3] = w2c[:3, 3]
    Rt[3, 3] = 1.0
    Rt = np.float32(np.linalg.inv(np.linalg.inv(Rt)))
    V = torch.tensor(Rt).transpose(0, 1).contiguous()
    fovx = 2 * math.atan(width / (2 * focal))
    fovy = 2 * math.atan(height / (2 * focal))
    Pt = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (V.unsqueeze(0).bmm(Pt.unsqueeze(0))).squeeze(0).contiguous()
    center = V.inverse()[3, :3].contiguous()
    return Camera(height, width, fovx, fovy, V, full, center)


def toy_cameras(size=512, count=None):
    """Cameras of the reference fixture (focal 1400 px at 512^2; tanfov kept when resized)."""
    with open(_DATA) as f:
        d = json.load(f)
    focal = d["focal_len"] * size / 512.0
    frames = d["frames"] if count is None else d["frames"][:count]
    return [camera_from_c2w(fr["transform_matrix"], focal, size, size) for fr in frames]


def synthetic_gaussians(n, sh_degree=1, seed=0, device="cpu"):
    """Seeded random Gaussians with the distributions of SURVEY.md §8d (raw, pre-activation)."""
    g = torch.Generator().manual_seed(seed)
    M = (sh_degree + 1) ** 2
    xyz = (torch.rand(n, 3, generator=g) * 0.2 - 0.1)
    raw_scale = math.log(0.004) + 0.3 * torch.randn(n, 3, generator=g)
    # reference stores the softplus pre-image (scene/gaussian_model.py:43-44)
    scaling = torch.exp(raw_scale)
    raw_scaling = scaling + torch.log(-torch.expm1(-scaling))
    rot = torch.randn(n, 4, generator=g)
    raw_opacity = 1.5 * torch.randn(n, 1, generator=g)
    f_dc = torch.rand(n, 1, 3, generator=g) * 2 - 1
    f_rest = 0.1 * torch.randn(n, M - 1, 3, generator=g)
    out = dict(xyz=xyz, scaling=raw_scaling, rotation=rot, opacity=raw_opacity,
               features_dc=f_dc, features_rest=f_rest)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def activated(params):
    """Apply the reference activations (scene/gaussian_model.py:43-51,168-196)."""
    return dict(
        means3D=params["xyz"],
        scales=torch.nn.functional.softplus(params["scaling"]),
        rotations=torch.nn.functional.normalize(params["rotation"]),
        opacities=torch.sigmoid(params["opacity"]),
        shs=torch.cat([params["features_dc"], params["features_rest"]], dim=1),
    )


def synthetic_frame(size=512, seed=0, device="cpu", priors=False, background=False):
    """Per-frame training inputs of config C3 (audio window, AU vector, GT image, masks).  ``priors`` adds the
    monocular normal [3,H,W] (unit vectors) / depth [H,W] maps of train_face.py:466-504, ``background`` the
    per-camera scene background [3,H,W] of train_fuse_con.py:113."""
    g = torch.Generator().manual_seed(1000 + seed)
    auds = torch.randn(8, 29, 16, generator=g)
    au_exp = torch.rand(6, generator=g)
    gt = torch.rand(3, size, size, generator=g)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    c = size / 2
    r2 = (xx - c) ** 2 + (yy - c) ** 2
    head = r2 < (0.33 * size) ** 2
    mouth = ((xx - c) ** 2 + (yy - 1.25 * c) ** 2) < (0.06 * size) ** 2
    hair = (r2 < (0.36 * size) ** 2) & (yy < 0.35 * size) & ~head
    # lips_rect = (xmin, xmax, ymin, ymax) as the reference stores it: it indexes attn[1, xmin:xmax, ymin:ymax]
    lips_rect = torch.tensor([int(1.25 * c - 0.07 * size), int(1.25 * c + 0.07 * size),
                              int(c - 0.1 * size), int(c + 0.1 * size)], dtype=torch.int32)
    out = dict(auds=auds, au_exp=au_exp, gt_image=gt, face_mask=head, hair_mask=hair, mouth_mask=mouth,
               lips_rect=lips_rect)
    if priors:
        out["normal"] = torch.nn.functional.normalize(torch.randn(3, size, size, generator=g), dim=0)
        out["depth"] = 0.8 + 0.1 * torch.rand(size, size, generator=g)
    if background:
        out["background"] = torch.rand(3, size, size, generator=g)
    return {k: v.to(device) for k, v in out.items()}
