"""Shared builders for parity tests: same seeded scene on the CPU oracle and on the HIP path."""
import torch

from instag_amd.scene_synth import activated, synthetic_gaussians, toy_cameras
from oracle.rasterize_ref import RasterSettings


def make_scene(n, size, sh_degree=1, seed=0, cam_index=0, bg=(0.0, 1.0, 0.0), scale_mult=1.0):
    cam = toy_cameras(size)[cam_index]
    a = activated(synthetic_gaussians(n, sh_degree=sh_degree, seed=seed))
    a["scales"] = a["scales"] * scale_mult
    a["extra"] = torch.ones(n, 1)
    settings = dict(image_height=size, image_width=size, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                    bg=torch.tensor(bg, dtype=torch.float32), scale_modifier=1.0,
                    viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                    sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
    return a, settings


def oracle_settings(settings):
    return RasterSettings(**settings)


def hip_settings(settings, device="cuda"):
    from instag_amd.diff_gauss import GaussianRasterizationSettings
    s = dict(settings)
    for k in ("bg", "viewmatrix", "projmatrix", "campos"):
        s[k] = s[k].to(device)
    return GaussianRasterizationSettings(**s)


def leaf(t, device=None):
    t = t.detach().clone()
    if device is not None:
        t = t.to(device)
    return t.requires_grad_(True)
