"""Gaussian scene state for the MI355X train step: parameters, activations, Adam groups,
densification statistics and densify / prune.

Counterpart of /root/reference/scene/gaussian_model.py (only what the hot path touches):
  activations            :43-51   softplus scale, sigmoid opacity, normalised rotation
  getters                :168-196
  create_from_pcd        :206-335 (3-NN scale init through simple_knn._C.distCUDA2; create_random = synthetic cloud)
  training_setup         :349-403 7 per-Gaussian Adam groups (eps 1e-15) + PMF groups
  update_learning_rate   :421-427 with utils/general_utils.py get_expon_lr_func
  prune / cat / densify  :563-681
  add_densification_stats:683-685
All per-Gaussian tensors stay resident on the GPU; nothing here is CPU-fallback code, it is plain
torch bookkeeping that runs on whatever device the parameters live on (so the host logic is
testable on CPU).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

PARAM_NAMES = ("xyz", "f_dc", "f_rest", "identity", "opacity", "scaling", "rotation")


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def inverse_softplus(x):
    return x + torch.log(-torch.expm1(-x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with optional warm-up (utils/general_utils.py:37-69)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


def quat_to_rotmat(q):
    """Normalised quaternion (r,x,y,z) -> [N,3,3] (utils/general_utils.py:87-105)."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real spherical-harmonic basis [N, (deg+1)^2] at unit directions, in the coefficient order and sign
    convention of utils/sh_utils.py:57-117 (degree <= 3); normalisation constants from their closed forms."""
    assert 0 <= deg <= 3
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    pi = math.pi
    cols = [torch.full_like(x, 0.5 / math.sqrt(pi))]
    if deg >= 1:
        k1 = math.sqrt(3.0 / (4.0 * pi))
        cols += [-k1 * y, k1 * z, -k1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        k2 = 0.5 * math.sqrt(15.0 / pi)
        cols += [k2 * x * y, -k2 * y * z, 0.25 * math.sqrt(5.0 / pi) * (2.0 * zz - xx - yy), -k2 * x * z,
                 0.5 * k2 * (xx - yy)]
    if deg >= 3:
        a = 0.25 * math.sqrt(35.0 / (2.0 * pi))
        b = 0.5 * math.sqrt(105.0 / pi)
        c = 0.25 * math.sqrt(21.0 / (2.0 * pi))
        d = 0.25 * math.sqrt(7.0 / pi)
        cols += [-a * y * (3 * xx - yy), b * x * y * z, -c * y * (4 * zz - xx - yy),
                 d * z * (2 * zz - 3 * xx - 3 * yy), -c * x * (4 * zz - xx - yy), 0.5 * b * z * (xx - yy),
                 -a * x * (xx - 3 * yy)]
    return torch.stack(cols, dim=-1)


def sh_to_rgb(deg: int, features: torch.Tensor, xyz: torch.Tensor, camera_center: torch.Tensor) -> torch.Tensor:
    """View-dependent colour clamp_min(eval_sh + 0.5, 0) of features [N, M, 3] seen from camera_center
    (train_face.py:729-741, gaussian_renderer/__init__.py:100-104)."""
    d = xyz - camera_center.reshape(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    m = (deg + 1) ** 2
    # (elementwise, not einsum: the batched-GEMM route would load the BLAS library for an [N, 4] x [N, 4, 3] contraction --
    # ~0.3 s on first use, in the middle of a density-control event)
    rgb = (sh_basis(deg, d)[:, :, None] * features[:, :m, :]).sum(dim=1)
    return torch.clamp_min(rgb + 0.5, 0.0)


class OptimizationParams:
    """Hot-path values of /root/reference/arguments/__init__.py:79-99."""
    iterations = 10000
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 45000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.003
    rotation_lr = 0.001
    percent_dense = 0.005
    lambda_dssim = 0.2
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 9000
    densify_grad_threshold = 0.0005


class GaussianModel:
    def __init__(self, sh_degree: int = 1, neural_motion_grid: Optional[nn.Module] = None):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        self._p: Dict[str, nn.Parameter] = {}
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0.0
        self.spatial_lr_scale = 1.0
        self.neural_motion_grid = neural_motion_grid
        # GridRenderer: built with the point cloud (scene/gaussian_model.py:317), registered in Adam (:394), stored in
        # every checkpoint (:130), never evaluated by any InsTaG script
        self.neural_renderer = None
        self.scaling_activation = torch.nn.functional.softplus
        self.scaling_inverse_activation = inverse_softplus
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ---- parameters ------------------------------------------------------------------------------
    _xyz = property(lambda s: s._p["xyz"])
    _features_dc = property(lambda s: s._p["f_dc"])
    _features_rest = property(lambda s: s._p["f_rest"])
    _identity = property(lambda s: s._p["identity"])
    _opacity = property(lambda s: s._p["opacity"])
    _scaling = property(lambda s: s._p["scaling"])
    _rotation = property(lambda s: s._p["rotation"])

    @property
    def get_xyz(self):
        return self._p["xyz"]

    @property
    def get_scaling(self):
        return self.scaling_activation(self._p["scaling"])

    @property
    def get_rotation(self):
        return self.rotation_activation(self._p["rotation"])

    @property
    def get_opacity(self):
        return self.opacity_activation(self._p["opacity"])

    @property
    def get_features(self):
        return torch.cat((self._p["f_dc"], self._p["f_rest"]), dim=1)

    @property
    def get_features_pair(self):
        """(features_dc [N,1,3], features_rest [N,M-1,3]) for the MI355X rasterizer, which reads the two parameters
        in place (no concatenation forward, no split of the gradient backward)."""
        return self._p["f_dc"], self._p["f_rest"]

    @property
    def get_identity(self):
        return self._identity

    def get_covariance(self, scaling_modifier=1):
        """Upper triangle (xx, xy, xz, yy, yz, zz) of R S S^T R^T per Gaussian (scene/gaussian_model.py:33-41, 196-197;
        utils/general_utils.py:71-117), from the activated scales and the raw (un-normalised) rotation."""
        R = quat_to_rotmat(self._rotation)
        L = R * (scaling_modifier * self.get_scaling).unsqueeze(1)          # R @ diag(s)
        cov = L @ L.transpose(1, 2)
        return torch.stack((cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]), dim=1)

    def oneupSHdegree(self):
        """scene/gaussian_model.py:201-203."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    @property
    def num_points(self):
        return self._p["xyz"].shape[0]

    def load_raw(self, raw: Dict[str, torch.Tensor], device):
        """raw: dict with xyz, scaling, rotation, opacity, features_dc, features_rest (pre-activation)."""
        n = raw["xyz"].shape[0]
        vals = dict(xyz=raw["xyz"], f_dc=raw["features_dc"], f_rest=raw["features_rest"],
                    identity=torch.zeros(n, 1), opacity=raw["opacity"], scaling=raw["scaling"],
                    rotation=raw["rotation"])
        self._p = {k: nn.Parameter(v.detach().clone().float().to(device).contiguous().requires_grad_(True))
                   for k, v in vals.items()}
        self.max_radii2D = torch.zeros(n, device=device)
        self._build_neural_renderer()
        return self

    def _build_neural_renderer(self):
        """scene/gaussian_model.py:317-320: bound = 1.2 x half the largest extent of the cloud, centred on its mean."""
        from .neural_renderer import GridRenderer
        xyz = self._p["xyz"].detach()
        bound = ((xyz.max(0).values - xyz.min(0).values).max()) / 2 * 1.2
        self.neural_renderer = GridRenderer(bound=bound.cpu(), coord_center=xyz.mean(0).cpu()).to(xyz.device)

    def create_from_pcd(self, points, colors, spatial_lr_scale: float, device="cuda"):
        """Initialise from a point cloud like scene/gaussian_model.py:206-335: SH DC = RGB2SH(colour), opacity 0.1,
        identity rotation, raw scale = log(sqrt(mean squared distance to the 3 nearest neighbours)) clamped at 1e-7
        (``simple_knn._C.distCUDA2`` :246-259).  points [N,3], colors [N,3] in [0,1] (tensors or arrays)."""
        from simple_knn._C import distCUDA2
        pts = torch.as_tensor(points, dtype=torch.float32).to(device)
        col = torch.as_tensor(colors, dtype=torch.float32).to(device)
        n = pts.shape[0]
        M = (self.max_sh_degree + 1) ** 2
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        rot = torch.zeros(n, 4, device=device)
        rot[:, 0] = 1
        raw = dict(xyz=pts, scaling=torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3), rotation=rot,
                   opacity=inverse_sigmoid(0.1 * torch.ones(n, 1, device=device)),
                   features_dc=((col - 0.5) / 0.28209479177387814)[:, None, :],          # utils/sh_utils.py RGB2SH
                   features_rest=torch.zeros(n, M - 1, 3, device=device))
        self.spatial_lr_scale = spatial_lr_scale
        return self.load_raw(raw, device)

    def create_random(self, n, device, spatial_lr_scale=1.0, seed=0, init_scale=0.004):
        """Random cloud in [-0.1,0.1]^3 like scene/dataset_readers.py:353 + create_from_pcd :206-335."""
        g = torch.Generator().manual_seed(seed)
        M = (self.max_sh_degree + 1) ** 2
        scales = torch.full((n, 3), init_scale)
        rot = torch.zeros(n, 4)
        rot[:, 0] = 1
        raw = dict(xyz=torch.rand(n, 3, generator=g) * 0.2 - 0.1, scaling=inverse_softplus(scales), rotation=rot,
                   opacity=inverse_sigmoid(0.1 * torch.ones(n, 1)),
                   features_dc=(torch.rand(n, 1, 3, generator=g) - 0.5) / 0.28209479177387814,
                   features_rest=torch.zeros(n, M - 1, 3))
        self.spatial_lr_scale = spatial_lr_scale
        return self.load_raw(raw, device)

    # ---- checkpoints (scene/gaussian_model.py:115-166, 430-527) -------------------------------------------------------
    def construct_list_of_attributes(self):
        M = self._p["f_rest"].shape[1]
        names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
        names += [f"f_rest_{i}" for i in range(3 * M)] + ["opacity"]
        names += [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
        return names

    def _ply_columns(self, xyz, scale, rotation):
        from collections import OrderedDict
        import numpy as np
        xyz = xyz.detach().cpu().numpy()
        # channel-major like the reference: [N,M,3] -> transpose(1,2) -> [N,3,M] -> flatten
        f_dc = self._p["f_dc"].detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        f_rest = self._p["f_rest"].detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, self._p["opacity"].detach().cpu().numpy(),
                               scale.detach().cpu().numpy(), rotation.detach().cpu().numpy()), axis=1)
        return OrderedDict((k, cols[:, i]) for i, k in enumerate(self.construct_list_of_attributes()))

    def save_ply(self, path):
        from .ply_io import write_vertex_ply
        write_vertex_ply(path, self._ply_columns(self._p["xyz"], self._p["scaling"], self._p["rotation"]))

    def save_deformed_ply(self, xyz, scale, rotation, path):
        """scale is the raw (pre-activation) scale of the deformed Gaussians; stored as log(activation(scale))."""
        from .ply_io import write_vertex_ply
        write_vertex_ply(path, self._ply_columns(xyz, torch.log(self.scaling_activation(scale)), rotation))

    def load_ply(self, path, device="cuda"):
        import numpy as np
        from .ply_io import read_vertex_ply
        c = read_vertex_ply(path)
        n = len(c["x"])
        M1 = (self.max_sh_degree + 1) ** 2 - 1
        rest_names = sorted((k for k in c if k.startswith("f_rest_")), key=lambda k: int(k.split("_")[-1]))
        assert len(rest_names) == 3 * M1, "SH degree of the file does not match the model"
        stack = lambda names: np.stack([np.asarray(c[k], dtype=np.float32) for k in names], axis=1)
        f_dc = stack(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(n, 3, 1)
        f_rest = stack(rest_names).reshape(n, 3, M1) if M1 else np.zeros((n, 3, 0), np.float32)
        scale_names = sorted((k for k in c if k.startswith("scale_")), key=lambda k: int(k.split("_")[-1]))
        rot_names = sorted((k for k in c if k.startswith("rot")), key=lambda k: int(k.split("_")[-1]))
        raw = dict(xyz=torch.from_numpy(stack(["x", "y", "z"])), opacity=torch.from_numpy(stack(["opacity"])),
                   features_dc=torch.from_numpy(f_dc).transpose(1, 2).contiguous(),
                   features_rest=torch.from_numpy(f_rest).transpose(1, 2).contiguous(),
                   scaling=torch.from_numpy(stack(scale_names)), rotation=torch.from_numpy(stack(rot_names)))
        self.active_sh_degree = self.max_sh_degree
        return self.load_raw(raw, device)

    def capture(self):
        """Same 15-tuple as the reference's checkpoint (scene/gaussian_model.py:115-131), including the GridRenderer's
        and the personalised field's state_dicts and the optimizer state in torch.optim's format."""
        opt = self.optimizer.state_dict() if hasattr(self.optimizer, "state_dict") else None
        nmg = None if self.neural_motion_grid is None else self.neural_motion_grid.state_dict()
        nr = None if self.neural_renderer is None else self.neural_renderer.state_dict()
        p = self._p
        return (self.active_sh_degree, p["xyz"], p["f_dc"], p["f_rest"], p["identity"], p["scaling"], p["rotation"],
                p["opacity"], self.max_radii2D, self.xyz_gradient_accum, self.denom, opt, self.spatial_lr_scale, nr, nmg)

    def restore(self, model_args, training_args=None):
        (self.active_sh_degree, xyz, f_dc, f_rest, identity, scaling, rotation, opacity, max_radii2D,
         xyz_gradient_accum, denom, opt_dict, self.spatial_lr_scale, neural_renderer_state, nmg_state) = model_args
        dev = xyz.device
        vals = dict(xyz=xyz, f_dc=f_dc, f_rest=f_rest, identity=identity, opacity=opacity, scaling=scaling,
                    rotation=rotation)
        self._p = {k: nn.Parameter(v.detach().clone().float().contiguous().requires_grad_(True)) for k, v in vals.items()}
        self.max_radii2D = max_radii2D.detach().clone()
        if neural_renderer_state is not None:
            from .neural_renderer import GridRenderer
            self.neural_renderer = GridRenderer()                    # :153-156
            self.neural_renderer.recover_from_ckpt(neural_renderer_state)
            self.neural_renderer.to(dev)
        elif self.neural_renderer is None:
            self._build_neural_renderer()
        if nmg_state is not None and self.neural_motion_grid is not None:
            self.neural_motion_grid.load_state_dict(nmg_state)
        if training_args is not None:
            self.training_setup(training_args)
            if opt_dict is not None and hasattr(self.optimizer, "load_state_dict"):
                self.optimizer.load_state_dict(opt_dict)
        self.xyz_gradient_accum = xyz_gradient_accum.detach().clone().to(dev)
        self.denom = denom.detach().clone().to(dev)
        return self

    # ---- optimizer ---------------------------------------------------------------------------------
    def training_setup(self, opt=OptimizationParams, fused: Optional[bool] = None, capturable: bool = False):
        dev = self._p["xyz"].device
        n = self.num_points
        self.percent_dense = opt.percent_dense
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        lrs = dict(xyz=opt.position_lr_init * self.spatial_lr_scale, f_dc=opt.feature_lr,
                   f_rest=opt.feature_lr / 20.0, identity=1e-2, opacity=opt.opacity_lr, scaling=opt.scaling_lr,
                   rotation=opt.rotation_lr)
        groups = [{"params": [self._p[k]], "lr": lrs[k], "name": k} for k in PARAM_NAMES]
        if self.neural_renderer is not None:
            # same groups at the same position as scene/gaussian_model.py:394, so that a reference optimizer
            # state_dict (groups matched by position) loads; the parameters never receive a gradient
            groups += [dict(g, lazy=True) for g in self.neural_renderer.get_params(lr=5e-3, lr_net=5e-4)]
        if self.neural_motion_grid is not None:
            groups += self.neural_motion_grid.get_params(lr=1e-3, lr_net=1e-4)
        if fused is None:
            fused = dev.type == "cuda"
        if fused:
            # GPU: every group in one HIP launch, learning rates / step counter in device memory (instag_amd/optim.py)
            from .optim import MultiTensorAdam
            self.optimizer = MultiTensorAdam(groups, lr=0.0, betas=(0.9, 0.999), eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(
            lr_init=opt.position_lr_init * self.spatial_lr_scale, lr_final=opt.position_lr_final * self.spatial_lr_scale,
            lr_delay_mult=opt.position_lr_delay_mult, max_steps=opt.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group.get("name") == "xyz":
                lr = float(self.xyz_scheduler_args(iteration))
                if isinstance(group["lr"], torch.Tensor):
                    group["lr"].fill_(lr)
                else:
                    group["lr"] = lr
                return lr

    # ---- densification ---------------------------------------------------------------------------------
    @torch.no_grad()
    def add_densification_stats(self, viewspace_grad, update_filter):
        """viewspace_grad: means2D.grad [N,3] (NDC units), update_filter: radii > 0."""
        norm = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True)
        m = update_filter[:, None].to(norm.dtype)
        self.xyz_gradient_accum += norm * m
        self.denom += m

    @torch.no_grad()
    def _rebuild(self, keep: Optional[torch.Tensor], extra: Optional[Dict[str, torch.Tensor]], append_first: bool = False):
        """New parameter set = cat(old[keep], extra) -- or, ``append_first``, cat(old, extra)[keep]; Adam moments follow
        (zeros for new rows).  ``keep``: a boolean mask
        or the row indices to keep (one ``nonzero`` -- one host round trip for the count -- serves all 21 tensors; a
        boolean index per tensor is a round trip each)."""
        if keep is not None and keep.dtype == torch.bool:
            keep = keep.nonzero(as_tuple=False).squeeze(1)
        for group in self.optimizer.param_groups:
            name = group.get("name", "")
            if name not in self._p:
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)

            def remap(t, fill_zero):
                if keep is not None and not append_first:
                    t = t.index_select(0, keep)
                if extra is not None:
                    add = torch.zeros_like(extra[name]) if fill_zero else extra[name]
                    t = torch.cat((t, add), dim=0)
                if keep is not None and append_first:
                    t = t.index_select(0, keep)
                return t
            new = nn.Parameter(remap(old.data, False).contiguous().requires_grad_(True))
            if state is not None and "exp_avg" in state:
                state["exp_avg"] = remap(state["exp_avg"], True).contiguous()
                state["exp_avg_sq"] = remap(state["exp_avg_sq"], True).contiguous()
                self.optimizer.state[new] = state
            group["params"][0] = new
            self._p[name] = new
        if hasattr(self.optimizer, "invalidate"):
            self.optimizer.invalidate()

    @torch.no_grad()
    def prune_points(self, mask):
        keep = (~mask).nonzero(as_tuple=False).squeeze(1)
        self._rebuild(keep, None)
        self.xyz_gradient_accum = self.xyz_gradient_accum.index_select(0, keep)
        self.denom = self.denom.index_select(0, keep)
        self.max_radii2D = self.max_radii2D.index_select(0, keep)

    @torch.no_grad()
    def _append(self, extra):
        self._rebuild(None, extra)
        dev, n = self._p["xyz"].device, self.num_points
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    @torch.no_grad()
    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = (torch.norm(grads, dim=-1) >= grad_threshold) & \
              (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)
        idx = sel.nonzero(as_tuple=False).squeeze(1)
        self._append({k: self._p[k].data.index_select(0, idx) for k in PARAM_NAMES})

    @torch.no_grad()
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        n0 = self.num_points
        padded = torch.zeros(n0, device=grads.device)
        padded[:grads.shape[0]] = grads.squeeze(-1)
        sel = (padded >= grad_threshold) & (self.get_scaling.max(dim=1).values > self.percent_dense * scene_extent)
        idx = sel.nonzero(as_tuple=False).squeeze(1)
        scaling_sel = self.get_scaling.index_select(0, idx)
        stds = scaling_sel.repeat(N, 1)
        samples = torch.randn(stds.shape, device=stds.device, generator=generator) * stds
        rots = quat_to_rotmat(self._p["rotation"].data.index_select(0, idx)).repeat(N, 1, 1)
        extra = {k: self._p[k].data.index_select(0, idx).repeat(N, *([1] * (self._p[k].dim() - 1))) for k in PARAM_NAMES}
        # (R @ sample written elementwise: torch.bmm would load the BLAS library -- 0.9 s on first use, inside the event)
        extra["xyz"] = (rots * samples.unsqueeze(1)).sum(dim=-1) + extra["xyz"]
        extra["scaling"] = self.scaling_inverse_activation(scaling_sel.repeat(N, 1) / (0.8 * N))
        self._append(extra)
        prune = torch.cat((sel, torch.zeros(N * idx.numel(), device=sel.device, dtype=torch.bool)))
        self.prune_points(prune)

    @torch.no_grad()
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """scene/gaussian_model.py:663-681 -- clone, split, prune -- as ONE rebuild of the parameter set: the reference's
        sequence rebuilds every parameter and both of its Adam moments four times (append clones, append children, drop
        the split parents, prune), each a gather + concatenation per tensor and a host round trip for the new count.  The
        final set is [old rows that are neither split nor pruned, clones, children] in that order either way; what each
        prune test would see is known before anything is rebuilt (a clone carries its parent's opacity and scale, a
        child its parent's opacity and 1 / 1.6 of its scale; the screen radii are zero after an append).  Same rows, same
        values, same random draws as ``densify_and_prune_stepwise`` (tests/test_host_logic.py)."""
        if max_grad <= 0:
            return self.densify_and_prune_stepwise(max_grad, min_opacity, extent, max_screen_size, generator)
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        N = 2
        scal = self.get_scaling
        big = scal.max(dim=1).values > self.percent_dense * extent
        hot = torch.norm(grads, dim=-1) >= max_grad
        sel_split = (grads.squeeze(-1) >= max_grad) & big
        idx_c = (hot & ~big).nonzero(as_tuple=False).squeeze(1)
        idx_s = sel_split.nonzero(as_tuple=False).squeeze(1)
        scaling_sel = scal.index_select(0, idx_s)
        stds = scaling_sel.repeat(N, 1)
        samples = torch.randn(stds.shape, device=stds.device, generator=generator) * stds
        rots = quat_to_rotmat(self._p["rotation"].data.index_select(0, idx_s)).repeat(N, 1, 1)
        child = {k: self._p[k].data.index_select(0, idx_s).repeat(N, *([1] * (self._p[k].dim() - 1))) for k in PARAM_NAMES}
        child["xyz"] = (rots * samples.unsqueeze(1)).sum(dim=-1) + child["xyz"]
        child["scaling"] = self.scaling_inverse_activation(scaling_sel.repeat(N, 1) / (0.8 * N))
        extra = {k: torch.cat((self._p[k].data.index_select(0, idx_c), child[k]), dim=0) for k in PARAM_NAMES}
        opacity_all = torch.cat((self._p["opacity"].data, extra["opacity"]), dim=0)
        remove = (self.opacity_activation(opacity_all) < min_opacity).squeeze(-1)
        if max_screen_size:
            # (the screen radii are all zero at this point of the reference's sequence: every append resets them)
            scaling_all = torch.cat((self._p["scaling"].data, extra["scaling"]), dim=0)
            remove = remove | (self.scaling_activation(scaling_all).max(dim=1).values > 0.1 * extent)
        remove[:sel_split.shape[0]] |= sel_split
        keep = (~remove).nonzero(as_tuple=False).squeeze(1)
        self._rebuild(keep, extra, append_first=True)
        dev, n = self._p["xyz"].device, self.num_points
        self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        self.denom = torch.zeros(n, 1, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    @torch.no_grad()
    def densify_and_prune_stepwise(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """The reference's own sequence of four rebuilds (kept as the statement densify_and_prune is tested against)."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        prune_mask = (self.get_opacity < min_opacity).squeeze(-1)
        if max_screen_size:
            prune_mask = prune_mask | (self.max_radii2D > max_screen_size) | \
                (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune_mask)

    @torch.no_grad()
    def reset_opacity(self):
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        group = next(g for g in self.optimizer.param_groups if g.get("name") == "opacity")
        old = group["params"][0]
        state = self.optimizer.state.pop(old, None)
        p = nn.Parameter(new.contiguous().requires_grad_(True))
        if state is not None and "exp_avg" in state:
            state["exp_avg"] = torch.zeros_like(new)
            state["exp_avg_sq"] = torch.zeros_like(new)
            self.optimizer.state[p] = state
        group["params"][0] = p
        self._p["opacity"] = p
        if hasattr(self.optimizer, "invalidate"):
            self.optimizer.invalidate()

    def per_gaussian_parameters(self):
        return [self._p[k] for k in PARAM_NAMES]
