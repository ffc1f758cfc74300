"""One adaptation train step of InsTaG's face branch on synthetic frames, single- or multi-GPU.

Counterpart of /root/reference/train_face.py:110-788 restricted to the hot path: render_motion
(:346-350) -> L1 + 0.2*(1-SSIM) (:450-456) + regularisers (:508-540) -> backward (:625) ->
densification statistics (:675-686) -> AdamW / Adam steps (:781-788).  LPIPS, the few-shot
normal/depth priors and logging are out of scope.

Data parallelism over frames (an addition): identical replicas, rank r renders its
own frame, gradients of [Gaussians | UMF | PMF] are flattened into ONE bucket and all-reduced
(RCCL over xGMI; gloo in the CPU tests), densification statistics are all-reduced too, so every
replica applies identical optimizer and densify/prune decisions.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.distributed as dist

from .gaussian_model import GaussianModel, OptimizationParams
from .losses import l1_and_ssim


@dataclass
class Frame:
    """What the reference keeps per camera: matrices + talking_dict (scene/cameras.py, dataset_readers.py)."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor
    talking_dict: dict
    original_image: torch.Tensor      # [3,H,W] in [0,1]


def make_frame(cam, frame_data) -> Frame:
    td = dict(auds=frame_data["auds"], au_exp=frame_data["au_exp"], face_mask=frame_data["face_mask"],
              hair_mask=frame_data["hair_mask"], mouth_mask=frame_data["mouth_mask"])
    return Frame(cam.image_height, cam.image_width, cam.FoVx, cam.FoVy, cam.world_view_transform,
                 cam.full_proj_transform, cam.camera_center, td, frame_data["gt_image"])


def flat_grad_bucket(params: List[torch.Tensor]) -> torch.Tensor:
    """Concatenate gradients (zeros where .grad is None) into one contiguous fp32 bucket."""
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def scatter_grad_bucket(params: List[torch.Tensor], bucket: torch.Tensor):
    o = 0
    for p in params:
        n = p.numel()
        g = bucket[o:o + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n


def allreduce_gradients(params: List[torch.Tensor], extras: Optional[List[torch.Tensor]] = None, average=True):
    """One fused-bucket all-reduce(SUM) of all gradients (+ extra stat tensors, summed not averaged)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    extras = extras or []
    g = flat_grad_bucket(params)
    n_g = g.numel()
    bucket = torch.cat([g] + [e.reshape(-1).to(g.dtype) for e in extras])
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    if average:
        bucket[:n_g] /= world
    scatter_grad_bucket(params, bucket[:n_g])
    o = n_g
    for e in extras:
        e.copy_(bucket[o:o + e.numel()].view_as(e))
        o += e.numel()


class FaceTrainer:
    """Holds the Gaussians, the UMF (motion_net) and the PMF (gaussians.neural_motion_grid) and steps them."""

    def __init__(self, gaussians: GaussianModel, motion_net, background, opt=OptimizationParams,
                 cameras_extent: float = 0.2, densify: bool = True, seed: int = 0):
        self.g = gaussians
        self.motion_net = motion_net
        self.bg = background
        self.opt = opt
        self.extent = cameras_extent
        self.densify = densify
        self.iteration = 0
        dev = gaussians.get_xyz.device
        self.gen = torch.Generator(device=dev).manual_seed(seed)     # identical on every rank
        fused = dev.type == "cuda"
        # train_face.py:59-60: AdamW(betas .9/.99, eps 1e-8, wd .01), lr x0.1 during warm-up then 0.5^(it/iters)
        self.motion_optimizer = torch.optim.AdamW(motion_net.get_params(5e-3, 5e-4), betas=(0.9, 0.99), eps=1e-8,
                                                  weight_decay=0.01, **({"fused": True} if fused else {}))
        warm_step, iters = 3000, opt.iterations
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.motion_optimizer, lambda it: 0.1 if it < warm_step else 0.5 ** (it / iters))
        self.g.training_setup(opt)
        self.last = {}

    def _all_params(self):
        ps = self.g.per_gaussian_parameters()
        ps += [p for p in self.motion_net.parameters()]
        if self.g.neural_motion_grid is not None:
            ps += [p for p in self.g.neural_motion_grid.parameters()]
        return ps

    def loss_fn(self, frame: Frame, pkg, warm: bool):
        dev = self.bg.device
        td = frame.talking_dict
        face_mask = td["face_mask"].to(dev)
        hair_mask = td["hair_mask"].to(dev)
        mouth_mask = td["mouth_mask"].to(dev)
        head_mask = face_mask | hair_mask
        image, alpha = pkg["render"], pkg["alpha"]
        gt = frame.original_image.to(dev)
        gt_white = gt * head_mask + self.bg[:, None, None] * ~head_mask
        gt_white = torch.where(mouth_mask[None], self.bg[:, None, None].expand_as(gt_white), gt_white)
        Ll1, ssim_val = l1_and_ssim(image, gt_white)
        loss = Ll1 + self.opt.lambda_dssim * (1.0 - ssim_val)
        if warm:
            m, pm = pkg["motion"], pkg["p_motion"]
            loss = loss + 1e-5 * (m["d_xyz"].abs().mean() + m["d_rot"].abs().mean() + m["d_opa"].abs().mean()
                                  + m["d_scale"].abs().mean() + pm["p_xyz"].abs().mean())
            loss = loss + 1e-3 * (((1 - alpha) * head_mask).mean() + (alpha * ~head_mask).mean())
            attn = pkg["attn"]
            loss = loss + 1e-4 * (attn[1][hair_mask].mean() + attn[0][hair_mask].mean())
        return loss, Ll1

    def step(self, frame: Frame, sync_stats: bool = False):
        from .renderer import render_motion
        self.iteration += 1
        it = self.iteration
        self.g.update_learning_rate(it)
        pkg = render_motion(frame, self.g, self.motion_net, None, self.bg, return_attn=True, personalized=False,
                            align=True)
        loss, Ll1 = self.loss_fn(frame, pkg, warm=True)
        loss.backward()

        with torch.no_grad():
            vis = pkg["visibility_filter"]
            radii = pkg["radii"].to(self.g.max_radii2D.dtype)
            vs_grad = pkg["viewspace_points"].grad
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # statistics become the sum over ranks; gradients the mean (== accumulation over the frames)
                norm = torch.norm(vs_grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
                cnt = vis[:, None].to(norm.dtype)
                allreduce_gradients(self._all_params(), extras=[norm, cnt])
                rmax = torch.where(vis, radii, torch.zeros_like(radii))
                dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
                self.g.max_radii2D = torch.max(self.g.max_radii2D, rmax)
                self.g.xyz_gradient_accum += norm
                self.g.denom += cnt
            else:
                self.g.max_radii2D = torch.where(vis, torch.max(self.g.max_radii2D, radii), self.g.max_radii2D)
                self.g.add_densification_stats(vs_grad, vis)

            self.motion_optimizer.step()
            self.g.optimizer.step()
            self.motion_optimizer.zero_grad(set_to_none=True)
            self.g.optimizer.zero_grad(set_to_none=True)
            self.scheduler.step()

            if self.densify and it < self.opt.densify_until_iter and it > self.opt.densify_from_iter \
                    and it % self.opt.densification_interval == 0:
                size_threshold = 20 if it > self.opt.opacity_reset_interval else None
                self.g.densify_and_prune(self.opt.densify_grad_threshold, 0.05 + 0.25 * it / self.opt.densify_until_iter,
                                         self.extent, size_threshold, generator=self.gen)
        self.last = dict(loss=loss.detach(), l1=Ll1.detach(), num_points=self.g.num_points)
        return self.last


def build_trainer(n_gaussians, device, sh_degree=1, seed=0, densify=False, encoder_cls=None, raw=None):
    """Synthetic config-C3 trainer: N Gaussians + PMF + UMF with random-init weights."""
    from .motion_net import MotionNetwork, PersonalizedMotionNetwork
    from .scene_synth import synthetic_gaussians
    torch.manual_seed(seed)
    args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    pmf = PersonalizedMotionNetwork(args=args, encoder_cls=encoder_cls).to(device)
    umf = MotionNetwork(args=args, encoder_cls=encoder_cls).to(device)
    g = GaussianModel(sh_degree, neural_motion_grid=pmf)
    g.load_raw(raw if raw is not None else synthetic_gaussians(n_gaussians, sh_degree=sh_degree, seed=seed), device)
    bg = torch.tensor([0.0, 1.0, 0.0], device=device)
    return FaceTrainer(g, umf, bg, densify=densify, seed=seed)
