"""Train steps of the two stages that follow the face branch: the mouth branch and the face+mouth fuse stage.

Counterparts of /root/reference/train_mouth.py:106-293 and /root/reference/train_fuse_con.py:75-245 restricted to
the hot path (render -> loss -> backward -> statistics / density control -> optimizers).  Same kernels as the face
branch behind other callers: ``render_motion_mouth_con`` / ``render_motion`` of
instag_amd/renderer.py, the fused L1+SSIM operator, the single-launch Adam.  Frame selection by AU25, LPIPS,
logging and checkpoint cadence are the reference's data pipeline / control plane and stay out.
"""
from __future__ import annotations

import random
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .gaussian_model import GaussianModel, OptimizationParams, sh_to_rgb
from .losses import l1_and_ssim
from .train import Frame

GEOMETRY = ("xyz", "opacity", "scaling", "rotation")


def _lips_mask(like: torch.Tensor, lips_rect) -> torch.Tensor:
    """lips_mask[xmin:xmax, ymin:ymax] = True (train_mouth.py:168-170: the rect indexes rows first).  A tensor
    rect stays on its device (comparisons against row / column indices: no host round trip, capturable)."""
    H, W = like.shape[-2:]
    if torch.is_tensor(lips_rect):
        lr = lips_rect.to(device=like.device, dtype=torch.int64)
        rows = torch.arange(H, device=like.device)[:, None]
        cols = torch.arange(W, device=like.device)[None, :]
        return (rows >= lr[0]) & (rows < lr[1]) & (cols >= lr[2]) & (cols < lr[3])
    r0, r1, c0, c1 = [int(v) for v in lips_rect]
    m = torch.zeros(H, W, dtype=torch.bool, device=like.device)
    m[r0:r1, c0:c1] = True
    return m


def mouth_loss(image, alpha, gt, mouth_mask, lips_mask, bg, p_xyz=None, warm=True, lambda_dssim=0.2):
    """Loss block of the mouth branch (train_mouth.py:186-221) -> (loss, Ll1).  Outside the lips rectangle the
    mouth-mask fringe of the render is painted with the background; the target shows the image inside the mouth mask
    only.  ``warm`` (iteration > warm_step) adds the alignment and alpha terms."""
    bg3 = bg[:, None, None]
    gt_green = torch.where(mouth_mask[None], gt, bg3.expand_as(gt))
    image_green = torch.where((lips_mask ^ mouth_mask)[None], bg3.expand_as(image), image)
    Ll1, s = l1_and_ssim(image_green, gt_green)
    loss = Ll1 + lambda_dssim * (1.0 - s)
    if warm:
        lm = lips_mask.to(alpha.dtype)
        if p_xyz is not None:
            loss = loss + 1e-5 * p_xyz.abs().mean()
        loss = loss + 1e-3 * (((1 - alpha) * lm).mean() + (alpha * (1 - lm)).mean())
    return loss, Ll1


def fuse_loss(image, gt, lambda_dssim=0.2):
    """train_fuse_con.py:176-181 (iteration >= bg_iter = 0, always): whole-frame L1 + DSSIM -> (loss, Ll1)."""
    if image.is_cuda and image.dim() == 3 and image.shape[0] == 3:
        from .losses import plain_loss_fused
        return plain_loss_fused(image, gt, lambda_dssim)
    Ll1, s = l1_and_ssim(image, gt)
    return Ll1 + lambda_dssim * (1.0 - s), Ll1


@dataclass(frozen=True)
class MouthPhase:
    align: bool = True
    warm: bool = True
    late: bool = False        # iteration > bg_iter: black background, geometry and the motion field frozen


def mouth_phase(iteration: int, opt=OptimizationParams, warm_step: int = 3000,
                bg_iter: Optional[int] = None) -> MouthPhase:
    """train_mouth.py:48-52, 152-196 (mode_long = False): bg_iter = motion_stop_iter = iterations - 1000."""
    bg_iter = opt.iterations - 1000 if bg_iter is None else bg_iter
    align = iteration > 1000 if iteration < warm_step else True
    return MouthPhase(align=align, warm=iteration > warm_step, late=iteration > bg_iter)


def _make_optimizers(gaussians: GaussianModel, motion_net, opt, on_gpu: bool):
    groups = motion_net.get_params(5e-3, 5e-4) if motion_net is not None else None
    motion_opt = None
    if groups is not None:
        if on_gpu:
            from .optim import MultiTensorAdam
            motion_opt = MultiTensorAdam(groups, lr=5e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01,
                                         decoupled=True)
        else:
            motion_opt = torch.optim.AdamW(groups, lr=5e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    gaussians.training_setup(opt, fused=on_gpu)
    return motion_opt


def _push_lrs(*optimizers):
    for o in optimizers:
        if o is not None and hasattr(o, "set_lrs"):
            o.set_lrs()


def _combine(*optimizers):
    """One launch (and one learning-rate upload) for several fused optimizers -- the reference steps them back to back
    (train_mouth.py:286-291, train_fuse_con.py:236-240); None when one of them is not the fused kind (CPU tests)."""
    from .optim import CombinedAdam, MultiTensorAdam
    if all(isinstance(o, MultiTensorAdam) for o in optimizers):
        return CombinedAdam(list(optimizers))
    return None


class GraphedStage:
    """A stage trainer's whole step (forward, loss, backward, statistics, optimizers) captured once into a hipGraph
    and replayed: ``body(frame) -> (outputs..., keepalive)`` must be free of host round trips; the rasterizer runs in
    its capacity mode with one slot per rasterizer call of the step, sized from eager warm-up steps
    (instag_amd/diff_gauss.py:CapacityPlan)."""

    def __init__(self, body, example: Frame, device, headroom: float = 1.5, warmup_steps: int = 2, pre=None):
        """``pre()`` = the host-side part of a step (iteration counter, learning-rate table): run before every
        warm-up step, never captured."""
        from . import diff_gauss
        from .train import _no_gc
        assert device.type == "cuda", "graph mode needs the GPU"
        import sys
        self.static = example.clone_static()
        diff_gauss.set_capacity_plan(None)
        counts = None
        for _ in range(max(1, warmup_steps)):
            diff_gauss.RENDERED_LOG.clear()
            if pre is not None:
                pre()
            body(self.static)
            got = list(diff_gauss.RENDERED_LOG)
            counts = got if counts is None else [max(a, b) for a, b in zip(counts, got)]
        self.capacities = [int(r * headroom) + 4096 for r in counts]
        self.plan = diff_gauss.CapacityPlan(self.capacities, device)
        diff_gauss.set_capacity_plan(self.plan)
        side = _lib.warmup_stream(device)              # allocator / library warm-up in capacity mode
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self.plan.begin_step()
                if pre is not None:
                    pre()
                body(self.static)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        self.plan.begin_step()
        with _no_gc(), _lib.graph_capture(self.graph):
            try:
                out = body(self.static)
            except BaseException:
                # an operator that is not capturable raised: report it before the capture is torn down (ending an
                # invalidated capture can crash the process on ROCm 7.2, which would hide the message)
                import traceback
                traceback.print_exc()
                sys.stderr.flush()
                raise
        # body returns (..., keepalive): the render package stays referenced until the capture has ended (ROCm 7.2:
        # releasing it inside the capture window intermittently crashes hipStreamEndCapture), then it is dropped
        # (detached: a retained loss would keep the step's autograd graph -- and every parameter's AccumulateGrad node,
        # bound to the capture stream -- alive into the next backward on another stream)
        self.out = tuple(o.detach() if torch.is_tensor(o) else o for o in out[:-1])
        del out

    CHECK_EVERY = 64         # replays between two looks at the (sticky, device-side) overflow flags

    def replay(self, frame: Frame):
        self.static.copy_from(frame)
        self.graph.replay()
        self._replays = getattr(self, "_replays", 0) + 1
        return self.out

    def check_overflow(self):
        return self.plan.overflowed()

    def overflow_due(self):
        """True every CHECK_EVERY replays if some rasterizer call of a replayed step needed more instances than its
        capacity (the flag is sticky on the device, one synchronising read per CHECK_EVERY steps): the caller drops the
        graph -- steps run eagerly, or are captured again with larger capacities."""
        if getattr(self, "_replays", 0) < self.CHECK_EVERY:
            return False
        self._replays = 0
        return bool(self.plan.poll_overflow())        # asynchronous: the answer of the previous poll, no device wait


_ONE = {}


def _backward(loss, device):
    """loss.backward() with the MLPs' weight-gradient GEMMs batched into one launch behind it (instag_amd/deferred.py);
    on the device the root gradient is a cached constant (no fill launch per step)."""
    from .deferred import deferred_grads
    with deferred_grads(device if device.type == "cuda" else None):
        if device.type == "cuda" and loss.dim() == 0:
            key = (loss.device, loss.dtype)
            one = _ONE.get(key)
            if one is None:
                if torch.cuda.is_current_stream_capturing():
                    one = torch.ones((), dtype=loss.dtype, device=loss.device)
                else:
                    one = _ONE[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
            loss.backward(gradient=one)
        else:
            loss.backward()


def _drop_graph(trainer):
    if getattr(trainer, "_graph", None) is not None:
        from . import diff_gauss
        diff_gauss.set_capacity_plan(None)
    trainer._graph = None
    trainer._graph_key = None


class MouthTrainer:
    """One iteration of train_mouth.py: the mouth Gaussians + MouthMotionNetwork are optimised, the (trained) face
    Gaussians + face field only supply the jaw-movement feature."""

    def __init__(self, gaussians: GaussianModel, motion_net, gaussians_face: GaussianModel, motion_net_face,
                 background, opt=OptimizationParams, cameras_extent: float = 0.2, densify: bool = True, seed: int = 0,
                 warm_step: int = 3000, bg_iter: Optional[int] = None):
        self.bg_iter = bg_iter
        self.g, self.motion_net = gaussians, motion_net
        self.g_face, self.motion_net_face = gaussians_face, motion_net_face
        self.bg = background
        self.opt, self.extent, self.densify, self.warm_step = opt, cameras_extent, densify, warm_step
        self.device = gaussians.get_xyz.device
        self.on_gpu = self.device.type == "cuda"
        self.iteration = 0
        self.rng = random.Random(seed)                                       # k = randint(10, 50), train_mouth.py:175
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.motion_optimizer = _make_optimizers(gaussians, motion_net, opt, self.on_gpu)
        self._combined = _combine(self.motion_optimizer, self.g.optimizer) if self.on_gpu else None
        self._base_lr = [float(g["lr"]) for g in self.motion_optimizer.param_groups]
        self._graph = None
        self._graph_key = None
        # the selection size of a captured step lives on the device; it rides on the optimizers' learning-rate upload
        # (MultiTensorAdam.reserve_extra_i64) instead of a fill launch per step -- `_k_own` until that table exists
        self._k_own = torch.full((1,), 10, dtype=torch.int64, device=self.device) if self.on_gpu else None
        if self._combined is not None:
            self._combined.reserve_extra_i64(1)
        self.last = {}

    @property
    def _k_dev(self):
        view = self._combined.extra_i64() if self._combined is not None else None
        return view if view is not None else self._k_own

    def _stage_k(self, k):
        """Call in FRONT of _set_learning_rates (whose upload carries the value)."""
        view = self._combined.extra_i64() if self._combined is not None else None
        if view is not None:
            self._combined.set_extra_i64([k])
        elif self._k_own is not None:
            self._k_own.fill_(k)

    def _set_learning_rates(self, it):
        f = 0.1 if (it - 1) < self.warm_step else 0.5 ** ((it - 1) / self.opt.iterations)     # LambdaLR, :64
        for grp, base in zip(self.motion_optimizer.param_groups, self._base_lr):
            grp["lr"] = base * f
        self.g.update_learning_rate(it)
        if self._combined is not None:
            _push_lrs(self._combined)
        else:
            _push_lrs(self.motion_optimizer, self.g.optimizer)

    def _freeze_late(self):
        """train_mouth.py:189-196: after bg_iter the motion field and the Gaussians' geometry stop learning."""
        for p in self.motion_net.parameters():
            p.requires_grad_(False)
        for k in GEOMETRY:
            self.g._p[k].requires_grad_(False)      # (every late iteration: density control rebuilds the leaves)

    def forward(self, frame: Frame, phase: MouthPhase, k: int):
        from .renderer import render_motion_mouth_con
        bg = torch.zeros_like(self.bg) if phase.late else self.bg
        pkg = render_motion_mouth_con(frame, self.g, self.motion_net, self.g_face, self.motion_net_face, None, bg,
                                      personalized=False, align=phase.align, k=k)
        td = frame.talking_dict
        dev = self.device
        mouth = td["mouth_mask"].to(dev)
        want_p = phase.warm and pkg["p_motion"] is not None
        if self.on_gpu and torch.is_tensor(td["lips_rect"]) and pkg["render"].shape[0] == 3:
            from .losses import mouth_loss_fused
            p_raw = dict.get(pkg["p_motion"], "_p") if want_p else None
            p_xyz = pkg["p_motion"]["p_xyz"] if (want_p and p_raw is None) else None
            loss, Ll1 = mouth_loss_fused(pkg["render"], pkg["alpha"], frame.original_image.to(dev), mouth,
                                         td["lips_rect"].to(dev), bg, p_xyz, warm=phase.warm,
                                         lambda_dssim=self.opt.lambda_dssim, p_raw=p_raw)
            return pkg, loss, Ll1
        p_xyz = pkg["p_motion"]["p_xyz"] if want_p else None
        lips = _lips_mask(mouth, td["lips_rect"])
        loss, Ll1 = mouth_loss(pkg["render"], pkg["alpha"], frame.original_image.to(dev), mouth, lips, bg, p_xyz,
                               warm=phase.warm, lambda_dssim=self.opt.lambda_dssim)
        return pkg, loss, Ll1

    def _stats_on(self, it):
        return self.densify and it < self.opt.densify_until_iter

    def _density_due(self, it):
        o = self.opt
        return self._stats_on(it) and ((it > o.densify_from_iter and it % o.densification_interval == 0)
                                       or it % o.opacity_reset_interval == 0)

    @torch.no_grad()
    def _accumulate_stats(self, pkg):
        """train_mouth.py:262-263 (device-only: part of a captured step)."""
        vis = pkg["visibility_filter"]
        radii = pkg["radii"].to(self.g.max_radii2D.dtype)
        self.g.max_radii2D.copy_(torch.max(self.g.max_radii2D, torch.where(vis, radii, torch.zeros_like(radii))))
        self.g.add_densification_stats(pkg["viewspace_points"].grad, vis)

    @torch.no_grad()
    def _density_control(self, it, frame: Frame):
        """train_mouth.py:265-283, behind the statistics of this iteration."""
        o = self.opt
        if not self._stats_on(it):
            return
        if it > o.densify_from_iter and it % o.densification_interval == 0:
            size_threshold = 20 if it > o.opacity_reset_interval else None
            self.g.densify_and_prune(o.densify_grad_threshold, 0.05 + 0.25 * it / o.densify_until_iter, self.extent,
                                     size_threshold, generator=self.gen)
            if it > 2000:
                # Gaussians that took the background's green are pushed towards removal (:276-279)
                rgb = sh_to_rgb(self.g.active_sh_degree, self.g.get_features, self.g.get_xyz,
                                frame.camera_center.to(self.device))
                green = (rgb[:, 0] < 100 / 255) & (rgb[:, 1] > 180 / 255) & (rgb[:, 2] < 100 / 255)
                self.g.xyz_gradient_accum[green] /= 2
                self.g._opacity.data[green] = self.g.inverse_opacity_activation(
                    torch.ones_like(self.g._opacity.data[green]) * 0.1)
                self.g._scaling.data[green] /= 10
        if it % o.opacity_reset_interval == 0:
            self.g.reset_opacity()

    def _step_optimizers(self):
        if self._combined is not None:
            self._combined.step()
            return
        self.motion_optimizer.step()
        self.g.optimizer.step()

    def _zero_grad(self):
        self.motion_optimizer.zero_grad(set_to_none=True)
        self.g.optimizer.zero_grad(set_to_none=True)

    def _body(self, frame: Frame, phase: MouthPhase, k, stats_on: bool):
        """Everything of an iteration without a density-control event; free of host round trips when `k` is a
        device tensor (the captured form)."""
        from .losses import defer_finalize
        with defer_finalize():          # (backward follows at once; the loss value is read after the step)
            pkg, loss, Ll1 = self.forward(frame, phase, k)
        _backward(loss, self.device)
        if stats_on:
            self._accumulate_stats(pkg)
        self._step_optimizers()
        self._zero_grad()
        return loss, Ll1, pkg

    def _key(self, it):
        return (mouth_phase(it, self.opt, self.warm_step, self.bg_iter), self._stats_on(it), self.g.active_sh_degree)

    def enable_graph(self, example: Frame, headroom: float = 1.5, warmup_steps: int = 2):
        """Capture the step of the NEXT iterations' phase (the warm-up steps are real steps: they advance the
        iteration counter).  step() falls back to eager launches on density-control iterations and drops the graph
        when the phase or the parameter set changes."""
        _drop_graph(self)
        total = max(1, warmup_steps) + 2
        key = self._key(self.iteration + total + 1)
        phase, stats_on, _ = key
        if phase.late:
            self._freeze_late()

        def pre():
            self.iteration += 1
            self._stage_k(self.rng.randint(10, 50))
            self._set_learning_rates(self.iteration)

        def body(frame):
            return self._body(frame, phase, self._k_dev, stats_on)
        self._graph = GraphedStage(body, example, self.device, headroom, warmup_steps, pre=pre)
        self._graph_key = key
        return self._graph

    def step(self, frame: Frame):
        self.iteration += 1
        it = self.iteration
        k = self.rng.randint(10, 50)
        if self._graph is not None:
            self._stage_k(k)                    # (travels with the learning rates)
        self._set_learning_rates(it)
        if it % 1000 == 0:
            self.g.oneupSHdegree()                                                                # train_mouth.py:110-111
        phase = mouth_phase(it, self.opt, self.warm_step, self.bg_iter)
        if phase.late:
            self._freeze_late()
        due = self._density_due(it)
        steps = it < self.opt.iterations
        if self._graph is not None and (due or not steps or self._key(it) != self._graph_key):
            _drop_graph(self)
        if self._graph is not None:
            loss, Ll1 = self._graph.replay(frame)[:2]
            if self._graph.overflow_due():
                _drop_graph(self)               # the scene outgrew the captured capacities: eager launches from here on
        else:
            from . import diff_gauss
            if diff_gauss._CAPACITY_PLAN is not None:
                diff_gauss._CAPACITY_PLAN.begin_step()
            pkg, loss, Ll1 = self.forward(frame, phase, k)
            _backward(loss, self.device)
            if self._stats_on(it):
                self._accumulate_stats(pkg)
                self._density_control(it, frame)
            if steps:
                self._step_optimizers()
                self._zero_grad()
        self.last = dict(loss=loss.detach(), l1=Ll1.detach(), num_points=self.g.num_points, phase=phase, k=k)
        return self.last


class FuseTrainer:
    """One iteration of train_fuse_con.py: face and mouth are rendered, composited over the per-camera background
    and compared with the whole frame; both motion fields and most of the geometry are frozen from the first
    iteration (bg_iter = 0), so the step tunes colours (both models) and the face's opacity."""

    FROZEN_FACE = ("xyz", "scaling", "rotation")
    FROZEN_MOUTH = ("xyz", "opacity", "scaling", "rotation")

    def __init__(self, gaussians: GaussianModel, motion_net, gaussians_mouth: GaussianModel, motion_net_mouth,
                 background, opt=OptimizationParams, seed: int = 0):
        self.g, self.motion_net = gaussians, motion_net
        self.g_mouth, self.motion_net_mouth = gaussians_mouth, motion_net_mouth
        self.bg, self.opt = background, opt
        self.device = gaussians.get_xyz.device
        self.on_gpu = self.device.type == "cuda"
        self.iteration = 0
        gaussians.training_setup(opt, fused=self.on_gpu)
        gaussians_mouth.training_setup(opt, fused=self.on_gpu)
        self._combined = _combine(gaussians.optimizer, gaussians_mouth.optimizer) if self.on_gpu else None
        for net in (motion_net, motion_net_mouth):
            for p in net.parameters():
                p.requires_grad_(False)
        for k in self.FROZEN_FACE:
            gaussians._p[k].requires_grad_(False)
        for k in self.FROZEN_MOUTH:
            gaussians_mouth._p[k].requires_grad_(False)
        self._graph = None
        self._graph_key = None
        self.last = {}

    def forward(self, frame: Frame):
        from .renderer import render_fuse
        dev = self.device
        scene_bg = frame.talking_dict.get("background")
        out = render_fuse(frame, self.g, self.motion_net, self.g_mouth, self.motion_net_mouth, None, self.bg,
                          scene_background=None if scene_bg is None else scene_bg.to(dev))
        loss, Ll1 = fuse_loss(out["image"], frame.original_image.to(dev), self.opt.lambda_dssim)
        return out, loss, Ll1

    def _set_learning_rates(self, it):
        self.g.update_learning_rate(it)           # train_fuse_con.py:85 (the mouth model keeps its initial rates)
        _push_lrs(self._combined if self._combined is not None else self.g.optimizer)

    def _body(self, frame: Frame):
        from .losses import defer_finalize
        with defer_finalize():          # (backward follows at once; the loss value is read after the step)
            out, loss, Ll1 = self.forward(frame)
        _backward(loss, self.device)
        if self._combined is not None:
            self._combined.step()
        else:
            self.g.optimizer.step()
            self.g_mouth.optimizer.step()
        self.g.optimizer.zero_grad(set_to_none=True)
        self.g_mouth.optimizer.zero_grad(set_to_none=True)
        return loss, Ll1, out["image"], out

    def enable_graph(self, example: Frame, headroom: float = 1.5, warmup_steps: int = 2):
        """Capture the whole step (the stage has a single phase and no density control: densify_until_iter = 0)."""
        _drop_graph(self)

        def pre():
            self.iteration += 1
            self._set_learning_rates(self.iteration)
        self._graph = GraphedStage(self._body, example, self.device, headroom, warmup_steps, pre=pre)
        return self._graph

    def step(self, frame: Frame):
        self.iteration += 1
        it = self.iteration
        self._set_learning_rates(it)
        if self._graph is not None and it >= self.opt.iterations:
            _drop_graph(self)
        if self._graph is not None:
            loss, Ll1, image = self._graph.replay(frame)[:3]
            if self._graph.overflow_due():
                _drop_graph(self)
        elif it < self.opt.iterations:
            from . import diff_gauss
            if diff_gauss._CAPACITY_PLAN is not None:
                diff_gauss._CAPACITY_PLAN.begin_step()
            loss, Ll1, image = self._body(frame)[:3]
        else:
            out, loss, Ll1 = self.forward(frame)         # last iteration: no optimizer step (:242)
            _backward(loss, self.device)
            self.g.optimizer.zero_grad(set_to_none=True)
            self.g_mouth.optimizer.zero_grad(set_to_none=True)
            image = out["image"]
        self.last = dict(loss=loss.detach(), l1=Ll1.detach(), image=image.detach())
        return self.last
