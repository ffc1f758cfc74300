"""One adaptation train step of InsTaG's face branch on synthetic frames, single- or multi-GPU.

Counterpart of /root/reference/train_face.py:110-788 restricted to the hot path: render_motion
(:346-350) -> L1 + 0.2*(1-SSIM) (:450-456) + regularisers (:508-540) -> backward (:625) ->
densification statistics / density control (:667-746) -> AdamW / Adam steps (:781-788), in the reference's
iteration-dependent phases (face_phase: alignment, warm terms, hair iterations, monocular normal / depth
priors :458-504).  LPIPS and logging are out of scope.

Two execution modes with identical arithmetic:
  * eager  -- every operator is launched from Python (one host round trip per rasterizer pass);
  * graph  -- the whole step (forward, backward, statistics, optimizers) is captured ONCE into a
    hipGraph (torch.cuda.CUDAGraph) and replayed; the rasterizer runs in its sync-free capacity mode
    (diff_gauss.CapacityPlan), per-frame inputs are copied into static device buffers, learning rates
    live in device scalars.  The step is launch-bound in eager mode (~590 kernels), so this is the
    MI355X-native way to run it.

Data parallelism over frames (an addition): identical replicas, rank r renders its
own frame, gradients of [Gaussians | UMF | PMF] are flattened into ONE bucket and all-reduced
(RCCL over xGMI; gloo in the CPU tests); densification statistics stay per-rank sums / maxima and are
exchanged when a densification reads them, so every replica applies identical optimizer and
densify/prune decisions with one collective per step.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import os
import torch
import torch.distributed as dist

from . import _lib
from .gaussian_model import GaussianModel, OptimizationParams
from .losses import face_loss


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

    TENSORS = ("world_view_transform", "full_proj_transform", "camera_center", "original_image")
    DICT_TENSORS = ("auds", "au_exp", "face_mask", "hair_mask", "mouth_mask", "lips_rect")
    # present only for the stages that read them: monocular normal [3,H,W] / depth [H,W] priors
    # (train_face.py:466-504), per-camera scene background [3,H,W] in [0,1] (train_fuse_con.py:113)
    OPTIONAL_DICT_TENSORS = ("normal", "depth", "background")

    def _dict_keys(self):
        return self.DICT_TENSORS + tuple(k for k in self.OPTIONAL_DICT_TENSORS if self.talking_dict.get(k) is not None)

    def _named_tensors(self):
        return [(k, getattr(self, k)) for k in self.TENSORS] + [(k, self.talking_dict[k]) for k in self._dict_keys()]

    def packed(self, device=None, pin: bool = False) -> "Frame":
        """Copy whose tensors are views into ONE byte buffer, so that feeding a frame to a captured step is a
        single device copy instead of one per tensor.  ``pin`` (host copies): page-locked memory, so that the upload
        of the whole frame is one asynchronous copy (HostFrameFeeder)."""
        named = self._named_tensors()
        device = device if device is not None else named[0][1].device
        offs, total = [], 0
        for _, t in named:
            offs.append(total)
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        buf = torch.zeros(total, dtype=torch.uint8, device=device)
        if pin and buf.device.type == "cpu":
            buf = buf.pin_memory()
        views = {}
        for (k, t), o in zip(named, offs):
            nbytes = t.numel() * t.element_size()
            v = buf[o:o + nbytes].view(t.dtype).view(t.shape)
            v.copy_(t)
            views[k] = v
        td = {k: views[k] for k in self._dict_keys()}
        f = Frame(self.image_height, self.image_width, self.FoVx, self.FoVy, views["world_view_transform"],
                  views["full_proj_transform"], views["camera_center"], td, views["original_image"])
        f._buf = buf
        f._layout = tuple((k, tuple(t.shape), t.dtype) for k, t in named)
        return f

    def clone_static(self):
        return self.packed()

    def copy_from(self, other: "Frame"):
        assert (self.image_height, self.image_width) == (other.image_height, other.image_width)
        assert abs(self.FoVx - other.FoVx) < 1e-12 and abs(self.FoVy - other.FoVy) < 1e-12, \
            "graph mode bakes the field of view into the captured launches"
        if getattr(self, "_buf", None) is not None and getattr(other, "_buf", None) is not None \
                and self._layout == other._layout:
            self._buf.copy_(other._buf, non_blocking=True)
            return
        for k in self.TENSORS:
            getattr(self, k).copy_(getattr(other, k), non_blocking=True)
        for k in self._dict_keys():
            self.talking_dict[k].copy_(other.talking_dict[k], non_blocking=True)


class HostFrameFeeder:
    """Frames that live in (pinned) HOST memory, uploaded ahead of the step that reads them.

    The reference uploads a frame's tensors inside the iteration that uses them (train_face.py:324-327 masks and image,
    gaussian_renderer/__init__.py:188-189 audio window and expression vector: a handful of blocking ``.cuda()`` copies,
    ~4 MB at 512x512).  Here a frame is ONE packed byte buffer (Frame.packed): while step i runs, frame i+1 travels on a
    second stream into one of ``slots`` device staging buffers, and step i+1's first action -- the device-to-device copy
    into the captured step's static frame -- reads the staging buffer.

    All ordering is done on the HOST (event.synchronize(), no stream-to-stream waits): a cross-stream wait is a barrier
    packet in front of the replayed graph, and two of them per step cost 2.6 % of the step (scripts/probes/
    host_frames_probe2.py: 0.933 -> 0.958 ms), more than the upload itself, which overlaps (0.938 ms).  The host runs a few
    steps ahead of the device in replay mode, so a staging slot is reused only ``slots`` steps later and the host-side
    wait for its last reader returns at once."""

    def __init__(self, example: Frame, device, slots: int = 4):
        self.device = torch.device(device)
        self.slots = int(slots)
        self.stage = [example.packed(self.device) for _ in range(self.slots)]
        # (no stream of its own: the capture warm-up stream exists anyway and is idle outside enable_graph)
        self.stream = _lib.warmup_stream(self.device)
        self.ready = [torch.cuda.Event() for _ in range(self.slots)]
        self.consumed = [torch.cuda.Event() for _ in range(self.slots)]
        for e in self.consumed:
            e.record(torch.cuda.current_stream(self.device))

    @staticmethod
    def to_host(frame: Frame) -> Frame:
        return frame.packed("cpu", pin=True)

    def prefetch(self, host_frame: Frame, slot: int):
        """Start the upload of ``host_frame`` into staging slot ``slot`` (once the slot's last reader has finished)."""
        self.consumed[slot].synchronize()
        with torch.cuda.stream(self.stream):
            self.stage[slot].copy_from(host_frame)
            self.ready[slot].record(self.stream)

    def take(self, slot: int) -> Frame:
        """The staged frame (its upload has completed when this returns)."""
        self.ready[slot].synchronize()
        return self.stage[slot]

    def release(self, slot: int):
        """Everything enqueued on the current stream so far is the last reader of staging slot ``slot``."""
        self.consumed[slot].record(torch.cuda.current_stream(self.device))

    def run(self, step, host_frames, n: int):
        """``step(frame)`` over n frames (host_frames cyclically), each uploaded one step ahead."""
        k = self.slots
        self.prefetch(host_frames[0], 0)
        for i in range(n):
            if i + 1 < n:
                self.prefetch(host_frames[(i + 1) % len(host_frames)], (i + 1) % k)
            step(self.take(i % k))
            self.release(i % k)


def make_frame(cam, frame_data) -> Frame:
    td = dict(auds=frame_data["auds"], au_exp=frame_data["au_exp"], face_mask=frame_data["face_mask"],
              hair_mask=frame_data["hair_mask"], mouth_mask=frame_data["mouth_mask"],
              lips_rect=frame_data["lips_rect"])
    for k in Frame.OPTIONAL_DICT_TENSORS:
        if frame_data.get(k) is not None:
            td[k] = frame_data[k]
    f = Frame(cam.image_height, cam.image_width, cam.FoVx, cam.FoVy, cam.world_view_transform,
              cam.full_proj_transform, cam.camera_center, td, frame_data["gt_image"])
    return f.packed() if frame_data["gt_image"].is_cuda else f


def with_grad(params: List[torch.Tensor]) -> List[torch.Tensor]:
    """The parameters that received a gradient this step.  Which ones do depends on the step's phase only (e.g. the
    personalised field's deformation head is never evaluated with personalized=False), so every rank gets the same
    list; a parameter without gradient is not exchanged and -- as on one GPU -- not stepped."""
    return [p for p in params if p.grad is not None]


def flat_grad_bucket(params: List[torch.Tensor]) -> torch.Tensor:
    """Concatenate the gradients of ``params`` (all of which have one, see with_grad) into one contiguous fp32 bucket."""
    return torch.cat([p.grad.reshape(-1) for p in params])


def scatter_grad_bucket(params: List[torch.Tensor], bucket: torch.Tensor):
    """Hand the reduced gradients back as VIEWS of the bucket (no copy: ~45 parameter tensors would be ~45 launches per
    step, a third of a captured step's time).  The bucket must stay alive and untouched until the optimizers have read
    it -- it does: the next step's gradients are concatenated into a fresh tensor (eager) / the captured step rewrites
    it only in its first graph, the optimizers run in the second."""
    o = 0
    for p in params:
        n = p.numel()
        p.grad = bucket[o:o + n].view_as(p)
        o += n


def allreduce_gradients(params: List[torch.Tensor], extras: Optional[List[torch.Tensor]] = None, average=True):
    """One fused-bucket all-reduce(SUM) of all gradients (+ extra stat tensors, summed not averaged)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    extras = extras or []
    params = with_grad(params)
    if not params and not extras:
        return                      # nothing to exchange (same on every rank: the set depends on the phase only)
    g = flat_grad_bucket(params) if params else torch.empty(0, dtype=extras[0].dtype, device=extras[0].device)
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


@dataclass(frozen=True)
class FacePhase:
    """What a face-branch iteration computes (train_face.py:340-350, 426-575)."""
    align: bool = True             # personalised field's alignment on (iteration > 1000)
    warm: bool = True              # iteration > warm_step: motion / alpha / attention regularisers
    hair_mask_iter: bool = False   # hair painted to background in image and target, hair attention terms off
    priors: bool = False           # iteration > warm_step + 2000: monocular-normal term
    prior_depth: bool = False      # ... and, outside the 100 iterations after an opacity reset, the depth term


C3_PHASE = FacePhase()             # the phase config C3 / bench.py measures (warm_step < iteration <= warm_step + 2000)


def face_phase(iteration: int, opt=OptimizationParams, warm_step: int = 3000, hair_mask_interval: int = 7,
               mode_long: bool = False) -> FacePhase:
    """Phase of iteration `iteration` under the reference's schedule (train_face.py:39-46, 340-350, 458-478)."""
    lpips_start_iter = opt.densify_until_iter - 1500
    hair = (warm_step < iteration < lpips_start_iter - 1000) and iteration % hair_mask_interval != 0
    align = iteration > 1000 if iteration < warm_step else True
    priors = (not mode_long) and iteration > warm_step + 2000
    return FacePhase(align=align, warm=iteration > warm_step, hair_mask_iter=hair, priors=priors,
                     prior_depth=priors and iteration % opt.opacity_reset_interval > 100)


_DENSITY_WARM = set()


@torch.no_grad()
def warm_density_control(device, sh_degree: int = 1, opt=OptimizationParams):
    """Run every torch operator of a density-control event once, on a 256-Gaussian dummy model.  torch loads a kernel's
    code object on the kernel's first launch; the masked gathers, concatenations, random draws and reductions of
    densify / prune / opacity reset otherwise pay that (~0.8 s in total on this image) inside the first event of a run,
    with the device idle.  Called once per device by FaceTrainer.enable_graph when density control is on."""
    device = torch.device(device)
    key = (device.type, device.index, sh_degree)
    if key in _DENSITY_WARM or device.type != "cuda":
        return
    _DENSITY_WARM.add(key)
    from .gaussian_model import sh_to_rgb
    g = GaussianModel(sh_degree).create_random(256, device, seed=0)
    g.training_setup(opt, fused=True)
    gen = torch.Generator(device=device).manual_seed(0)
    for _ in range(2):
        n = g.num_points
        g.xyz_gradient_accum = torch.rand(n, 1, device=device)
        g.denom = torch.ones(n, 1, device=device)
        g.max_radii2D = torch.rand(n, device=device) * 30.0
        g._p["scaling"].data[: n // 2] -= 3.0                        # both the clone and the split selection non-empty
        g.densify_and_prune(0.5, 0.005, 0.2, 20, generator=gen)
        g.reset_opacity()
        rgb = sh_to_rgb(g.active_sh_degree, g.get_features, g.get_xyz, torch.zeros(3, device=device))
        g.prune_points((rgb[:, 0] < 30 / 255) & (rgb[:, 1] > 225 / 255) & (rgb[:, 2] < 30 / 255))
        g.prune_points(g.get_xyz[:, -1] < -0.07)
    torch.cuda.synchronize(device)


class FaceTrainer:
    """Holds the Gaussians, the UMF (motion_net) and the PMF (gaussians.neural_motion_grid) and steps them.
    ``schedule=None`` runs every step in the C3 phase; ``schedule="reference"`` follows train_face.py's
    iteration-dependent phases (face_phase) and its densification order."""

    def __init__(self, gaussians: GaussianModel, motion_net, background, opt=OptimizationParams,
                 cameras_extent: float = 0.2, densify: bool = True, seed: int = 0, schedule: Optional[str] = None):
        assert schedule in (None, "reference")
        self.schedule = schedule
        self.g = gaussians
        self.motion_net = motion_net
        self.bg = background
        self.opt = opt
        self.extent = cameras_extent
        self.densify = densify
        self.iteration = 0
        dev = gaussians.get_xyz.device
        self.device = dev
        self.gen = torch.Generator(device=dev).manual_seed(seed)     # identical on every rank
        self.on_gpu = dev.type == "cuda"
        # train_face.py:59-60: AdamW(betas .9/.99, eps 1e-8, wd .01), lr x0.1 during warm-up then 0.5^(it/iters)
        self._setup_optimizers()
        self.last = {}
        self._graph = None
        self._graph_phase = None
        self._graph_cache = {}
        self._graph_mode = None       # set by enable_graph: headroom / split / sticky capacity of the captured steps
        self._pool = None             # one private memory pool for every capture of this trainer (re-captures reuse it)
        self.recaptures = 0

    # ---- optimizers: learning rates are device scalars on the GPU so a captured step can be replayed -------
    def _setup_optimizers(self):
        groups = self.motion_net.get_params(5e-3, 5e-4)
        self._motion_base_lr = [float(g["lr"]) for g in groups]
        if self.on_gpu:
            from .optim import MultiTensorAdam
            self.motion_optimizer = MultiTensorAdam(groups, lr=5e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01,
                                                    decoupled=True)
        else:
            self.motion_optimizer = torch.optim.AdamW(groups, lr=5e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
        self.g.training_setup(self.opt, fused=self.on_gpu)
        self._combined = None
        if self.on_gpu:
            from .optim import CombinedAdam, MultiTensorAdam
            if isinstance(self.g.optimizer, MultiTensorAdam):
                # one launch steps both optimizers (train_face.py:781-788 steps them back to back)
                # step("early") / step("late") = the same step as two launches (GraphedStep's single-graph form): the
                # per-Gaussian parameters whose gradients are final when the backward pass reaches the motion fields
                # (_forward_backward_cut) are stepped beside the rest of the pass, positions and networks behind it
                early = lambda q: any(q is v for k, v in self.g._p.items() if k != "xyz")
                self._combined = CombinedAdam([self.motion_optimizer, self.g.optimizer], partition=early)

    def _step_optimizers(self, part=None):
        if self._combined is not None:
            self._combined.step(part)
        else:
            self.motion_optimizer.step()
            self.g.optimizer.step()

    def _motion_lr_factor(self, it):
        warm_step, iters = 3000, self.opt.iterations
        return 0.1 if it < warm_step else 0.5 ** (it / iters)

    def _set_learning_rates(self, it):
        """Per-step schedules (train_face.py:60, scene/gaussian_model.py:421-427) written into the lr slots."""
        f = self._motion_lr_factor(it - 1)      # LambdaLR: step `it` runs with lambda(it - 1)
        for grp, base in zip(self.motion_optimizer.param_groups, self._motion_base_lr):
            grp["lr"] = base * f
        self.g.update_learning_rate(it)
        for opt_ in ((self._combined,) if self._combined is not None else (self.motion_optimizer, self.g.optimizer)):
            if hasattr(opt_, "set_lrs"):
                opt_.set_lrs()          # one small copy into the device-side learning-rate table

    def _all_params(self):
        ps = self.g.per_gaussian_parameters()
        ps += [p for p in self.motion_net.parameters()]
        if self.g.neural_motion_grid is not None:
            ps += [p for p in self.g.neural_motion_grid.parameters()]
        return ps

    # ---- loss block (train_face.py:450-456, 508-540) --------------------------------------------------------
    def phase_of(self, it: int) -> FacePhase:
        return face_phase(it, self.opt) if self.schedule == "reference" else C3_PHASE

    def loss_fn(self, frame: Frame, pkg, warm: bool, hair_mask_iter: bool = False, priors: bool = False,
                prior_depth: bool = False):
        """-> (loss, Ll1).  `warm` = iteration > warm_step: motion regularisers, alpha and attention terms;
        `priors` / `prior_depth`: the monocular normal / depth terms of train_face.py:458-504."""
        dev = self.bg.device
        td = frame.talking_dict
        extra = alpha = attn = lips = None
        w_extra = 1e-5
        if warm:
            m, pm = pkg["motion"], pkg["p_motion"]
            if pkg.get("motion_reg") is not None:
                extra, w_extra = pkg["motion_reg"], 1.0        # already weighted, computed by the deform operator
            else:
                # (the dictionary entries: render_motion has written the combined, scaled displacements back into
                # them, as the reference's in-place arithmetic does)
                extra = (m["d_xyz"].abs().mean() + m["d_rot"].abs().mean() + m["d_opa"].abs().mean()
                         + m["d_scale"].abs().mean() + pm["p_xyz"].abs().mean())
            alpha, attn, lips = pkg["alpha"], pkg["attn"], td["lips_rect"].to(dev)
        loss, Ll1 = face_loss(pkg["render"], frame.original_image.to(dev), td["face_mask"].to(dev),
                              td["hair_mask"].to(dev), td["mouth_mask"].to(dev), self.bg, alpha=alpha, attn=attn,
                              lips_rect=lips, extra=extra, lambda_dssim=self.opt.lambda_dssim, w_extra=w_extra,
                              hair_mask_iter=hair_mask_iter)
        if priors:
            from .losses import geometry_prior_loss
            loss = loss + geometry_prior_loss(pkg["normal"], pkg["depth"], td["normal"].to(dev),
                                              td["depth"].to(dev) if prior_depth else None, td["face_mask"].to(dev),
                                              td["hair_mask"].to(dev), td["mouth_mask"].to(dev), use_depth=prior_depth)
        return loss, Ll1

    # ---- one step ---------------------------------------------------------------------------------------------
    def _forward_backward(self, frame: Frame, phase: FacePhase = C3_PHASE, fold_aux: bool = False):
        from .renderer import render_motion
        pkg = render_motion(frame, self.g, self.motion_net, None, self.bg, return_attn=True, personalized=False,
                            align=phase.align, motion_reg_weight=1e-5 if phase.warm else None)
        from contextlib import nullcontext
        from .losses import defer_finalize
        # (backward follows at once and the loss value is read after the step: the loss block's scalar stage rides in
        # its backward launch -- unless the prior terms are ADDED to the value here, which needs it now)
        with (nullcontext() if phase.priors else defer_finalize()):
            loss, Ll1 = self.loss_fn(frame, pkg, warm=phase.warm, hair_mask_iter=phase.hair_mask_iter,
                                     priors=phase.priors, prior_depth=phase.prior_depth)
        from .deferred import deferred_grads
        from . import diff_gauss
        # fold_aux (only callers that run _stats_and_optimizers(pkg) next): the auxiliary image's share of the screen-space
        # gradient is added by the statistics kernel instead of by a launch of its own at the end of backward
        fold = bool(fold_aux and self.on_gpu)
        diff_gauss.FOLD_AUX_M2D = fold
        try:
            with deferred_grads(self.device if self.on_gpu else None):
                # the MLPs' weight gradients are batched into one launch at the end (deferred.py); the root gradient is
                # a cached constant (no fill launch per step)
                if self.on_gpu:
                    if getattr(self, "_one", None) is None:
                        self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
                    loss.backward(gradient=self._one)
                else:
                    loss.backward()
        except BaseException:
            diff_gauss.reset_aux_state()       # a failed backward must not withhold the NEXT step's aux share
            raise
        finally:
            diff_gauss.FOLD_AUX_M2D = False
        if fold:
            pkg["_m2d_aux"] = diff_gauss.take_folded_aux(pkg["viewspace_points"])
        return pkg, loss, Ll1

    def _forward_backward_cut(self, frame: Frame, phase: FacePhase = C3_PHASE, fold_aux: bool = False):
        """The step's forward and the FIRST part of its backward: from the loss through the loss block, the rasterizer
        and the deform operator -- up to the tensors render_motion names as the cut (the motion fields' head outputs,
        the routed position, the attention colours).  Afterwards the gradients of every per-Gaussian parameter except
        the position (SH coefficients, opacity, scale, rotation: 20 of 24 floats per Gaussian at SH degree 1, 8 MB at
        100k) are FINAL and in ``.grad``; ``finish()`` runs the rest of the backward pass (motion fields, encoders,
        position).  A data-parallel step exchanges the first bucket while ``finish()`` computes (GraphedStep "early").
        -> (pkg, loss, Ll1, early_params, finish)"""
        from . import renderer
        from .renderer import render_motion
        from .deferred import deferred_grads
        from . import diff_gauss
        assert self.on_gpu
        renderer.MARK_BACKWARD_CUT = True
        try:
            pkg = render_motion(frame, self.g, self.motion_net, None, self.bg, return_attn=True, personalized=False,
                                align=phase.align, motion_reg_weight=1e-5 if phase.warm else None)
        finally:
            renderer.MARK_BACKWARD_CUT = False
        from contextlib import nullcontext
        from .losses import defer_finalize
        # (backward follows at once and the loss value is read after the step: the loss block's scalar stage rides in
        # its backward launch -- unless the prior terms are ADDED to the value here, which needs it now)
        with (nullcontext() if phase.priors else defer_finalize()):
            loss, Ll1 = self.loss_fn(frame, pkg, warm=phase.warm, hair_mask_iter=phase.hair_mask_iter,
                                     priors=phase.priors, prior_depth=phase.prior_depth)
        cut = dict.get(pkg, "_cut")
        if not cut:
            raise RuntimeError("the three-segment step needs render_motion's fused path (align=True on the GPU)")
        cut = [c for c in cut if c is not None and c.requires_grad]
        vs = pkg["viewspace_points"]
        early = [q for k, q in self.g._p.items() if k != "xyz" and q.requires_grad]
        if getattr(self, "_one", None) is None:
            self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
        diff_gauss.FOLD_AUX_M2D = "always" if fold_aux else False    # (see _forward_backward)
        try:
            with deferred_grads(self.device):
                got = torch.autograd.grad(loss, cut + early + [vs], grad_outputs=self._one, allow_unused=True,
                                          retain_graph=False)
        except BaseException:
            diff_gauss.reset_aux_state()
            raise
        finally:
            diff_gauss.FOLD_AUX_M2D = False
        if fold_aux:
            pkg["_m2d_aux"] = diff_gauss.take_folded_aux(vs)
        g_cut, g_early, g_vs = got[:len(cut)], got[len(cut):len(cut) + len(early)], got[-1]
        have = []
        for q, gq in zip(early, g_early):
            if gq is not None:
                q.grad = gq
                have.append(q)
        # (the aux image's share of the screen-space gradient was put into vs.grad by the block's exit)
        if g_vs is not None:
            vs.grad = g_vs if vs.grad is None else vs.grad.add_(g_vs)
        roots = [(c, gc) for c, gc in zip(cut, g_cut) if gc is not None]

        def finish():
            with deferred_grads(self.device):
                torch.autograd.backward([c for c, _ in roots], [gc for _, gc in roots])

        return pkg, loss, Ll1, have, finish

    @torch.no_grad()
    def _update_stats(self, vs_grad, radii, grad_add=None):
        """Densification statistics of this rank's frame (train_face.py:670-671; scene/gaussian_model.py:683-685).
        With several ranks they stay LOCAL sums / maxima and are exchanged once, when a densification reads them
        (sync_densification_stats): sum and max commute with the per-step accumulation."""
        g = self.g
        if vs_grad.is_cuda and g.max_radii2D.dtype == torch.float32 and radii.dtype == torch.int32 \
                and vs_grad.is_contiguous():
            from .glue import densify_stats
            densify_stats(vs_grad, radii, g.max_radii2D, g.xyz_gradient_accum, g.denom, grad_add)
            return
        if grad_add is not None:
            vs_grad.add_(grad_add)
        vis = radii > 0
        rmax = torch.where(vis, radii.to(g.max_radii2D.dtype), torch.zeros_like(g.max_radii2D))
        g.max_radii2D.copy_(torch.max(g.max_radii2D, rmax))
        g.add_densification_stats(vs_grad, vis)

    @torch.no_grad()
    def sync_densification_stats(self):
        """Several ranks: every replica gets the statistics of all ranks' frames (sum of the gradient norms and
        visibility counts, maximum of the screen radii) before a densification decides on them."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        g = self.g
        both = torch.cat([g.xyz_gradient_accum.reshape(-1), g.denom.reshape(-1)])
        dist.all_reduce(both, op=dist.ReduceOp.SUM)
        n = g.xyz_gradient_accum.numel()
        g.xyz_gradient_accum.copy_(both[:n].view_as(g.xyz_gradient_accum))
        g.denom.copy_(both[n:].view_as(g.denom))
        dist.all_reduce(g.max_radii2D, op=dist.ReduceOp.MAX)

    @torch.no_grad()
    def _stats_and_optimizers(self, pkg, distributed: bool, it: Optional[int] = None, frame: Optional[Frame] = None):
        """Everything of an iteration behind loss.backward(), in the reference's order (train_face.py:667-788):
        statistics -> [gradient exchange] -> [densify / prune / opacity reset] -> optimizers."""
        self._update_stats(pkg["viewspace_points"].grad, pkg["radii"], dict.get(pkg, "_m2d_aux"))
        if distributed:
            allreduce_gradients(self._all_params())
        if it is not None:
            self._maybe_densify(it, frame)
        self._step_optimizers()

    def _zero_grad(self):
        self.motion_optimizer.zero_grad(set_to_none=True)
        self.g.optimizer.zero_grad(set_to_none=True)

    def _densify_due(self, it):
        o = self.opt
        if not self.densify:
            return False
        densify = it < o.densify_until_iter and it > o.densify_from_iter and it % o.densification_interval == 0
        reset = it < o.densify_until_iter and it % o.opacity_reset_interval == 0
        prune = self.schedule == "reference" and it > o.densify_from_iter and it % o.densification_interval == 0
        return densify or reset or prune

    @torch.no_grad()
    def _maybe_densify(self, it, frame: Optional[Frame] = None):
        """Adaptive density control of the face branch (train_face.py:667-746).  Runs BEFORE the optimizers, as in
        the reference: the rebuilt Gaussian parameters carry no gradient, so only the motion network steps in such an
        iteration.  Any change of the parameter set drops a captured graph."""
        if not self._densify_due(it):
            return False
        import time
        t0 = time.perf_counter()
        o = self.opt
        interval_hit = it > o.densify_from_iter and it % o.densification_interval == 0
        if it < o.densify_until_iter:
            if interval_hit:
                self.sync_densification_stats()
                size_threshold = 20 if it > o.opacity_reset_interval else None
                self.g.densify_and_prune(o.densify_grad_threshold, 0.05 + 0.25 * it / o.densify_until_iter,
                                         self.extent, size_threshold, generator=self.gen)
            if it % o.opacity_reset_interval == 0:
                self.g.reset_opacity()
        if self.schedule == "reference" and interval_hit:
            # train_face.py:729-746: Gaussians that took the background's green, and the ones behind z = -0.07
            from .gaussian_model import sh_to_rgb
            center = frame.camera_center.to(self.device)
            rgb = sh_to_rgb(self.g.active_sh_degree, self.g.get_features, self.g.get_xyz, center)
            green = (rgb[:, 0] < 30 / 255) & (rgb[:, 1] > 225 / 255) & (rgb[:, 2] < 30 / 255)
            # (two prunes in the reference; both tests are per Gaussian, so one rebuild with the union removes the same rows)
            self.g.prune_points(green | (self.g.get_xyz[:, -1] < -0.07))
        self._drop_graph(keep_mode=True)          # (graph mode stays on: the next iteration captures its step again)
        self.density_seconds = getattr(self, "density_seconds", 0.0) + time.perf_counter() - t0     # host time (it syncs)
        return True

    def _drop_graph(self, keep_mode: bool = False):
        """Forget every captured step (the parameter set changed, or the caller wants eager launches).  ``keep_mode``:
        graph mode stays on -- step() captures again when it next needs a step of some phase."""
        if self._graph is not None or getattr(self, "_graph_cache", None):
            from . import diff_gauss
            diff_gauss.set_capacity_plan(None)
        self._graph = None
        self._graph_phase = None
        self._graph_cache = {}
        if not keep_mode:
            self._graph_mode = None

    def _prepare_optimizers(self):
        """Device-side tables of the fused optimizers brought up to date with the parameter set (no step)."""
        for o in ((self._combined,) if self._combined is not None else (self.motion_optimizer, self.g.optimizer)):
            if hasattr(o, "prepare"):
                o.prepare()

    def _recapture(self, frame: Frame, phase: FacePhase, min_capacity: int = 0):
        """Capture the step of ``phase`` again WITHOUT running a single train step: graph mode is on (enable_graph was
        called once, its warm-up steps warmed every library and measured the instance counts) and only the parameter
        set (densify / prune / opacity reset) or the needed capacity changed since.  The capacity follows the Gaussian
        count; the training state and the iteration counter are untouched, so the decision to capture may be taken by
        every rank of a data-parallel run independently of what the others replay."""
        import time
        t0 = time.perf_counter()
        mode = self._graph_mode
        n = max(1, self.g.num_points)
        cap = max(int(mode["capacity"] * max(1.0, n / mode["capacity_n"])), int(min_capacity))
        g = GraphedStep(self, frame, mode["headroom"], 0, mode["split"], phase, min_capacity=cap)
        mode["capacity"], mode["capacity_n"] = g.capacity, n
        self._graph_cache[phase] = g
        self.recaptures = getattr(self, "recaptures", 0) + 1
        self.recapture_seconds = getattr(self, "recapture_seconds", 0.0) + time.perf_counter() - t0      # host time
        return g

    def _overflow_decision(self, graph) -> int:
        """Every CHECK_EVERY replays: did any step since the last look need more instances than the captured capacity?
        One synchronising read of the sticky device flags, made COLLECTIVE with several ranks (all-reduce MAX of flag
        and peak), so that every rank takes the same decision in the same step and captures with the same capacity.
        -> 0, or the peak need."""
        over = graph.plan.overflowed()
        peak = max([v for _, v in over], default=0)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([peak], dtype=torch.int64, device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            peak = int(t.item())
        return peak

    def step(self, frame: Frame):
        self.iteration += 1
        it = self.iteration
        self._set_learning_rates(it)
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        phase = self.phase_of(it)
        mode = getattr(self, "_graph_mode", None)
        g = None
        if self._densify_due(it):
            # the parameter set is about to change: every captured step is stale.  This iteration runs eagerly (density
            # control sits between backward and the optimizers); the next one captures again
            self._drop_graph(keep_mode=True)
        elif mode is not None or self._graph_cache or self._graph is not None:
            # one captured step per phase (FacePhase): the schedule alternates between a few of them (the hair
            # iterations toggle six times out of seven)
            g = self._graph_cache.get(phase)
            if g is None and self._graph is not None and self._graph_phase == phase:
                g = self._graph
            if g is None and mode is not None and mode["auto"]:
                g = self._recapture(frame, phase)
            if g is None and self._graph is not None:
                from . import diff_gauss
                diff_gauss.set_capacity_plan(None)          # a phase nobody captured (auto off): eager launches
            self._graph, self._graph_phase = g, (phase if g is not None else None)
        if g is not None:
            g.replay(frame)
            loss, Ll1 = g.loss, g.l1
            if g.check_due():
                peak = self._overflow_decision(g)
                if peak:
                    # some replayed step needed more instances than the captured capacity (its image was truncated to
                    # the nearest Gaussians, gradients of the dropped ones zero): capture again, sized from the peak
                    # need -- lazily, phase by phase, without consuming iterations (_recapture)
                    if mode is None:
                        mode = self._graph_mode = dict(headroom=1.4, split=g.split, auto=True, capacity=g.capacity,
                                                       capacity_n=max(1, self.g.num_points))
                    mode["capacity"] = max(mode["capacity"], int(1.4 * peak) + 4096)
                    mode["auto"] = True
                    self._drop_graph(keep_mode=True)
        else:
            from . import diff_gauss
            if diff_gauss._CAPACITY_PLAN is not None:
                diff_gauss._CAPACITY_PLAN.begin_step()
            pkg, loss, Ll1 = self._forward_backward(frame, phase, fold_aux=True)
            self._stats_and_optimizers(pkg, distributed, it, frame)
            self._zero_grad()
        self.last = dict(loss=loss.detach(), l1=Ll1.detach(), num_points=self.g.num_points, phase=phase)
        return self.last

    # ---- state snapshot (benchmark windows start from the same state; in place, so captured graphs stay valid) ----
    def _optimizer_states(self):
        """[(optimizer state dict of one parameter)] in a fixed order (motion optimizer first, groups, parameters)."""
        out = []
        for opt_ in (self.motion_optimizer, self.g.optimizer):
            for grp in opt_.param_groups:
                for p in grp["params"]:
                    out.append(opt_.state.get(p) if hasattr(opt_.state, "get") else None)
        return out

    def _state_tensors(self):
        ts = [p.data for p in self._all_params()]
        for st in self._optimizer_states():
            if st:
                ts += [st[k] for k in ("exp_avg", "exp_avg_sq", "step") if torch.is_tensor(st.get(k))]
        ts += [self.g.xyz_gradient_accum, self.g.denom, self.g.max_radii2D]
        return ts

    @torch.no_grad()
    def snapshot(self):
        opt = [None if not st else {k: st[k].detach().clone() for k in ("exp_avg", "exp_avg_sq", "step")
                                     if torch.is_tensor(st.get(k))} for st in self._optimizer_states()]
        return dict(iteration=self.iteration, params=[p.data.detach().clone() for p in self._all_params()], opt=opt,
                    stats=[t.detach().clone() for t in (self.g.xyz_gradient_accum, self.g.denom, self.g.max_radii2D)])

    @torch.no_grad()
    def restore(self, snap):
        """Copy a snapshot() back IN PLACE (same parameter set required: no densification in between), so captured
        graphs stay valid.  Optimizer state that did not exist yet at the snapshot (no step had run) goes back to
        zero moments and a zero step count, which is what a first step starts from."""
        params = self._all_params()
        assert len(params) == len(snap["params"]), "the parameter set changed since the snapshot"
        for p, src in zip(params, snap["params"]):
            assert p.shape == src.shape, "the parameter set changed since the snapshot"
            p.data.copy_(src)
        for st, saved in zip(self._optimizer_states(), snap["opt"]):
            if not st:
                continue
            for k in ("exp_avg", "exp_avg_sq", "step"):
                if torch.is_tensor(st.get(k)):
                    if saved and k in saved:
                        st[k].copy_(saved[k])
                    else:
                        st[k].zero_()
        for dst, src in zip((self.g.xyz_gradient_accum, self.g.denom, self.g.max_radii2D), snap["stats"]):
            dst.copy_(src)
        self.iteration = snap["iteration"]

    # ---- graph mode --------------------------------------------------------------------------------------------
    def enable_graph(self, example_frame: Frame, headroom: float = 1.4, warmup_steps: int = 3,
                     split_for_allreduce: Optional[bool] = None, phase: Optional[FacePhase] = None,
                     min_capacity: int = 0, auto_recapture: bool = True, keep_state: bool = False):
        """Capture the whole step into a hipGraph and switch graph mode on.  Runs `warmup_steps` eager steps plus two
        capacity-mode steps first to measure the instance counts and warm every library: real train steps that advance
        the iteration counter -- unless ``keep_state``, which restores parameters, optimizer state, statistics and the
        counter afterwards.  The graph holds the launches of one phase (FacePhase): `phase`, or the phase of the
        iteration right after the capture.  Captured steps are kept per phase; every one of them is dropped when the
        parameter set changes (densify / prune / opacity reset) or a replay overflowed the instance capacity.  With
        ``auto_recapture`` step() then captures the step it needs again by itself, without warm-up steps and without
        touching the training state (_recapture); without it such a step launches eagerly."""
        if not hasattr(self, "_graph_cache"):
            self._graph_cache = {}
        self._graph = None
        if phase is None:
            after = self.iteration + (0 if keep_state else max(1, warmup_steps) + 2) + 1
            phase = self.phase_of(after)                                       # the iteration right after capture
        snap = self.snapshot() if keep_state else None
        if self.densify and self.on_gpu:
            warm_density_control(self.device, self.g.max_sh_degree, self.opt)
        self._graph = GraphedStep(self, example_frame, headroom, max(1, warmup_steps), split_for_allreduce, phase,
                                  min_capacity=min_capacity, restore=snap)
        self._graph_phase = phase
        self._graph_cache[phase] = self._graph
        prev = getattr(self, "_graph_mode", None)
        self._graph_mode = dict(headroom=headroom, split=self._graph.split, auto=bool(auto_recapture),
                                capacity=max(self._graph.capacity, prev["capacity"] if prev else 0),
                                capacity_n=max(1, self.g.num_points))
        return self._graph


class _no_gc:
    """No cyclic garbage collection inside a stream-capture window.  The crash this once papered over (a segmentation
    fault in capture_end) is addressed at its cause in instag_amd/_keepalive.py: tensors that cross streams are no
    longer marked with record_stream inside a capture, the capture's owner keeps them alive until it has ended.  The
    collector stays off during the window all the same: a collection there frees an earlier step's blocks into the
    capture's private pool at an arbitrary point of the captured sequence, which makes captures irreproducible.
    ``collect=False`` (re-captures inside a train loop): no full collection in front either -- it costs tens of
    milliseconds, as much as ten train steps."""

    def __init__(self, collect: bool = True):
        self.collect = collect

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        if self.collect:
            gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class GraphedStep:
    CHECK_EVERY = 64         # replays between two looks at the (sticky, device-side) overflow flags

    def __init__(self, trainer: FaceTrainer, example: Frame, headroom: float, warmup_steps: int,
                 split_for_allreduce: Optional[bool] = None, phase: FacePhase = C3_PHASE, min_capacity: int = 0,
                 restore=None):
        """``warmup_steps`` > 0: the cold path (eager steps measure the instance count and warm every library; with
        ``restore`` = a trainer.snapshot() the training state is put back afterwards).  0: the warm path of
        FaceTrainer._recapture -- nothing runs, ``min_capacity`` is the capacity."""
        from . import diff_gauss
        self.phase = phase
        t = self.trainer = trainer
        dev = t.device
        assert dev.type == "cuda", "graph mode needs the GPU"
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # two graphs with the (eager) gradient all-reduce between them; can be forced for single-rank tests.
        # "early": THREE graphs -- A' (forward, backward down to the motion fields' outputs), A'' (the motion fields'
        # backward), B (scale, hand back, statistics, optimizers): the bucket of the per-Gaussian gradients that are
        # final after A' (8 of the 10 MB at 100k Gaussians) is exchanged WHILE A'' runs, the small second bucket
        # (positions + networks) behind it.  INSTAG_DP_EARLY_ALLREDUCE=1 makes it the form several ranks use.
        if split_for_allreduce is None:
            split_for_allreduce = ("early" if os.environ.get("INSTAG_DP_EARLY_ALLREDUCE", "1") == "1" else True) \
                if self.distributed else False
        # (the cut runs through the fused deform operator, which needs the alignment on: a phase without it -- the same
        # on every rank -- takes the two-graph form)
        self.early = split_for_allreduce == "early" and bool(phase.align)
        self.split = bool(split_for_allreduce)
        self.early_optimizer = (not self.split and bool(phase.align) and t._combined is not None
                                and os.environ.get("INSTAG_EARLY_OPTIMIZER", "0") == "1")
        self.static = example.clone_static()
        cold = warmup_steps > 0
        if cold:
            # 1. eager warm-up in the normal (host round trip) mode: measures R of both raster passes
            diff_gauss.set_capacity_plan(None)
            needed = 0
            for _ in range(warmup_steps):
                t.iteration += 1
                t._set_learning_rates(t.iteration)
                # (only the package is kept, and only for the statistics: a live loss tensor would keep this step's
                # autograd graph -- and with it every parameter's gradient accumulator, bound to THIS stream -- alive
                # into the capture, whose backward would then hop to this stream for every AccumulateGrad)
                pkg = t._forward_backward(self.static, phase, fold_aux=True)[0]
                t._stats_and_optimizers(pkg, self.distributed)
                t._zero_grad()
                del pkg
                needed = max(needed, diff_gauss.LAST_STATS["num_rendered"])
            cap = max(int(needed * headroom) + 4096, int(min_capacity))
        else:
            assert min_capacity > 0, "a warm capture needs the capacity of an earlier one"
            cap = int(min_capacity)
            t._prepare_optimizers()          # tables of the new parameter set: allocations / uploads outside the capture
            from . import renderer
            renderer.prepare_screenspace(t.g)      # (likewise the zeros behind the screen-space gradient carrier)
        self.plan = diff_gauss.CapacityPlan([cap, cap], dev)
        self._replays = 0
        diff_gauss.set_capacity_plan(self.plan)
        if cold:
            # 2. two eager steps in capacity mode on a side stream (allocator / library warm-up for capture)
            s = _lib.warmup_stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(2):
                    t.iteration += 1
                    t._set_learning_rates(t.iteration)
                    self.plan.begin_step()
                    pkg = t._forward_backward(self.static, phase, fold_aux=True)[0]
                    t._stats_and_optimizers(pkg, self.distributed)
                    t._zero_grad()
                    del pkg
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            if restore is not None:
                t.restore(restore)
                t._set_learning_rates(max(1, t.iteration))
        # 3. capture.  With several ranks the gradient exchange stays outside the graphs:
        #    graph A = forward + backward (+ bucket fill), eager all-reduce, graph B = statistics + optimizers.
        # other threads (the collective library's watchdog) may touch the HIP runtime while this thread captures
        mode = {"capture_error_mode": "thread_local"} if self.distributed else {}
        if t._pool is None:
            t._pool = _lib.GraphPool(dev)
        mode["pool"] = t._pool.handle
        mode["light"] = not cold
        self.graph_a = torch.cuda.CUDAGraph()
        self.graph_b = None
        dot = os.environ.get("INSTAG_GRAPH_DOT")       # diagnostics: the captured step's nodes and edges (DOT)
        if dot:
            self.graph_a.enable_debug_mode()
        self.plan.begin_step()
        self.graph_a2 = None
        # single graph, optimizers in two launches (INSTAG_EARLY_OPTIMIZER=1, off by default): statistics and the
        # per-Gaussian parameters except the positions (20 of 24 floats per Gaussian) on a side stream beside the motion
        # fields' backward, positions + networks behind it.  The last launch of the step shrinks from 24 to 15 us, but
        # whatever the side launches run beside pays for it (sigma_net's backward 64 -> 73 us): no gain measured from
        # any fork point
        if self.early_optimizer:
            side = _lib.side_stream(dev, "early_optimizer")
            with _no_gc(cold), _lib.graph_capture(self.graph_a, **mode):
                main = torch.cuda.current_stream(dev)
                pkg, loss, l1, early, finish = t._forward_backward_cut(self.static, phase, fold_aux=True)

                def early_launches():
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side), torch.no_grad():
                        t._update_stats(pkg["viewspace_points"].grad, pkg["radii"], dict.get(pkg, "_m2d_aux"))
                        t._step_optimizers("early")

                # where the side launches start: beside the largest MLP's backward they cost it 9 us and take 65 us
                # themselves (24 alone); behind it they run beside the heads' and the encoder's backward
                at = os.environ.get("INSTAG_EARLY_OPTIMIZER_AT", "sigma_backward")
                from . import deferred
                if at == "cut":
                    early_launches()
                else:
                    deferred.on_milestone(at, early_launches)
                try:
                    finish()
                    deferred.milestone(at)          # (an operator path without that milestone: launch now)
                finally:
                    deferred.clear_milestones()
                main.wait_stream(side)
                with torch.no_grad():
                    t._step_optimizers("late")
                t._zero_grad()
            del pkg, finish
        elif not self.split:
            with _no_gc(cold), _lib.graph_capture(self.graph_a, **mode):
                pkg, loss, l1 = t._forward_backward(self.static, phase, fold_aux=True)
                t._stats_and_optimizers(pkg, False)
                t._zero_grad()
            # nothing captured is released before the capture has ended (ROCm 7.2: frees inside the capture
            # window intermittently crash hipStreamEndCapture)
            del pkg
        elif self.early:
            with _no_gc(cold), _lib.graph_capture(self.graph_a, **mode):
                pkg, loss, l1, early, finish = t._forward_backward_cut(self.static, phase)
                self._vs_grad, self._radii = pkg["viewspace_points"].grad, pkg["radii"]
                self._params_early = early
                self._bucket_early = flat_grad_bucket(early)
            self.graph_a2 = torch.cuda.CUDAGraph()
            with _no_gc(False), _lib.graph_capture(self.graph_a2, **mode):
                finish()
                ids = {id(q) for q in early}
                self._params = [q for q in with_grad(t._all_params()) if id(q) not in ids]
                self._bucket = flat_grad_bucket(self._params)
            del pkg, finish
            self.graph_b = torch.cuda.CUDAGraph()
            with _no_gc(False), _lib.graph_capture(self.graph_b, **mode):
                with torch.no_grad():
                    if self.distributed:
                        self._bucket_early.mul_(1.0 / dist.get_world_size())
                        self._bucket.mul_(1.0 / dist.get_world_size())
                    scatter_grad_bucket(self._params_early, self._bucket_early)
                    scatter_grad_bucket(self._params, self._bucket)
                    t._update_stats(self._vs_grad, self._radii)
                    t._step_optimizers()
                    t._zero_grad()
        else:
            with _no_gc(cold), _lib.graph_capture(self.graph_a, **mode):
                pkg, loss, l1 = t._forward_backward(self.static, phase)
                self._vs_grad, self._radii = pkg["viewspace_points"].grad, pkg["radii"]
                self._params = with_grad(t._all_params())
                self._bucket = flat_grad_bucket(self._params)
            del pkg
            self.graph_b = torch.cuda.CUDAGraph()
            with _no_gc(False), _lib.graph_capture(self.graph_b, **mode):
                with torch.no_grad():
                    if self.distributed:
                        # mean over ranks of the summed gradients
                        self._bucket.mul_(1.0 / dist.get_world_size())
                    scatter_grad_bucket(self._params, self._bucket)
                    t._update_stats(self._vs_grad, self._radii)        # local; exchanged when a densification reads them
                    t._step_optimizers()
                    t._zero_grad()
        if dot:
            self.graph_a.debug_dump(dot)
        # detached: a retained loss would keep the captured step's autograd graph alive, and with it every parameter's
        # AccumulateGrad node, bound to the capture stream -- the next backward on any other stream (an eager step, the
        # instrumented pass of bench.py) then hops to the capture stream for every parameter
        self.loss, self.l1 = loss.detach(), l1.detach()
        del loss, l1
        self.capacity = cap

    def replay(self, frame: Frame):
        self._replays += 1
        self.static.copy_from(frame)
        self.graph_a.replay()
        if self.graph_a2 is not None:
            # the first bucket travels while the motion fields' backward runs (the collective library works on its own
            # stream, ordered behind graph A' by the stream it was issued from); graph B waits for both exchanges
            work = dist.all_reduce(self._bucket_early, op=dist.ReduceOp.SUM, async_op=True) if self.distributed else None
            self.graph_a2.replay()
            if self.distributed:
                work2 = dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, async_op=True)
                work.wait()
                work2.wait()
            self.graph_b.replay()
            return
        if self.graph_b is not None:
            if self.distributed:
                dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM)       # the division by the world size is in graph B
            self.graph_b.replay()

    def check_overflow(self):
        """Host-side (synchronising) check that no replayed step exceeded the instance capacity (the device flag is
        sticky: every step since the capture / the last clear counts)."""
        return self.plan.overflowed()

    def check_due(self) -> bool:
        """True every CHECK_EVERY replays: the caller then reads the sticky overflow flags (FaceTrainer._overflow_decision:
        one synchronising read per CHECK_EVERY steps -- a function of the replay count alone, hence the same step on every
        rank)."""
        if self._replays < self.CHECK_EVERY:
            return False
        self._replays = 0
        return True


def build_trainer(n_gaussians, device, sh_degree=1, seed=0, densify=False, encoder_cls=None, raw=None, schedule=None):
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
    return FaceTrainer(g, umf, bg, densify=densify, seed=seed, schedule=schedule)
