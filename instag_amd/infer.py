"""Forward-only streaming render of the fused head (face + mouth), the inference path of the reference
(synthesize_fuse.py:34-92: per view ``render_motion`` + ``render_motion_mouth_con(inference=True)`` + compositing).

``FuseRenderer.render(frame)`` runs the same operators as training under ``torch.no_grad``; ``enable_graph`` captures
the whole frame (both rasterizer passes in sync-free capacity mode, packed frame inputs) into one hipGraph so that a
frame costs one small copy and one graph launch.
"""
from __future__ import annotations

import torch

from . import diff_gauss
from .renderer import render_fuse
from .train import Frame, _no_gc


class FuseRenderer:
    def __init__(self, gaussians, motion_net, gaussians_mouth, motion_net_mouth, background, personalized=False):
        self.g, self.net, self.gm, self.netm = gaussians, motion_net, gaussians_mouth, motion_net_mouth
        self.bg = background
        self.personalized = personalized
        self._graph = None

    @torch.no_grad()
    def _render(self, frame: Frame, scene_background=None):
        out = render_fuse(frame, self.g, self.net, self.gm, self.netm, None, self.bg,
                          scene_background=scene_background, personalized=self.personalized, inference=True)
        return out["image"].clamp(0, 1)

    def render(self, frame: Frame, scene_background=None):
        """-> image [3,H,W] in [0,1].  With a captured graph the returned tensor is the graph's static output buffer
        (valid until the next call)."""
        if self._graph is None:
            return self._render(frame, scene_background)
        self._static.copy_from(frame)
        if scene_background is not None:
            self._static_bg.copy_(scene_background, non_blocking=True)
        self._plan.begin_step()
        self._graph.replay()
        return self._out

    def enable_graph(self, example: Frame, headroom: float = 1.5):
        dev = self.bg.device
        self._static = example.clone_static()
        self._static_bg = torch.zeros(3, example.image_height, example.image_width, device=dev)
        diff_gauss.set_capacity_plan(None)
        needed = []
        for _ in range(2):                              # eager warm-up measures the instance counts of both passes
            self._render(self._static, self._static_bg)
            needed.append(diff_gauss.LAST_STATS["num_rendered"])
        # LAST_STATS holds the last (mouth) pass; size both slots by the larger scene to stay safe
        n_face, n_mouth = self.g.num_points, self.gm.num_points
        cap = int(max(needed) * headroom * max(1.0, n_face / max(1, n_mouth))) + 4096
        self._plan = diff_gauss.CapacityPlan([cap, cap], dev)
        diff_gauss.set_capacity_plan(self._plan)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._plan.begin_step()
                self._render(self._static, self._static_bg)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        self._plan.begin_step()
        with _no_gc(), torch.cuda.graph(self._graph):
            self._out = self._render(self._static, self._static_bg)
        return self

    def check_overflow(self):
        return self._plan.overflowed() if self._graph is not None else []

    def close(self):
        self._graph = None
        diff_gauss.set_capacity_plan(None)
