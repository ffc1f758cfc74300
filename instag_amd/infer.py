"""Forward-only streaming render of the fused head (face + mouth), the inference path of the reference
(synthesize_fuse.py:34-92: per view ``render_motion`` + ``render_motion_mouth_con(inference=True)`` + compositing).

``FuseRenderer.render(frame)`` runs the same operators as training under ``torch.no_grad``; ``enable_graph`` captures
the whole frame (both rasterizer passes in sync-free capacity mode, packed frame inputs) into one hipGraph so that a
frame costs one small copy and one graph launch.

Streaming ("batch frames per launch"): ``enable_graph(frames_per_replay=K)`` captures K frames into ONE
graph, each on a stream of its own.  A frame's kernels are bound by serial chains (one workgroup per populated tile
walking its list, ~400 of them on 256 CUs), so K independent frames overlap almost for free: ``render_batch`` feeds K
frames with K small copies and one launch and returns the K images.
"""
from __future__ import annotations

import torch

from . import _lib, diff_gauss
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
        return self.render_batch([frame], None if scene_background is None else [scene_background])[0]

    def render_batch(self, frames, scene_backgrounds=None):
        """-> images [len(frames),3,H,W].  With a captured graph of K frames per replay the frames go through in
        groups of K (one launch per group; a short last group is padded with its last frame); the returned tensor
        is a copy only when more than one group was needed."""
        if self._graph is None:
            return torch.stack([self._render(f, None if scene_backgrounds is None else scene_backgrounds[i])
                                for i, f in enumerate(frames)])
        K = len(self._static)
        outs = []
        for g0 in range(0, len(frames), K):
            group = frames[g0:g0 + K]
            for k in range(K):
                j = min(g0 + k, len(frames) - 1)
                self._static[k].copy_from(frames[j])
                if scene_backgrounds is not None:
                    self._static_bg[k].copy_(scene_backgrounds[j], non_blocking=True)
            self._plan.begin_step()
            self._graph.replay()
            if len(frames) <= K:
                return self._out[:len(group)]
            outs.append(self._out[:len(group)].clone())
        return torch.cat(outs)

    def enable_graph(self, example: Frame, headroom: float = 1.5, frames_per_replay: int = 1):
        dev = self.bg.device
        K = max(1, int(frames_per_replay))
        self._static = [example.clone_static() for _ in range(K)]
        self._static_bg = [torch.zeros(3, example.image_height, example.image_width, device=dev) for _ in range(K)]
        diff_gauss.set_capacity_plan(None)
        needed = []
        for _ in range(2):                              # eager warm-up measures the instance counts of both passes
            self._render(self._static[0], self._static_bg[0])
            needed.append(diff_gauss.LAST_STATS["num_rendered"])
        # LAST_STATS holds the last (mouth) pass; size both slots by the larger scene to stay safe
        n_face, n_mouth = self.g.num_points, self.gm.num_points
        cap = int(max(needed) * headroom * max(1.0, n_face / max(1, n_mouth))) + 4096
        self._plan = diff_gauss.CapacityPlan([cap, cap] * K, dev)
        diff_gauss.set_capacity_plan(self._plan)
        lanes = [_lib.side_stream(dev, ("infer_lane", k)) for k in range(K)]

        def all_frames():
            """frame k on lane k, forked from / joined into the current stream.  (On a forked lane the operators keep
            their own work on that one stream -- _lib.may_fork: a fork of a fork inside a capture crashes
            hipStreamEndCapture on ROCm 7.2; the lanes provide the concurrency instead.)"""
            if K == 1:
                return [self._render(self._static[0], self._static_bg[0])]
            main = torch.cuda.current_stream(dev)
            outs = []
            for k in range(K):
                lanes[k].wait_stream(main)
                with torch.cuda.stream(lanes[k]):
                    outs.append(self._render(self._static[k], self._static_bg[k]))
            for k in range(K):
                main.wait_stream(lanes[k])
            return outs

        s = _lib.warmup_stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._plan.begin_step()
                all_frames()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        self._plan.begin_step()
        with _no_gc(), _lib.graph_capture(self._graph):
            outs = all_frames()
            self._out = torch.stack(outs)
        self._lanes = lanes
        return self

    def check_overflow(self):
        return self._plan.overflowed() if self._graph is not None else []

    def close(self):
        self._graph = None
        diff_gauss.set_capacity_plan(None)
