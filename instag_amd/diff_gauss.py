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

from . import _keepalive, _lib
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
KEEP_LAST_STATE = False       # measurement only (bench.py's traversed-pair count): LAST_STATS["state"] = last forward's state
# instance count of every rasterizer call since the caller last cleared the list (sizing of a CapacityPlan)
RENDERED_LOG = []


class CapacityPlan:
    """Sync-free (hipGraph-capturable) mode of the rasterizer.

    The k-th rasterizer call of a step uses the fixed instance capacity ``capacities[k]`` instead of
    reading the instance count back to the host, and reports into the static device tensor ``status[k]``
    (int32[4]: instances needed by the last call, STICKY overflow flag, largest need since the last clear(),
    instances binned by the last call).  Call ``begin_step()`` before every step (eager or capture)."""

    def __init__(self, capacities, device):
        self.capacities = [int(c) for c in capacities]
        self._all = torch.zeros(len(self.capacities), 4, dtype=torch.int32, device=device)
        self.status = [self._all[k] for k in range(len(self.capacities))]       # views: one copy reads every slot
        # per slot: how far every tile's walk went in the slot's previous call (instag_raster_args.walk_hints) -- the
        # forward blend starts helper workgroups for the tiles that walked far; up to 16,384 tiles (2048 x 2048 pixels)
        self.walk_hints = [torch.zeros(16384, dtype=torch.int32, device=device) for _ in self.capacities]
        self.index = 0
        self._host = None
        self._event = None

    def begin_step(self):
        self.index = 0

    def next_slot(self):
        if self.index >= len(self.capacities):
            raise RuntimeError("CapacityPlan: more rasterizer calls in a step than planned capacities")
        k = self.index
        self.index += 1
        return self.capacities[k], self.status[k], self.walk_hints[k]

    def overflowed(self):
        """Host check (synchronises): list of (slot, largest R needed) of every slot whose capacity was exceeded by ANY
        call since the last clear() -- the flag is sticky on the device, so a check after many replays misses nothing.
        Raises if a binning sort / scan look-back gave up in the meantime (its lists are wrong: sort_stalls())."""
        st = self._all.cpu()
        stalls = sort_stalls()
        if stalls:
            raise RuntimeError(f"rasterizer binning: {stalls} sort / scan look-back(s) gave up waiting for their "
                               "predecessor since the last check; the affected steps rendered wrong tile lists")
        return [(k, int(st[k, 2])) for k in range(len(self.capacities)) if int(st[k, 1]) != 0]

    def poll_overflow(self):
        """Asynchronous form for a replay loop: never waits for the device.  Returns the overflow list read by the
        PREVIOUS call's copy if that copy has landed (else None) and starts a new copy into pinned memory."""
        result = None
        if self._event is not None and self._event.query():
            st = self._host
            if int(self._host_stalls[0]) != 0:
                raise RuntimeError(f"rasterizer binning: {int(self._host_stalls[0])} sort / scan look-back(s) gave up "
                                   "waiting for their predecessor; the affected steps rendered wrong tile lists")
            result = [(k, int(st[k, 2])) for k in range(len(self.capacities)) if int(st[k, 1]) != 0]
            self._event = None
        if self._event is None:
            if self._host is None:
                self._host = torch.zeros(len(self.capacities), 4, dtype=torch.int32).pin_memory()
                self._host_stalls = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._host.copy_(self._all, non_blocking=True)
            check(_lib.lib().instag_raster_sort_stalls(C.c_void_p(self._host_stalls.data_ptr()), 0,
                                                       _lib.current_stream()), "sort_stalls")
            self._event = torch.cuda.Event()
            self._event.record()
        return result

    def needed(self):
        """Instances the most recent call of every slot needed."""
        return [int(v) for v in self._all.cpu()[:, 0]]

    def peak(self):
        """Largest need of every slot since the last clear()."""
        return [int(v) for v in self._all.cpu()[:, 2]]

    def clear(self):
        self._all.zero_()
        self._event = None


def sort_stalls(clear: bool = False) -> int:
    """Sticky count of binning look-backs that gave up (csrc/raster_sort.hip g_sort_stalls); synchronises."""
    L = _lib.lib()
    n = C.c_uint32(0)
    check(L.instag_raster_sort_stalls(C.byref(n), 1, _lib.current_stream()), "sort_stalls")
    if clear and n.value:
        check(L.instag_raster_sort_stalls_clear(_lib.current_stream()), "sort_stalls_clear")
    return int(n.value)


_CAPACITY_PLAN = None


def set_capacity_plan(plan):
    """Install (or clear with None) the capacity plan used by every subsequent rasterizer forward."""
    global _CAPACITY_PLAN
    _CAPACITY_PLAN = plan


class _State:
    """Opaque device buffers kept between forward and backward (geom / binning / image)."""
    __slots__ = ("args", "keep", "geom", "binning", "image", "R", "radii", "N", "H", "W", "E", "M", "aux", "split_sh")


def _make_args(s: GaussianRasterizationSettings, means3D, shs, colors, opac, scales, rots, cov3D, extra):
    N = means3D.shape[0]
    shs_rest = None
    if isinstance(shs, (tuple, list)):           # split SH storage: (dc [N,1,3], rest [N,M-1,3])
        shs, shs_rest = shs
        if shs_rest.shape[1] == 0:
            shs_rest = None
    M = 0 if shs is None else shs.shape[1] + (0 if shs_rest is None else shs_rest.shape[1])
    E = 0 if extra is None else (extra.shape[1] if extra.dim() > 1 else 1)
    bg, view, proj, campos = _f32c(s.bg), _f32c(s.viewmatrix), _f32c(s.projmatrix), _f32c(s.campos)
    _require_cuda(bg=bg, viewmatrix=view, projmatrix=proj, campos=campos)
    a = RasterArgs()
    a.N, a.M, a.sh_degree, a.E = N, M, int(s.sh_degree), E
    a.image_height, a.image_width = int(s.image_height), int(s.image_width)
    a.tanfovx, a.tanfovy, a.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
    a.prefiltered, a.debug = int(bool(s.prefiltered)), int(bool(s.debug))
    a.single_stream = 0 if _lib.may_fork(means3D.device) else 1        # no fork of a fork inside a capture (_lib.py)
    a.bg, a.viewmatrix, a.projmatrix, a.campos = ptr(bg), ptr(view), ptr(proj), ptr(campos)
    a.means3D, a.shs, a.colors_precomp, a.opacities = ptr(means3D), ptr(shs), ptr(colors), ptr(opac)
    a.scales, a.rotations, a.cov3Ds_precomp, a.extra_attrs = ptr(scales), ptr(rots), ptr(cov3D), ptr(extra)
    a.shs_rest = ptr(shs_rest)
    keep = (bg, view, proj, campos, means3D, shs, shs_rest, colors, opac, scales, rots, cov3D, extra)
    return a, keep, N, M, E


def rasterize_forward(settings, means3D, shs, colors, opac, scales, rots, cov3D, extra, aux_colors=None):
    """Run the forward (two-stage with one host round trip, or sync-free under a CapacityPlan).
    Returns (outputs, state); with aux_colors [N,3] the outputs carry a 7th image [3,H,W]."""
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
    aux_img = torch.empty(3, H, W, dtype=torch.float32, device=dev) if aux_colors is not None else None
    if _CAPACITY_PLAN is not None:
        R, status, hints = _CAPACITY_PLAN.next_slot()
        if ((H + 15) // 16) * ((W + 15) // 16) <= hints.numel():
            a.walk_hints = ptr(hints)
        binning = torch.empty(L.instag_raster_binning_bytes(R, H, W), dtype=torch.uint8, device=dev)
        check(L.instag_raster_forward_capacity(C.byref(a), ptr(geom), geom.numel(), ptr(binning), binning.numel(),
                                               ptr(image), image.numel(), R, ptr(radii), ptr(status), ptr(color),
                                               ptr(depth), ptr(normal), ptr(alpha),
                                               ptr(extra_img) if E > 0 else None, ptr(aux_colors), ptr(aux_img),
                                               stream), "rasterize_gaussians")
    else:
        Rc = C.c_int64(0)
        check(L.instag_raster_forward_stage1(C.byref(a), ptr(geom), geom.numel(), ptr(radii), C.byref(Rc), stream),
              "rasterize_gaussians")
        R = int(Rc.value)
        LAST_STATS["num_rendered"], LAST_STATS["num_gaussians"] = R, N
        if len(RENDERED_LOG) < 4096:
            RENDERED_LOG.append(R)
        binning = torch.empty(L.instag_raster_binning_bytes(R, H, W), dtype=torch.uint8, device=dev)
        check(L.instag_raster_forward_stage2(C.byref(a), ptr(geom), geom.numel(), ptr(binning), binning.numel(),
                                             ptr(image), image.numel(), R, ptr(color), ptr(depth), ptr(normal),
                                             ptr(alpha), ptr(extra_img) if E > 0 else None, ptr(aux_colors),
                                             ptr(aux_img), stream),
              "rasterize_gaussians")
    st = _State()
    st.args, st.keep, st.geom, st.binning, st.image = a, keep, geom, binning, image
    st.R, st.radii, st.N, st.H, st.W, st.E, st.M = R, radii, N, H, W, E, M
    st.aux = aux_colors
    st.split_sh = isinstance(shs, (tuple, list)) and shs[1].shape[1] > 0
    if KEEP_LAST_STATE:
        LAST_STATS["state"] = st
    if aux_colors is not None:
        return (color, depth, normal, alpha, radii, extra_img, aux_img), st
    return (color, depth, normal, alpha, radii, extra_img), st


_AUX_STREAMS = {}
_AUX_EVENTS = {}
DEFER_AUX_JOIN = False        # see _RasterizeGaussians.backward; only a caller that joins explicitly may set this
# Joining the mean-only aux pass only at the end of backward (nothing consumes it earlier) was measured on the C3 step:
# 1.235 -> 1.385 ms.  A branch left open through the whole backward takes one of the graph's four hardware queues away
# from the weight-gradient / personalised-field branches that fork later; it is joined where the aux pass always was.
_LATE_JOIN = False
FUSE_AUX_BACKWARD = None      # None: by image size (see _RasterizeGaussians.backward); True / False: force (tests)
_PENDING_AUX = []


def join_pending_aux(final: bool = False):
    """Make the current stream wait for every outstanding aux-image backward.  ``final=True`` (after the backward pass
    has finished) also adds the aux image's means2D contribution to ``means2D.grad`` and forgets the entries."""
    for entry in _PENDING_AUX:
        if entry.get("late") and not final:
            continue            # a mean-only pass: nothing consumes its result before the end of backward
        if not entry["joined"]:
            torch.cuda.current_stream(entry["dev"]).wait_stream(entry["side"])
            entry["joined"] = True
    if final:
        while _PENDING_AUX:
            entry = _PENDING_AUX.pop()
            leaf, m2d_aux = entry["leaf"], entry["m2d_aux"]
            if FOLD_AUX_M2D and (leaf.grad is not None or FOLD_AUX_M2D == "always"):
                _FOLDED.append((leaf, m2d_aux))          # the caller adds it (take_folded_aux)
            else:
                leaf.grad = m2d_aux if leaf.grad is None else leaf.grad.add_(m2d_aux)


# A trainer whose next launch after backward reads means2D.grad anyway (the densification statistics) may take the aux
# image's share from here and add it in that launch (glue.densify_stats(grad_add=...)) instead of paying an elementwise
# launch on the tail of the step: set FOLD_AUX_M2D around backward, call take_folded_aux(leaf) right after it.
# ("always": also when the leaf has no .grad at that point -- its gradient was asked for with torch.autograd.grad.)
FOLD_AUX_M2D = False
_FOLDED = []


def reset_aux_state():
    """Forget every pending / folded aux share (the backward pass that produced them raised)."""
    global FOLD_AUX_M2D
    FOLD_AUX_M2D = False
    del _PENDING_AUX[:]
    del _FOLDED[:]


def take_folded_aux(leaf):
    """The aux image's means2D share that join_pending_aux(final=True) left un-added for ``leaf`` (or None).  Shares of
    other leaves are added to their gradients now."""
    mine = None
    while _FOLDED:
        lf, m2d_aux = _FOLDED.pop()
        if lf is leaf and mine is None:
            mine = m2d_aux
        else:
            lf.grad = m2d_aux if lf.grad is None else lf.grad.add_(m2d_aux)
    return mine


def rasterize_aux_backward(st: "_State", g_aux, want_colors=True, want_means2D=True, stream=None):
    """Backward of the auxiliary colour image -> (dL_daux_colors [N,3], dL_dmeans2D contribution [N,3])."""
    L = _lib.lib()
    dev = st.geom.device
    N = st.N
    d_aux = torch.empty(N, 3, dtype=torch.float32, device=dev) if want_colors else None
    d_m2d = torch.empty(N, 3, dtype=torch.float32, device=dev) if want_means2D else None
    ws = torch.empty(L.instag_raster_backward_workspace_bytes(N, st.R), dtype=torch.uint8, device=dev)
    g_aux = _f32c(g_aux)
    check(L.instag_raster_aux_backward(C.byref(st.args), ptr(st.geom), st.geom.numel(), ptr(st.binning),
                                       st.binning.numel(), ptr(st.image), st.image.numel(), st.R, ptr(st.radii),
                                       ptr(st.aux), ptr(g_aux), ptr(ws), ws.numel(), ptr(d_aux), ptr(d_m2d),
                                       _lib.current_stream() if stream is None else stream),
          "rasterize_gaussians_aux_backward")
    return d_aux, d_m2d


def rasterize_backward(st: _State, g_color, g_depth, g_normal, g_alpha, g_extra, want, g_aux=None, want_aux=False,
                       aux_colors_only=False):
    """want: dict name -> bool for means3D, means2D, shs, colors, opacities, scales, rotations, cov3D, extra.
    g_aux [3,H,W]: upstream gradient of the auxiliary image -- differentiated in the same call (out["aux"] = gradient of
    the aux colours if want_aux; the aux image's share of the means2D gradient is included in out["means2D"])."""
    L = _lib.lib()
    dev = st.geom.device
    N, M = st.N, st.M
    stream = _lib.current_stream()

    def buf(flag, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=dev) if flag else None

    use_sh = st.args.shs is not None
    use_cov = st.args.cov3Ds_precomp is not None
    split = bool(st.split_sh)
    out = dict(
        means3D=buf(want["means3D"], N, 3), means2D=buf(want["means2D"], N, 3),
        shs=buf(want["shs"] and use_sh, N, 1 if split else M, 3),
        shs_rest=buf(want["shs"] and use_sh and split, N, M - 1, 3),
        colors=buf(want["colors"] and not use_sh, N, 3),
        opacities=buf(want["opacities"], N, 1), scales=buf(want["scales"] and not use_cov, N, 3),
        rotations=buf(want["rotations"] and not use_cov, N, 4), cov3D=buf(want["cov3D"] and use_cov, N, 6),
        extra=buf(want["extra"] and st.E > 0, N, st.E),
        aux=buf(want_aux and g_aux is not None and st.aux is not None, N, 3),
    )
    use_aux = g_aux is not None and st.aux is not None
    g_aux = _f32c(g_aux) if use_aux else None
    ws = torch.empty(L.instag_raster_backward_workspace_bytes(N, st.R), dtype=torch.uint8, device=dev)
    gs = [None if g is None else _f32c(g) for g in (g_color, g_depth, g_normal, g_alpha, g_extra)]
    check(L.instag_raster_backward(C.byref(st.args), ptr(st.geom), st.geom.numel(), ptr(st.binning),
                                   st.binning.numel(), ptr(st.image), st.image.numel(), st.R, ptr(st.radii),
                                   ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]),
                                   ptr(gs[4]) if st.E > 0 else None, ptr(ws), ws.numel(),
                                   ptr(out["means3D"]), ptr(out["means2D"]), ptr(out["shs"]), ptr(out["colors"]),
                                   ptr(out["opacities"]), ptr(out["scales"]), ptr(out["rotations"]),
                                   ptr(out["cov3D"]), ptr(out["extra"]), ptr(out["shs_rest"]),
                                   ptr(st.aux) if use_aux else None, ptr(g_aux), ptr(out["aux"]),
                                   1 if (use_aux and aux_colors_only) else 0, stream),
          "rasterize_gaussians_backward")
    return out


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                extra_attrs, raster_settings, aux_colors=None, sh_rest=None):
        _require_cuda(means3D=means3D)
        ctx.set_materialize_grads(False)      # unused outputs (depth / normal / extra) stay NULL in backward
        m3, shs, col = _f32c(means3D), _f32c(sh), _f32c(colors_precomp)
        if sh_rest is not None:
            _require_cuda(shs_rest=sh_rest)
            shs = (shs, _f32c(sh_rest))         # split SH storage: no concatenation, gradients come back split
        op, sc, ro = _f32c(opacities), _f32c(scales), _f32c(rotations)
        cov, ex, aux = _f32c(cov3Ds_precomp), _f32c(extra_attrs), _f32c(aux_colors)
        _require_cuda(shs=shs[0] if isinstance(shs, tuple) else shs, colors_precomp=col, opacities=op, scales=sc,
                      rotations=ro, cov3Ds_precomp=cov, extra_attrs=ex, aux_colors=aux)
        if aux is not None and tuple(aux.shape) != (m3.shape[0], 3):
            raise RuntimeError("aux_colors must be [N,3]")
        outs, st = rasterize_forward(raster_settings, m3, shs, col, op, sc, ro, cov, ex, aux)
        ctx.state = st
        ctx.shapes = (opacities.shape, None if extra_attrs is None else extra_attrs.shape)
        ctx.means2D_leaf = means2D if (means2D is not None and means2D.is_leaf and means2D.requires_grad) else None
        ctx.mark_non_differentiable(outs[4])
        return outs

    @staticmethod
    def backward(ctx, g_color, g_depth, g_normal, g_alpha, _g_radii, g_extra, g_aux=None):
        st = ctx.state
        need = ctx.needs_input_grad
        want = dict(means3D=need[0], means2D=need[1], shs=need[2], colors=need[3], opacities=need[4],
                    scales=need[5], rotations=need[6], cov3D=need[7], extra=need[8])
        d_aux = m2d_aux = None
        aux_needed = st.aux is not None and g_aux is not None and (need[10] or need[1])
        main_grads = any(g is not None for g in (g_color, g_depth, g_normal, g_alpha, g_extra))
        dev = st.geom.device
        tiles = ((st.W + 15) // 16) * ((st.H + 15) // 16)
        full = any(g is not None for g in (g_depth, g_normal)) or (g_extra is not None and st.E > 0)
        # Two ways to differentiate the aux image (the attention map) next to the main images:
        #  * fused -- one blend launch carries both images through one alpha / T recurrence (rgb-only main pass; an
        #    all-channel main pass runs the two launches back to back inside the same C call).  Least total work: right
        #    when the image has far more tiles than the chip has CUs;
        #  * side by side -- a second blend launch on a second stream.  At 512x512 only ~400 tiles are populated, one
        #    workgroup each, so the kernels are bound by the serial walk of a tile's list, not by throughput: two
        #    launches that each redo the recurrence finish sooner than one that does 1.3x the work per list entry
        #    (C3: 297 us fused against ~240 us for the pair).
        fused = aux_needed and main_grads and (FUSE_AUX_BACKWARD if FUSE_AUX_BACKWARD is not None
                                                 else (full or tiles >= 4096))
        if aux_needed and not fused:
            ready = _AUX_EVENTS.get((dev.type, dev.index))       # reused: no event is created / destroyed per step
            if ready is None:
                ready = _AUX_EVENTS[(dev.type, dev.index)] = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
        # side by side with an rgb-only main pass: the aux colours' gradient rides in idle columns of the main launch's
        # matrix product, the side launch only computes the screen-space mean's share (half the matrix work)
        split = aux_needed and main_grads and not fused and not full and need[10]
        if main_grads:
            g = rasterize_backward(st, g_color, g_depth, g_normal, g_alpha, g_extra, want,
                                   g_aux=g_aux if (fused or split) else None, want_aux=need[10], aux_colors_only=split)
            if fused or split:
                d_aux = g.get("aux")
        else:
            g = dict(means3D=None, means2D=None, shs=None, shs_rest=None, colors=None, opacities=None, scales=None,
                     rotations=None, cov3D=None, extra=None)
        if aux_needed and not fused:
            # the aux image only shares the forward state with the main images: its backward runs beside theirs,
            # enqueued AFTER the main one so that in a captured step the main chain keeps its graph branch
            main = torch.cuda.current_stream(dev)
            key = (dev.type, dev.index)
            side = _AUX_STREAMS.get(key)
            if side is None:
                side = _AUX_STREAMS[key] = _lib.side_stream(dev, "aux_backward")
            # (starting the mean-only pass BEHIND the main blend launch instead of beside it was measured: the main launch
            # got 35 us shorter, the step 35 us longer -- the pass then competes with the memory-bound kernels downstream)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                if split:
                    m2d_aux = rasterize_aux_backward(st, g_aux, False, need[1])[1] if need[1] else None
                else:
                    d_aux, m2d_aux = rasterize_aux_backward(st, g_aux, need[10], need[1])
            for t in (g_aux, d_aux, m2d_aux):
                if t is not None:
                    _keepalive.cross_stream(t, side)
            if DEFER_AUX_JOIN and (m2d_aux is None or ctx.means2D_leaf is not None):
                # The caller promised to call join_pending_aux() before it touches the aux gradients (the trainer
                # does, in the glue operator that consumes them) and join_pending_aux(final=True) after backward:
                # the main chain does not wait for the aux image's backward here, and the aux image's means2D
                # contribution is added to means2D.grad at the final join.
                if m2d_aux is not None:
                    _PENDING_AUX.append(dict(dev=dev, side=side, leaf=ctx.means2D_leaf, m2d_aux=m2d_aux, joined=False,
                                             late=bool(split) and _LATE_JOIN))
                else:
                    main.wait_stream(side)
            else:
                main.wait_stream(side)
                if m2d_aux is not None:
                    g["means2D"] = m2d_aux if g["means2D"] is None else g["means2D"].add_(m2d_aux)
        op_shape, ex_shape = ctx.shapes
        g_op = None if g["opacities"] is None else g["opacities"].reshape(op_shape)
        g_ex = None if g["extra"] is None else g["extra"].reshape(ex_shape)
        return (g["means3D"], g["means2D"], g["shs"], g["colors"], g_op, g["scales"], g["rotations"],
                g["cov3D"], g_ex, None, d_aux, g.get("shs_rest"))


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        extra_attrs, raster_settings, aux_colors=None):
    sh_rest = None
    if isinstance(sh, (tuple, list)):
        sh, sh_rest = sh
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, extra_attrs, raster_settings, aux_colors, sh_rest)


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
                cov3Ds_precomp=None, extra_attrs=None, aux_colors=None):
        """Same call and 6-tuple as the reference's rasterizer.  Extension: ``aux_colors`` [N,3] appends a 7th
        output, the image a second call with ``colors_precomp=aux_colors`` on the detached geometry would
        return (gradients reach aux_colors and means2D only), at the cost of three more blend channels.
        ``shs`` may also be the pair (features_dc [N,1,3], features_rest [N,M-1,3]) the Gaussian model stores
        (scene/gaussian_model.py:183-186 concatenates them on every call): same result without the copy."""
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or \
           ((scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if extra_attrs is not None and extra_attrs.dim() > 1 and extra_attrs.shape[1] > 1:
            raise RuntimeError("extra_attrs: only [N] / [N,1] is supported by the MI355X rasterizer")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3Ds_precomp, extra_attrs, self.raster_settings, aux_colors)


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
    flags = torch.zeros(N, dtype=torch.int32, device=dev)
    check(L.instag_raster_debug_export_flags(ptr(st.geom), st.geom.numel(), N, ptr(flags), _lib.current_stream()),
          "debug_export_flags")
    rect = out["rec2d"][:, 15].contiguous().view(torch.int32)
    # bounding tile rectangle (min x, min y, width, height) of the published binning; defined where radii > 0
    out["rect"] = torch.stack([rect & 1023, (rect >> 10) & 1023, (rect >> 20) & 4095, (flags >> 16) & 0xFFFF], dim=1)
    out["R"] = R
    out["radii"] = st.radii
    return out
