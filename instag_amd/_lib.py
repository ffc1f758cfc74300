"""ctypes binding of libinstag_hip.so (the C ABI declared in include/instag_hip.h).

There is no CPU fallback: if the shared library cannot be loaded (and cannot be built with
hipcc), importing any operator raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libinstag_hip.so")

_lib = None
_lock = threading.Lock()

vp = C.c_void_p
u32 = C.c_uint32
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
sz = C.c_size_t


class FaceLossCfg(C.Structure):
    """struct instag_face_loss_cfg (include/instag_hip.h)."""
    _fields_ = [("H", i32), ("W", i32), ("flags", i32), ("w_dssim", f32), ("w_alpha", f32),
                ("w_attn_hair", f32), ("w_attn_lips", f32), ("w_extra", f32)]


class WgradJob(C.Structure):
    """struct instag_wgrad_job (include/instag_hip.h)."""
    _fields_ = [("dz", vp), ("inp", vp), ("dw", vp), ("N", i32), ("O", i32), ("K", i32)]


class RasterArgs(C.Structure):
    """struct instag_raster_args (include/instag_hip.h)."""
    _fields_ = [
        ("N", i32), ("M", i32), ("sh_degree", i32), ("E", i32),
        ("image_height", i32), ("image_width", i32),
        ("tanfovx", f32), ("tanfovy", f32), ("scale_modifier", f32),
        ("prefiltered", i32), ("debug", i32), ("single_stream", i32),
        ("bg", vp), ("viewmatrix", vp), ("projmatrix", vp), ("campos", vp),
        ("means3D", vp), ("shs", vp), ("colors_precomp", vp), ("opacities", vp),
        ("scales", vp), ("rotations", vp), ("cov3Ds_precomp", vp), ("extra_attrs", vp),
        ("shs_rest", vp), ("walk_hints", vp),
    ]


# ---- streams -------------------------------------------------------------------------------------------------------
# torch.cuda.Stream() does not create a HIP stream: it hands out the next of 32 pooled streams per (device, priority),
# round robin.  A process that asks for more than 32 -- every capture used to take a fresh "capture stream" and a fresh
# warm-up stream, every module kept its own cache of side streams -- therefore gets the SAME HIP stream under two
# Python objects: a forked pass and the lane it forks to, a "leaf" stream and an operator's side lane, two branches of
# one captured step.  Every stream of the package now comes from this registry: one per (device, purpose), created once
# (outside any capture) and guaranteed distinct from every other stream the registry handed out.
_STREAMS = {}


def side_stream(device, tag, priority=0):
    """The persistent HIP stream of purpose ``tag`` on ``device`` (distinct handles for distinct tags)."""
    import torch
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, tag)
    s = _STREAMS.get(key)
    if s is None:
        taken = {v.cuda_stream for (d, _), v in _STREAMS.items() if d == index}
        for _ in range(64):
            s = torch.cuda.Stream(device=torch.device("cuda", index), priority=priority)
            if s.cuda_stream not in taken:
                break
        else:
            raise RuntimeError(f"no distinct HIP stream left for {tag!r} ({len(taken)} in use on device {index})")
        _STREAMS[key] = s
    return s


def capture_stream(device=None):
    """THE stream every step of the package is captured on (one per device, high priority: the side streams operators
    fork from it keep the default priority, so the nodes of the step's critical chain win the arbitration where they
    share the chip -- C3 step 0.5-2.5 % faster).  One stream for all captures, because the autograd engine runs a
    parameter's AccumulateGrad on the stream that was current when the node was created: with a stream per capture a
    node kept alive across captures made the engine hop to the EARLIER capture's stream inside the open capture (an
    un-announced fork; PyTorch's "AccumulateGrad node's stream does not match" warning)."""
    import torch
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return side_stream(device, "capture", priority=-1)


def warmup_stream(device):
    """Side stream for the eager capacity-mode warm-up steps in front of a capture."""
    return side_stream(device, "warmup")


# ---- stream capture bookkeeping -------------------------------------------------------------------------------------
# A fork of a fork inside ONE stream capture (origin stream -> stream A -> stream B, B joined back into A, A into the
# origin) crashes hipStreamEndCapture on ROCm 7.2 (segmentation fault inside capture_end; bisected with
# scripts/probes/infer_capture_probe2.py: any operator that forks internally -- the rasterizer's depth-sort stream, a
# motion network's per-frame branch -- breaks a capture when it is itself called on a forked stream, the same
# operators on the origin stream, or without their internal fork, capture fine).  So while a capture is open,
# operators fork only from the capture's ORIGIN stream, which the owners of our captures announce here.
_CAPTURE_ORIGIN = None


class graph_capture:
    """Stream capture of one graph on the package's capture stream; records the capture's origin stream for may_fork()
    and releases the cross-stream tensors held for the capture (_keepalive) when it has ended -- whether it succeeded
    or raised.  The default form is ``torch.cuda.graph`` (device synchronisation + allocator cache flush in front: right
    for a first capture).  ``light=True`` is for re-captures inside a train loop: ``torch.cuda.graph.__enter__`` costs a
    device synchronisation and an ``empty_cache()`` (3 ms of hipFree, and every eager allocation after it pays hipMalloc
    again); here the capture simply begins -- the private pool is kept open by a GraphPool, nothing of the current
    stream's pending work is touched by recording launches."""

    def __init__(self, graph, light: bool = False, **kw):
        import torch
        if "stream" not in kw:
            kw = dict(kw, stream=capture_stream())
        self._light = bool(light)
        self._graph = graph
        if self._light:
            self._stream_ctx = torch.cuda.stream(kw["stream"])
            self._begin = {k: v for k, v in kw.items() if k in ("pool", "capture_error_mode")}
        else:
            self._ctx = torch.cuda.graph(graph, **kw)

    def __enter__(self):
        global _CAPTURE_ORIGIN
        import torch
        if self._light:
            self._stream_ctx.__enter__()
            try:
                self._graph.capture_begin(**self._begin)
            except BaseException:
                self._stream_ctx.__exit__(None, None, None)
                raise
            r = None
        else:
            r = self._ctx.__enter__()
        self._prev = _CAPTURE_ORIGIN
        _CAPTURE_ORIGIN = torch.cuda.current_stream()
        return r

    def __exit__(self, *exc):
        global _CAPTURE_ORIGIN
        _CAPTURE_ORIGIN = self._prev
        try:
            if self._light:
                try:
                    self._graph.capture_end()
                finally:
                    self._stream_ctx.__exit__(*exc)
                return None
            return self._ctx.__exit__(*exc)
        finally:
            from . import _keepalive
            _keepalive.release()


class GraphPool:
    """A private memory pool that outlives the graphs captured into it.  torch frees a graph pool when the last graph
    using it is destroyed (and asserts if its handle is used again); a trainer that drops every captured step at a
    density-control event and captures again right after would allocate a fresh pool -- hundreds of MB of hipMalloc --
    every time.  A one-node graph captured into the pool and kept here holds it open, so every later capture reuses
    the same blocks."""

    def __init__(self, device):
        import torch
        self.handle = torch.cuda.graph_pool_handle()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, pool=self.handle, stream=capture_stream(device)):
            self._anchor = torch.zeros(64, device=device)


def may_fork(device=None) -> bool:
    """May an operator fork work onto a second stream from the current stream?  Outside a capture: unless the current
    stream carries a whole forked pass (leaf_stream); inside one only from the capture's origin stream (a capture
    somebody else opened: origin unknown -> no)."""
    import torch
    if _LEAF_STREAMS and any(torch.cuda.current_stream(device) == s for s in _LEAF_STREAMS):
        return False
    if not torch.cuda.is_current_stream_capturing():
        return True
    return _CAPTURE_ORIGIN is not None and torch.cuda.current_stream(device) == _CAPTURE_ORIGIN


# streams that carry a whole forked pass (renderer.render_fuse): operators running on them never fork further, eagerly
# or captured, so that the pass is the same chain of launches in both modes
_LEAF_STREAMS = []


def leaf_stream(stream):
    if all(stream != s for s in _LEAF_STREAMS):
        _LEAF_STREAMS.append(stream)


_PROTOS = {
    "instag_last_error": (C.c_char_p, []),
    "instag_abi_version": (C.c_int, []),
    "instag_grid_encode_forward": (C.c_int, [vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, u32, C.c_int, u32, vp]),
    "instag_grid_encode_backward": (C.c_int, [vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, vp, vp, u32,
                                              C.c_int, u32, vp]),
    "instag_grid_total_variation_workspace_bytes": (sz, [u32, u32]),
    "instag_grid_total_variation": (C.c_int, [vp, vp, vp, vp, f32, u32, u32, u32, u32, f32, u32, u32, C.c_int,
                                              u32, vp, sz, vp]),
    "instag_triplane_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, u32, f32, u32, u32, f32, u32, f32, u32, vp]),
    "instag_triplane_backward_workspace_bytes": (sz, [u32, u32]),
    "instag_triplane_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, u32, f32, vp,
                                           u32, u32, f32, u32, f32, u32, vp, vp, vp]),
    "instag_sh_encode_forward": (C.c_int, [vp, vp, u32, u32, u32, vp, vp]),
    "instag_sh_encode_backward": (C.c_int, [vp, vp, u32, u32, u32, vp, vp, vp]),
    "instag_raster_geom_bytes": (sz, [i32]),
    "instag_raster_image_bytes": (sz, [i32, i32]),
    "instag_raster_binning_bytes": (sz, [i64, i32, i32]),
    "instag_raster_backward_workspace_bytes": (sz, [i32, i64]),
    "instag_raster_forward_stage1": (C.c_int, [C.POINTER(RasterArgs), vp, sz, vp, C.POINTER(i64), vp]),
    "instag_raster_forward_stage2": (C.c_int, [C.POINTER(RasterArgs), vp, sz, vp, sz, vp, sz, i64,
                                               vp, vp, vp, vp, vp, vp, vp, vp]),
    "instag_raster_forward_capacity": (C.c_int, [C.POINTER(RasterArgs), vp, sz, vp, sz, vp, sz, i64, vp, vp,
                                                 vp, vp, vp, vp, vp, vp, vp, vp]),
    "instag_raster_aux_backward": (C.c_int, [C.POINTER(RasterArgs), vp, sz, vp, sz, vp, sz, i64, vp, vp, vp, vp, sz,
                                             vp, vp, vp]),
    "instag_raster_backward": (C.c_int, [C.POINTER(RasterArgs), vp, sz, vp, sz, vp, sz, i64, vp,
                                         vp, vp, vp, vp, vp, vp, sz,
                                         vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "instag_raster_debug_export": (C.c_int, [vp, sz, vp, sz, vp, sz, i32, i64, i32, i32,
                                             vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "instag_raster_debug_export_flags": (C.c_int, [vp, sz, i32, vp, vp]),
    "instag_debug_depth_sort": (C.c_int, [C.POINTER(RasterArgs), vp, sz, vp, vp, vp]),
    "instag_debug_depth_sort_blocks": (C.c_uint32, [i32]),
    "instag_raster_sort_stalls": (C.c_int, [vp, i32, vp]),
    "instag_raster_sort_stalls_clear": (C.c_int, [vp]),
    "instag_debug_scan_stall_probe": (C.c_int, [vp]),
    "instag_mlp_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "instag_mlp_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "instag_mlp_backward_add": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "instag_linear_weight_grad_batched": (C.c_int, [vp, i32, vp, sz, vp]),
    "instag_linear_weight_grad_batched_glue": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, i32, i32, vp, sz, vp]),
    "instag_mlp_backward_glue_supported": (C.c_int, [i32] * 6),
    "instag_mlp_backward_glue_num_partials": (C.c_int, [i32]),
    "instag_mlp_backward_glue": (C.c_int, [vp] * 18 + [i32] * 3 + [vp]),
    "instag_mlp_forward_glue": (C.c_int, [vp] * 13 + [i32] * 3 + [vp]),
    "instag_extreme_values_workspace_bytes": (sz, [i32, i32]),
    "instag_extreme_values": (C.c_int, [vp, i32, i32, vp, vp, vp, sz, vp]),
    "instag_jaw_feature_workspace_bytes": (sz, [i32, i32]),
    "instag_jaw_feature": (C.c_int, [vp, i32, i32, i32, f32, i32, vp, i32, vp, vp, sz, vp]),
    "instag_mlp2_supported": (C.c_int, [i32] * 5),
    "instag_mlp2_forward": (C.c_int, [vp] * 9 + [i32] * 6 + [vp]),
    "instag_mlp2_backward": (C.c_int, [vp] * 12 + [i32] * 6 + [vp]),
    "instag_linear_weight_grad_workspace_bytes": (sz, [i32, i32, i32]),
    "instag_linear_weight_grad": (C.c_int, [vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "instag_motion_glue_forward": (C.c_int, [vp] * 7 + [i32] * 4 + [vp]),
    "instag_motion_glue_backward_num_partials": (C.c_int, [i32, i32, i32, i32]),
    "instag_motion_glue_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "instag_deform_activate_num_reg_partials": (C.c_int, [i32]),
    "instag_deform_activate_forward": (C.c_int, [vp] * 11 + [f32, i32, vp]),
    "instag_deform_activate_backward": (C.c_int, [vp] * 16 + [f32, i32, vp]),
    "instag_abs_mean_num_partials": (C.c_int, [i32]),
    "instag_abs_mean_forward": (C.c_int, [vp, i32, i32, i32, f32, vp, vp]),
    "instag_abs_mean_backward": (C.c_int, [vp, vp, i32, i32, i32, f32, vp, vp]),
    "instag_mouth_glue_forward": (C.c_int, [vp] * 5 + [i32] * 4 + [vp]),
    "instag_mouth_glue_backward_num_partials": (C.c_int, [i32]),
    "instag_mouth_glue_backward": (C.c_int, [vp] * 4 + [i32] * 4 + [vp]),
    "instag_fuse_compose_forward": (C.c_int, [vp] * 8 + [i32, i32, vp]),
    "instag_fuse_compose_backward": (C.c_int, [vp] * 10 + [i32, i32, vp]),
    "instag_mouth_activate_forward": (C.c_int, [vp] * 6 + [f32] * 3 + [vp] * 4 + [i32, vp]),
    "instag_mouth_activate_backward": (C.c_int, [vp] * 5 + [f32] * 3 + [vp] * 10 + [i32, vp]),
    "instag_motion_l1_reg_num_partials": (C.c_int, [i32]),
    "instag_motion_l1_reg_forward": (C.c_int, [vp, vp, vp, i32, vp]),
    "instag_motion_l1_reg_backward": (C.c_int, [vp, vp, vp, vp, vp, i32, vp]),
    "instag_knn3_mean_dist2": (C.c_int, [vp, vp, i32, vp]),
    "instag_densify_stats": (C.c_int, [vp, vp, vp, vp, vp, i32, vp]),
    "instag_densify_stats_add": (C.c_int, [vp] * 6 + [i32, vp]),
    "instag_frame_code_saved_floats": (C.c_int64, [i32, i32, i32]),
    "instag_frame_code_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "instag_frame_code_backward_workspace_bytes": (sz, [i32, i32, i32]),
    "instag_frame_code_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "instag_l1_ssim_num_partials": (C.c_int, [i32, i32, i32]),
    "instag_l1_ssim_forward": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "instag_l1_ssim_backward": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    "instag_face_loss_num_partials": (C.c_int64, [i32, i32]),
    "instag_face_loss_forward": (C.c_int, [vp] * 11 + [i32] + [vp] * 4),
    "instag_face_loss_backward": (C.c_int, [vp] * 16),
    "instag_face_loss_forward_deferred": (C.c_int, [vp] * 11 + [i32] + [vp] * 3),
    "instag_face_loss_backward_deferred": (C.c_int, [vp] * 11 + [i32] + [vp] * 7),
    "instag_geometry_prior_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float,
                                                vp, vp, vp, vp]),
    "instag_geometry_prior_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float,
                                                 vp, vp, vp, vp, vp, vp]),
    "instag_adam_chunk_elems": (C.c_int, []),
    "instag_adam_step": (C.c_int, [vp, i32, vp, vp, vp, i32, vp, vp]),
    "instag_adam_grads_max": (C.c_int, []),
    "instag_adam_step_grads": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, vp, vp]),
    "instag_adam_step_grads_ticketed": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp]),
    "instag_prof_enable": (C.c_int, [C.c_int]),
    "instag_prof_reset": (C.c_int, []),
    "instag_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(i64)]),
    "instag_prof_graph_begin": (C.c_int, [i32]),
    "instag_prof_graph_pairs_used": (C.c_int, []),
    "instag_prof_graph_collect": (C.c_int, []),
    "instag_prof_graph_end": (C.c_int, []),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


ABI_VERSION = 10    # instag_abi_version() in csrc/raster_api.hip: a stale libinstag_hip.so must not be driven with these prototypes


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            from . import build as _build
            _build.build(verbose=False)          # raises if hipcc is missing: no silent fallback
        # torch first: its wheel bundles its own libamdhip64; loaded before ours, the dynamic linker resolves our
        # DT_NEEDED entry to that same runtime (one HIP runtime per process).  The other order gives two runtimes and
        # "no ROCm-capable device is detected" from the second one.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.instag_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {handle.instag_abi_version()}, the bindings expect "
                               f"{ABI_VERSION}: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = handle
    return _lib


def check(code: int, what: str = ""):
    """Turn a non-zero C-ABI return code into RuntimeError (the reference raises from C++)."""
    if code != 0:
        msg = lib().instag_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg}" if what else msg)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
