"""Backend module ``_gridencoder``: the three entry points the reference's pybind extension exports
(/root/reference/gridencoder/src/bindings.cpp:5-7, declared in gridencoder/src/gridencoder.h:12-15), same names,
same argument order, tensors in and out -- served by libinstag_hip.so's C ABI (include/instag_hip.h).

With this file on the import path the reference's own ``gridencoder/grid.py`` binds unchanged:
    try:    import _gridencoder as _backend          (grid.py:9-10)
Ownership as in the reference: the caller allocates every output (``outputs`` [L,B,C], ``dy_dx`` [B, L*D*C],
zero-filled ``grad_embeddings`` / ``grad_inputs``); the functions return None and write in place; argument errors
raise RuntimeError (the reference's TORCH_CHECK / std::runtime_error, gridencoder.cu:15-18,381,398).
"""
import torch

from instag_amd import _lib
from instag_amd._lib import check, ptr


def _f32(name, t, optional=False):
    if t is None:
        if optional:
            return None
        raise RuntimeError(f"{name} must not be None")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor (half-precision embeddings are not built: "
                           f"InsTaG never enables autocast, gridencoder/grid.py:43-44)")
    return t


def _i32(name, t):
    if not t.is_cuda or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous CUDA tensor")
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")
    return t


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                        interp):
    """gridencoder.h:12 -- outputs [L,B,C] (and dy_dx [B, L*D*C] when given) are written in place."""
    check(_lib.lib().instag_grid_encode_forward(
        ptr(_f32("inputs", inputs)), ptr(_f32("embeddings", embeddings)), ptr(_i32("offsets", offsets)),
        ptr(_f32("outputs", outputs)), int(B), int(D), int(C), int(L), float(S), int(H),
        ptr(_f32("dy_dx", dy_dx, optional=True)), int(gridtype), int(bool(align_corners)), int(interp),
        _lib.current_stream()), "grid_encode_forward")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                         gridtype, align_corners, interp):
    """gridencoder.h:13 -- accumulates into the (zero-filled) grad_embeddings [sO,C] and grad_inputs [B,D]."""
    check(_lib.lib().instag_grid_encode_backward(
        ptr(_f32("grad", grad)), ptr(_f32("inputs", inputs)), ptr(_f32("embeddings", embeddings)),
        ptr(_i32("offsets", offsets)), ptr(_f32("grad_embeddings", grad_embeddings)), int(B), int(D), int(C), int(L),
        float(S), int(H), ptr(_f32("dy_dx", dy_dx, optional=True)), ptr(_f32("grad_inputs", grad_inputs, optional=True)),
        int(gridtype), int(bool(align_corners)), int(interp), _lib.current_stream()), "grid_encode_backward")


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    """gridencoder.h:15 -- adds the total-variation gradient of the entries hit by `inputs` into `grad` [sO,C]."""
    lib = _lib.lib()
    total = int(embeddings.shape[0])
    ws = torch.empty(lib.instag_grid_total_variation_workspace_bytes(total, int(C)), dtype=torch.uint8,
                     device=embeddings.device)
    check(lib.instag_grid_total_variation(
        ptr(_f32("inputs", inputs)), ptr(_f32("embeddings", embeddings)), ptr(_f32("grad", grad)),
        ptr(_i32("offsets", offsets)), float(weight), int(B), int(D), int(C), int(L), float(S), int(H), int(gridtype),
        int(bool(align_corners)), total, ptr(ws), ws.numel(), _lib.current_stream()), "grad_total_variation")
