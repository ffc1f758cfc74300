"""Fused bias-free ReLU MLP operator (HIP, f32 MFMA) behind ``motion_net.MLP``.

Boundary: the reference's ``MLP.forward`` (scene/motion_net.py:167-173) applied to the per-Gaussian
feature matrix [N, dim_in].  C ABI: instag_mlp_forward / instag_mlp_backward /
instag_linear_weight_grad (include/instag_hip.h).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr


# Weight gradients are only needed by the optimizer, the input gradient by the rest of the backward pass.  With
# ``set_async_weight_grads(True)`` the weight-gradient kernels of every fused MLP go to one side stream and overlap
# the remaining backward chain; the caller MUST call ``join_weight_grads()`` before it reads any ``weight.grad``
# (FaceTrainer does, before the gradient exchange / optimizer steps).  Off by default: a plain drop-in user gets
# gradients that are ordered on the current stream.
_ASYNC = {"enabled": False, "streams": {}}


def set_async_weight_grads(enabled: bool):
    _ASYNC["enabled"] = bool(enabled)


class async_weight_grads:
    """``with async_weight_grads(device): loss.backward()`` -- defers the weight-gradient kernels inside the block and
    joins them with the current stream on exit."""

    def __init__(self, device):
        self.device = device
        self.active = device is not None and torch.device(device).type == "cuda"

    def __enter__(self):
        self.prev = _ASYNC["enabled"]
        if self.active:
            _ASYNC["enabled"] = True
        return self

    def __exit__(self, *exc):
        _ASYNC["enabled"] = self.prev
        if self.active:
            join_weight_grads(torch.device(self.device))
        return False


def _wgrad_stream(device):
    key = (device.type, device.index)
    st = _ASYNC["streams"].get(key)
    if st is None:
        st = _ASYNC["streams"][key] = torch.cuda.Stream(device=device)
    return st


def join_weight_grads(device=None):
    """Make the current stream wait for every deferred weight-gradient kernel."""
    for (typ, idx), st in _ASYNC["streams"].items():
        if device is None or (device.type, device.index) == (typ, idx):
            torch.cuda.current_stream(torch.device(typ, idx)).wait_stream(st)


def supported(dim_in, dim_hidden, dim_out, num_layers) -> bool:
    return num_layers in (2, 3) and 1 <= dim_in <= 96 and 1 <= dim_hidden <= 64 and 1 <= dim_out <= 32


def _weight_grad(L, dz, inp, O, K, stream):
    N = dz.shape[0]
    dw = torch.empty(O, K, dtype=torch.float32, device=dz.device)
    ws = torch.empty(L.instag_linear_weight_grad_workspace_bytes(N, O, K), dtype=torch.uint8, device=dz.device)
    check(L.instag_linear_weight_grad(ptr(dz), ptr(inp), ptr(dw), ptr(ws), ws.numel(), N, O, K, stream),
          "linear_weight_grad")
    return dw


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2, w3):
        L = _lib.lib()
        x = x.contiguous().float()
        w1c, w2c = w1.contiguous().float(), w2.contiguous().float()
        w3c = None if w3 is None else w3.contiguous().float()
        N, K0 = x.shape
        H = w1c.shape[0]
        NL = 2 if w3c is None else 3
        O = (w2c if NL == 2 else w3c).shape[0]
        y = torch.empty(N, O, dtype=torch.float32, device=x.device)
        need = any(ctx.needs_input_grad)
        a1 = torch.empty(N, H, dtype=torch.float32, device=x.device) if need else None
        a2 = torch.empty(N, H, dtype=torch.float32, device=x.device) if (need and NL == 3) else None
        check(L.instag_mlp_forward(ptr(x), ptr(w1c), ptr(w2c), ptr(w3c), ptr(y), ptr(a1), ptr(a2), N, K0, H, O, NL,
                                   _lib.current_stream()), "mlp_forward")
        ctx.save_for_backward(x, w1c, w2c, w3c, a1, a2)
        ctx.dims = (N, K0, H, O, NL)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w1, w2, w3, a1, a2 = ctx.saved_tensors
        N, K0, H, O, NL = ctx.dims
        dy = dy.contiguous().float()
        stream = _lib.current_stream()
        dev = dy.device
        dz1 = torch.empty(N, H, dtype=torch.float32, device=dev)
        dz2 = torch.empty(N, H, dtype=torch.float32, device=dev) if NL == 3 else None
        dx = torch.empty(N, K0, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        check(L.instag_mlp_backward(ptr(dy), ptr(a1), ptr(a2), ptr(w1), ptr(w2), ptr(w3), ptr(dz1), ptr(dz2), ptr(dx),
                                    N, K0, H, O, NL, stream), "mlp_backward")
        def weight_grads(stream):
            dw1 = _weight_grad(L, dz1, x, H, K0, stream) if ctx.needs_input_grad[1] else None
            if NL == 3:
                dw2 = _weight_grad(L, dz2, a1, H, H, stream) if ctx.needs_input_grad[2] else None
                dw3 = _weight_grad(L, dy, a2, O, H, stream) if ctx.needs_input_grad[3] else None
            else:
                dw2 = _weight_grad(L, dy, a1, O, H, stream) if ctx.needs_input_grad[2] else None
                dw3 = None
            return dw1, dw2, dw3

        if _ASYNC["enabled"]:
            main = torch.cuda.current_stream(dev)
            side = _wgrad_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dw1, dw2, dw3 = weight_grads(_lib.current_stream())
            for t in (dz1, dz2, dy, x, a1, a2):
                if t is not None:
                    t.record_stream(side)
            for t in (dw1, dw2, dw3):
                if t is not None:
                    t.record_stream(main)
        else:
            dw1, dw2, dw3 = weight_grads(stream)
        return dx, dw1, dw2, dw3


def fused_mlp(x, weights):
    """x [N, K0] on the GPU; weights = list of 2 or 3 torch.nn.Linear weights ([out, in])."""
    w3 = weights[2] if len(weights) == 3 else None
    return _FusedMLP.apply(x, weights[0], weights[1], w3)
