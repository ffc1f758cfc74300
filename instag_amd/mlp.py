"""Fused bias-free ReLU MLP operator (HIP, f32 MFMA) behind ``motion_net.MLP``.

Boundary: the reference's ``MLP.forward`` (scene/motion_net.py:167-173) applied to the per-Gaussian
feature matrix [N, dim_in].  C ABI: instag_mlp_forward / instag_mlp_backward /
instag_linear_weight_grad (include/instag_hip.h).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr


# multiply-add work launched so far (2 flops per MAC), read by bench.py for the MFMA roofline of the MLP kernels
STATS = {"fwd_flops": 0, "bwd_flops": 0}


def supported(dim_in, dim_hidden, dim_out, num_layers) -> bool:
    return num_layers in (2, 3) and 1 <= dim_in <= 96 and 1 <= dim_hidden <= 64 and 1 <= dim_out <= 32


def _weight_grad(L, dz, inp, dw, ws):
    N, (O, K) = dz.shape[0], dw.shape
    check(L.instag_linear_weight_grad(ptr(dz), ptr(inp), ptr(dw), ptr(ws), ws.numel(), N, O, K,
                                      _lib.current_stream()), "linear_weight_grad")


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
        ctx.weights = (w1, w2, w3)          # the caller's tensors (leaf parameters in the deferred-gradient mode)
        ctx.dims = (N, K0, H, O, NL)
        STATS["fwd_flops"] += 2 * N * (K0 * H + (H * H if NL == 3 else 0) + H * O)
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
        STATS["bwd_flops"] += 2 * N * (H * O + (H * H if NL == 3 else 0) + (K0 * H if dx is not None else 0))
        # weight gradients: only the optimizer reads them.  Inside a ``deferred_grads`` block (instag_amd/deferred.py)
        # they are queued and computed after the rest of the backward pass in one batched launch
        from . import deferred
        layers = [(dz1, x, 1, (H, K0))]
        layers += [(dz2, a1, 2, (H, H)), (dy, a2, 3, (O, H))] if NL == 3 else [(dy, a1, 2, (O, H))]
        layers = [(dz, inp, idx, shp) for dz, inp, idx, shp in layers if ctx.needs_input_grad[idx]]

        def compute():
            out = {}
            for dz, inp, idx, shp in layers:
                dw = torch.empty(*shp, dtype=torch.float32, device=dev)
                ws = torch.empty(L.instag_linear_weight_grad_workspace_bytes(N, shp[0], shp[1]), dtype=torch.uint8,
                                 device=dev)
                _weight_grad(L, dz, inp, dw, ws)
                out[idx] = dw
            return out

        if deferred.active() and all(w is None or w.is_leaf for w in ctx.weights):
            for dz, inp, idx, shp in layers:
                deferred.defer_weight_grad(dz, inp, ctx.weights[idx - 1])
            if EAGER_FLUSH_ROWS and N >= EAGER_FLUSH_ROWS:
                # a per-Gaussian MLP: its weight-gradient GEMMs are the big ones (sigma_net: a third of the step's MFMA
                # work) and their operands exist now -- start them on the side stream beside the rest of the backward
                # chain instead of in one lump beside the personalised field's backward, which they slow down
                deferred.flush_async(dev)
            return dx, None, None, None
        dws = compute()
        dw1, dw2, dw3 = dws.get(1), dws.get(2), dws.get(3)
        return dx, dw1, dw2, dw3


# Rows from which an MLP's weight gradients are launched right after its backward (0 = wait for the flush points of
# renderer.py / gridencoder.py).  Measured on the C3 step (round 2): starting sigma_net's three GEMMs beside the glue
# backward doubles that kernel (33 -> 75 us, it is on the critical path) and the graph executor then queues the
# personalised field's backward behind the next flush: 1.072 ms/step with 0, 1.088 with 32768.
EAGER_FLUSH_ROWS = 0


FUSE_SHARED_HEADS = True      # False: one launch per head (tests compare the two)


class _SharedInputMLPs(torch.autograd.Function):
    """Two 2-layer MLPs over the SAME input x plus x itself as a third output (for the input's other consumers):
    (MLP_a(x), MLP_b(x), x).  The reference runs aud_ch_att_net(enc_x), eye_att_net(enc_x) and then concatenates
    enc_x into sigma_net's input (scene/motion_net.py:281-306), so autograd sums three gradients of enc_x with two
    elementwise launches; here the sum happens inside the two backward kernels (instag_mlp_backward_add)."""

    @staticmethod
    def forward(ctx, x, wa1, wa2, wb1, wb2):
        L = _lib.lib()
        x = x.contiguous().float()
        N, K0 = x.shape
        outs, saved, dims = [], [x], []
        if FUSE_SHARED_HEADS and L.instag_mlp2_supported(K0, wa1.shape[0], wa2.shape[0], wb1.shape[0], wb2.shape[0]):
            # both heads in one launch: the tile of x is loaded once (csrc/mlp.hip: mlp2_forward_kernel)
            ws = [w.contiguous().float() for w in (wa1, wa2, wb1, wb2)]
            (HA, OA), (HB, OB) = (ws[0].shape[0], ws[1].shape[0]), (ws[2].shape[0], ws[3].shape[0])
            ya = torch.empty(N, OA, dtype=torch.float32, device=x.device)
            yb = torch.empty(N, OB, dtype=torch.float32, device=x.device)
            keep = any(ctx.needs_input_grad)          # forward only (inference, the jaw feature): nothing is kept
            a1a = torch.empty(N, HA, dtype=torch.float32, device=x.device) if keep else None
            a1b = torch.empty(N, HB, dtype=torch.float32, device=x.device) if keep else None
            check(L.instag_mlp2_forward(ptr(x), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(ws[3]), ptr(ya), ptr(yb),
                                        ptr(a1a), ptr(a1b), N, K0, HA, OA, HB, OB, _lib.current_stream()),
                  "mlp2_forward")
            STATS["fwd_flops"] += 2 * N * (K0 * HA + HA * OA + K0 * HB + HB * OB)
            ctx.save_for_backward(x, ws[0], ws[1], a1a, ws[2], ws[3], a1b)
            ctx.weights = (wa1, wa2, wb1, wb2)
            ctx.dims = (N, K0, [(HA, OA), (HB, OB)])
            ctx.fused = True
            return ya, yb, x.view_as(x)
        ctx.fused = False
        for w1, w2 in ((wa1, wa2), (wb1, wb2)):
            w1c, w2c = w1.contiguous().float(), w2.contiguous().float()
            H, O = w1c.shape[0], w2c.shape[0]
            y = torch.empty(N, O, dtype=torch.float32, device=x.device)
            a1 = torch.empty(N, H, dtype=torch.float32, device=x.device)
            check(L.instag_mlp_forward(ptr(x), ptr(w1c), ptr(w2c), None, ptr(y), ptr(a1), None, N, K0, H, O, 2,
                                       _lib.current_stream()), "mlp_forward")
            STATS["fwd_flops"] += 2 * N * (K0 * H + H * O)
            outs.append(y)
            saved += [w1c, w2c, a1]
            dims.append((H, O))
        ctx.save_for_backward(*saved)
        ctx.weights = (wa1, wa2, wb1, wb2)
        ctx.dims = (N, K0, dims)
        return outs[0], outs[1], x.view_as(x)

    @staticmethod
    def backward(ctx, dya, dyb, dx_other):
        from . import deferred
        L = _lib.lib()
        x = ctx.saved_tensors[0]
        N, K0, dims = ctx.dims
        dev = x.device
        stream = _lib.current_stream()
        want_dx = ctx.needs_input_grad[0]
        dx = torch.empty(N, K0, dtype=torch.float32, device=dev) if want_dx else None
        add = None if dx_other is None else dx_other.contiguous().float()
        jobs = []
        fused = ctx.fused and dya is not None and dyb is not None
        if fused:
            wa1, wa2, a1a, wb1, wb2, a1b = ctx.saved_tensors[1:7]
            (HA, OA), (HB, OB) = dims
            dya, dyb = dya.contiguous().float(), dyb.contiguous().float()
            dz1a = torch.empty(N, HA, dtype=torch.float32, device=dev)
            dz1b = torch.empty(N, HB, dtype=torch.float32, device=dev)
            check(L.instag_mlp2_backward(ptr(dya), ptr(dyb), ptr(a1a), ptr(a1b), ptr(wa1), ptr(wa2), ptr(wb1),
                                         ptr(wb2), ptr(dz1a), ptr(dz1b), ptr(dx), ptr(add) if want_dx else None,
                                         N, K0, HA, OA, HB, OB, stream), "mlp2_backward")
            STATS["bwd_flops"] += 2 * N * (HA * OA + HB * OB + (K0 * (HA + HB) if want_dx else 0))
            deferred.milestone("heads_backward")
            add = dx if want_dx else add
            jobs = [(dz1a, x, 1, (HA, K0)), (dya, a1a, 2, (OA, HA)), (dz1b, x, 3, (HB, K0)), (dyb, a1b, 4, (OB, HB))]
        for k, dy in enumerate(() if fused else (dya, dyb)):
            if dy is None:
                continue
            w1, w2, a1 = ctx.saved_tensors[1 + 3 * k:4 + 3 * k]
            H, O = dims[k]
            dy = dy.contiguous().float()
            dz1 = torch.empty(N, H, dtype=torch.float32, device=dev)
            check(L.instag_mlp_backward_add(ptr(dy), ptr(a1), None, ptr(w1), ptr(w2), None, ptr(dz1), None, ptr(dx),
                                            ptr(add) if want_dx else None, N, K0, H, O, 2, stream), "mlp_backward")
            STATS["bwd_flops"] += 2 * N * (H * O + (K0 * H if want_dx else 0))
            if want_dx:
                add = dx                      # the next kernel adds onto what is already there
            jobs += [(dz1, x, 1 + 2 * k, (H, K0)), (dy, a1, 2 + 2 * k, (O, H))]
        if want_dx and add is not dx:         # neither head carried a gradient
            dx = torch.zeros(N, K0, dtype=torch.float32, device=dev) if dx_other is None else dx_other
        jobs = [(dz, inp, idx, shp) for dz, inp, idx, shp in jobs if ctx.needs_input_grad[idx]]
        grads = [None] * 4
        if deferred.active() and all(w.is_leaf for w in ctx.weights):
            for dz, inp, idx, shp in jobs:
                deferred.defer_weight_grad(dz, inp, ctx.weights[idx - 1])
        else:
            for dz, inp, idx, shp in jobs:
                dw = torch.empty(*shp, dtype=torch.float32, device=dev)
                ws = torch.empty(L.instag_linear_weight_grad_workspace_bytes(N, shp[0], shp[1]), dtype=torch.uint8,
                                 device=dev)
                _weight_grad(L, dz, inp, dw, ws)
                grads[idx - 1] = dw
        return (dx, *grads)


def shared_input_mlps(x, weights_a, weights_b):
    """(MLP_a(x), MLP_b(x), x_for_other_consumers) for two 2-layer bias-free ReLU MLPs: see _SharedInputMLPs."""
    return _SharedInputMLPs.apply(x, weights_a[0], weights_a[1], weights_b[0], weights_b[1])


def fused_mlp(x, weights):
    """x [N, K0] on the GPU; weights = list of 2 or 3 torch.nn.Linear weights ([out, in])."""
    w3 = weights[2] if len(weights) == 3 else None
    return _FusedMLP.apply(x, weights[0], weights[1], w3)
