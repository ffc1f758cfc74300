"""Weight gradients of the fused MLPs, collected during ``loss.backward()`` and computed afterwards in ONE launch.

Only the optimizer reads a weight gradient; the rest of the backward pass needs the input gradient.  A train step has
nine such GEMMs (three MLPs of the universal field, one of the personalised field), each too small to fill the chip.

    with deferred_grads(device):
        loss.backward()            # instag_amd.mlp queues (dZ, input, weight) instead of launching
    # on exit: one instag_linear_weight_grad_batched call on the current stream, results accumulated into weight.grad

The block also lets the rasterizer's auxiliary-image backward run past the end of the rasterizer's own backward
(``diff_gauss.DEFER_AUX_JOIN``): the glue operator that consumes those gradients joins it, the block's exit at the latest.

Autograd receives ``None`` for those weights (it would otherwise copy or add a still-unwritten buffer), so the block
itself performs ``w.grad = dw`` / ``w.grad += dw``.  Outside a block nothing is queued and gradients flow through
autograd as usual.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _keepalive, _lib
from ._lib import check, ptr

_STATE = {"active": False, "jobs": [], "streams": {}, "forked": set(), "extra": []}
MAX_JOBS = 16


def flush_async(device):
    """Launch the weight gradients queued so far NOW, on a side stream behind the current stream's tail, and keep
    going: called by the trainer at a point of the backward pass where most jobs are queued and a long chain that
    does not depend on them is about to start (the personalised field's backward).  The block's exit joins."""
    if not _STATE["active"] or not _STATE["jobs"]:
        return
    device = torch.device(device)
    jobs, _STATE["jobs"] = _STATE["jobs"], []
    key = (device.type, device.index)
    side = _STATE["streams"].get(key)
    if side is None:
        side = _STATE["streams"][key] = _lib.side_stream(device, "weight_grads")
    main = torch.cuda.current_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        _flush(jobs)
    for dz, inp, _, virt in jobs:
        for t in (dz, inp) + tuple(virt or ()):
            _keepalive.cross_stream(t, side)
    _STATE["forked"].add(key)


_MILESTONES = {}


def on_milestone(name, fn):
    """Run ``fn()`` once, when the backward pass reaches ``milestone(name)`` (right after that operator's backward
    kernel was enqueued; the current stream is the operator's).  A trainer forks side work there that must not start
    earlier -- e.g. an optimizer launch that would otherwise share the chip with the largest MLP backward."""
    _MILESTONES[name] = fn


def clear_milestones():
    _MILESTONES.clear()


def milestone(name):
    fn = _MILESTONES.pop(name, None)
    if fn is not None:
        fn()


def active() -> bool:
    return _STATE["active"]


def join_at_exit(stream):
    """An operator's backward left work on `stream` that nothing downstream is certain to wait for (the consumer may be
    frozen): the block's exit makes the current stream wait for it.  Only inside an active block."""
    assert _STATE["active"]
    if all(stream is not s for s in _STATE["extra"]):
        _STATE["extra"].append(stream)


def defer_weight_grad(dz, inp, weight, virt=None):
    """Queue dW = dz^T @ inp for the leaf parameter ``weight`` ([O,K]); dz [N,O], inp [N,K] contiguous fp32.
    ``virt`` = (aud, eye_pre, enc_a, enc_e): the input rows were never stored, they are cat(inp, aud * enc_a,
    relu(eye_pre) * enc_e) (glue._GlueSigma; instag_linear_weight_grad_batched_glue assembles them while loading)."""
    _STATE["jobs"].append((dz, inp, weight, virt))


def _chunks(group):
    """Launches of at most MAX_JOBS jobs, at most one of them with virtual input rows."""
    out, cur = [], []
    for job in group:
        if len(cur) == MAX_JOBS or (job[3] is not None and any(j[3] is not None for j in cur)):
            out.append(cur)
            cur = []
        cur.append(job)
    if cur:
        out.append(cur)
    return out


def _flush(jobs):
    L = _lib.lib()
    by_n = {}
    for job in jobs:
        by_n.setdefault((job[0].device, job[0].shape[0]), []).append(job)
    for (dev, N), group in by_n.items():
        for chunk in _chunks(group):
            arr = (_lib.WgradJob * len(chunk))()
            dws, total, glue = [], 0, None
            for i, (dz, inp, w, virt) in enumerate(chunk):
                O, K = dz.shape[1], inp.shape[1]
                if virt is not None:
                    glue = (i, virt)
                    K += virt[0].shape[1] + virt[1].shape[1]
                dw = torch.empty(O, K, dtype=torch.float32, device=dev)
                dws.append(dw)
                arr[i] = _lib.WgradJob(dz.data_ptr(), inp.data_ptr(), dw.data_ptr(), N, O, K)
                total += (L.instag_linear_weight_grad_workspace_bytes(N, O, K) + 255) // 256 * 256
            ws = torch.empty(total, dtype=torch.uint8, device=dev)
            if glue is None:
                check(L.instag_linear_weight_grad_batched(arr, len(chunk), ptr(ws), total, _lib.current_stream()),
                      "linear_weight_grad_batched")
            else:
                i, (aud, eye_pre, enc_a, enc_e) = glue
                check(L.instag_linear_weight_grad_batched_glue(arr, len(chunk), i, ptr(aud), ptr(eye_pre), ptr(enc_a),
                                                               ptr(enc_e), aud.shape[1], eye_pre.shape[1], ptr(ws),
                                                               total, _lib.current_stream()),
                      "linear_weight_grad_batched_glue")
            for (dz, inp, w, _), dw in zip(chunk, dws):
                dw = dw.reshape(w.shape).to(w.dtype)
                w.grad = dw if w.grad is None else w.grad.add_(dw)


class deferred_grads:
    def __init__(self, device):
        self.device = None if device is None else torch.device(device)
        self.on = self.device is not None and self.device.type == "cuda"

    def __enter__(self):
        self.prev = _STATE["active"]
        if self.on:
            _STATE["active"] = True
            from . import diff_gauss
            self.prev_aux = diff_gauss.DEFER_AUX_JOIN
            diff_gauss.DEFER_AUX_JOIN = True       # joined by the glue operator, at the latest on exit below
        return self

    def __exit__(self, exc_type, exc, tb):
        _STATE["active"] = self.prev
        if self.on:
            from . import diff_gauss
            diff_gauss.DEFER_AUX_JOIN = self.prev_aux
            diff_gauss.join_pending_aux(final=True)
        if not self.on or self.prev:
            return False
        jobs, _STATE["jobs"] = _STATE["jobs"], []
        if exc_type is None and jobs:
            _flush(jobs)
        for typ, idx in _STATE["forked"]:
            torch.cuda.current_stream(torch.device(typ, idx)).wait_stream(_STATE["streams"][(typ, idx)])
        _STATE["forked"] = set()
        for st in _STATE["extra"]:
            torch.cuda.current_stream(st.device).wait_stream(st)
        _STATE["extra"] = []
        return False
