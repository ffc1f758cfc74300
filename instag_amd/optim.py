"""Single-launch multi-tensor Adam / AdamW (HIP, csrc/adam.hip) with a torch.optim-like surface.

Counterpart of the two optimizers of the reference's train loop (train_face.py:59, 781-788;
scene/gaussian_model.py:369-403).  ``param_groups`` / ``state[param]['exp_avg'|'exp_avg_sq']`` behave like
torch.optim.Adam's so that the densify / prune bookkeeping (scene/gaussian_model.py:563-621) works unchanged;
``lr`` of a group may be changed between steps (python float); learning rates and the step counter are kept in
device memory, so a captured hipGraph containing ``step()`` can be replayed.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

_TENSOR_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("group", "<i4"),
                       ("pad", "<i4")])
_GROUP_DT = np.dtype([("beta1", "<f4"), ("beta2", "<f4"), ("eps", "<f4"), ("wd", "<f4"), ("decoupled", "<i4"),
                      ("pad", "<i4")])


class MultiTensorAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled)
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g["params"] = list(g["params"])
            for k, v in self.defaults.items():
                g.setdefault(k, v)
            self.param_groups.append(g)
        self.state = defaultdict(dict)
        self._dev = None
        self._step = None
        self._layout_key = None
        self._partition = None
        self._extra_n, self._extra_vals, self._extra_off = 0, [], 0

    # ---- a few 64-bit words that ride on the learning-rate upload ------------------------------------------------------
    def reserve_extra_i64(self, n: int):
        """``n`` int64 words behind the learning rates in the SAME device table and the same per-step upload: a trainer
        that needs another small per-step value on the device (the mouth stage's random selection size, train_mouth.py:175)
        sets it with set_extra_i64() in front of set_lrs() instead of paying a fill launch of its own per step."""
        self._extra_n, self._extra_vals = int(n), [0] * int(n)
        self._layout_key = None

    def set_extra_i64(self, vals):
        assert len(vals) == self._extra_n
        self._extra_vals = [int(v) for v in vals]

    def extra_i64(self):
        """The device view of the reserved words (None before the first step / prepare(): no table yet)."""
        if self._extra_n == 0 or getattr(self, "_lr_dev", None) is None or self._dev is None:
            return None
        return self._lr_dev[self._extra_off:self._extra_off + 2 * self._extra_n].view(torch.int64)

    # ---- torch.optim surface ----------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    def _state_of(self, p):
        return self.state[p]

    def _ensure_state(self, p):
        st = self._state_of(p)
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def set_lrs(self):
        """Push the groups' current python-float learning rates to the device (one small copy)."""
        if self._dev is None:
            return
        # a ring of pinned rows: the copy reads its row when it EXECUTES, and a host that replays captured steps runs a
        # few steps ahead of the device -- one row would be overwritten with a later step's rates before it was read
        self._lr_slot = (getattr(self, "_lr_slot", -1) + 1) % self._lr_host.shape[0]
        host = self._lr_host[self._lr_slot]
        for i, g in enumerate(self.param_groups):
            host[i] = float(g["lr"])
        if self._extra_n:
            host.numpy()[self._extra_off:self._extra_off + 2 * self._extra_n].view(np.int64)[:] = self._extra_vals
        self._lr_dev.copy_(host, non_blocking=True)

    def _gather(self):
        """[(param, grad or None, exp_avg, exp_avg_sq, group index)] of everything one launch steps."""
        tensors = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if not p.is_cuda:
                    raise RuntimeError("MultiTensorAdam runs on the GPU only")
                if g.get("lazy") and p.grad is None and "exp_avg" not in self._state_of(p):
                    # a group marked "lazy" (the GridRenderer's: its tables never receive a gradient) gets state and work
                    # when a first gradient arrives, like torch.optim.Adam.  Every other parameter is part of the launch
                    # from the first step on, gradient or not, so that the layout a captured step bakes in never changes.
                    continue
                st = self._ensure_state(p)
                grad = p.grad
                if grad is not None and not grad.is_contiguous():
                    grad = grad.contiguous()
                tensors.append((p, grad, st["exp_avg"], st["exp_avg_sq"], gi))
        return tensors

    def _layout(self, tensors):
        """(Re)build the device-side tables when the parameter / state set changed: chunk table, group table, pointer
        table, learning rates, per-tensor step counters.  Allocations and host-to-device copies: never inside a stream
        capture -- a caller that is about to capture step() calls prepare() first."""
        L = _lib.lib()
        dev = tensors[0][0].device
        chunk = L.instag_adam_chunk_elems()
        key = tuple((t[0].data_ptr(), t[0].numel(), t[2].data_ptr()) for t in tensors)
        if self._dev == dev and key == self._layout_key:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("MultiTensorAdam: the parameter set changed since the last step; call prepare() before "
                               "capturing step() into a graph")
        # static part: chunk table, group table, pinned staging buffers
        chunks = [(ti, c) for ti, t in enumerate(tensors) for c in range((t[0].numel() + chunk - 1) // chunk)]
        self._n_late_chunks = len(chunks)
        if self._partition is not None:
            # step(part): "late" tensors' chunks first, "early" ones behind them -- each part is one range of the table
            early = [bool(self._partition(t[0])) for t in tensors]
            chunks = [c for c in chunks if not early[c[0]]] + [c for c in chunks if early[c[0]]]
            self._n_late_chunks = sum(1 for c in chunks if not early[c[0]])
        self._chunks = torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous()
        garr = np.zeros(len(self.param_groups), dtype=_GROUP_DT)
        for i, g in enumerate(self.param_groups):
            garr[i] = (g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], int(bool(g["decoupled"])), 0)
        self._groups_dev = torch.from_numpy(garr.view(np.uint8).copy()).to(dev)
        self._tensors_host = torch.zeros(len(tensors) * _TENSOR_DT.itemsize, dtype=torch.uint8).pin_memory()
        self._tensors_dev = torch.zeros(len(tensors) * _TENSOR_DT.itemsize, dtype=torch.uint8, device=dev)
        n_lr = len(self.param_groups)
        self._extra_off = (n_lr + 1) // 2 * 2          # (8-byte aligned: the table itself is 256-byte aligned)
        words = self._extra_off + 2 * self._extra_n
        self._lr_host = torch.zeros(256, words, dtype=torch.float32).pin_memory()
        self._lr_dev = torch.zeros(words, dtype=torch.float32, device=dev)
        # per-tensor step counters follow their parameter's state across re-layouts (densify / prune)
        steps = torch.zeros(len(tensors), dtype=torch.float32, device=dev)
        for i, t in enumerate(tensors):
            prev = self._state_of(t[0]).get("step")
            if prev is not None:
                steps[i:i + 1].copy_(prev.reshape(1))
        self._step = steps
        self._tickets = torch.zeros(len(tensors), dtype=torch.int32, device=dev)   # adam.hip: TICKET
        for i, t in enumerate(tensors):
            self._state_of(t[0])["step"] = steps[i:i + 1]
        self._dev, self._layout_key = dev, key
        self.set_lrs()
        if len(tensors) <= L.instag_adam_grads_max():
            # parameter / moment pointers sit in the device table (uploaded when the layout changes); the gradient
            # pointers, new every step, travel in the kernel arguments: nothing to copy in front of the launch
            tarr = self._tensors_host.numpy().view(_TENSOR_DT)
            for i, (p, grad, m, v, gi) in enumerate(tensors):
                tarr[i] = (p.data_ptr(), 0, m.data_ptr(), v.data_ptr(), p.numel(), gi, 0)
            self._tensors_dev.copy_(self._tensors_host, non_blocking=True)
            self._grads_host = np.zeros(len(tensors), dtype=np.uint64)

    @torch.no_grad()
    def prepare(self):
        """Bring the device-side tables up to date with the current parameter set WITHOUT stepping (after a densify /
        prune, in front of a stream capture of step())."""
        tensors = self._gather()
        if tensors:
            self._layout(tensors)

    @torch.no_grad()
    def step(self, part=None):
        """``part`` (with a ``partition`` set): "early" / "late" steps only the tensors the partition function sends
        there -- a step as two launches at different points of the backward pass; both parts together are one step()."""
        L = _lib.lib()
        tensors = self._gather()
        if not tensors:
            return
        self._layout(tensors)
        self._keep = [t[1] for t in tensors]          # keep contiguous grad copies alive until the launch ran
        if len(tensors) <= L.instag_adam_grads_max():
            gh = self._grads_host
            for i, t in enumerate(tensors):
                gh[i] = 0 if t[1] is None else t[1].data_ptr()
            first, count = 0, self._chunks.shape[0]
            if part is not None:
                assert self._partition is not None and part in ("early", "late")
                first, count = (0, self._n_late_chunks) if part == "late" else \
                    (self._n_late_chunks, count - self._n_late_chunks)
            # (one launch: the kernel counts the steps itself, per tensor)
            check(L.instag_adam_step_grads_ticketed(ptr(self._tensors_dev), gh.ctypes.data, len(tensors),
                                                    ptr(self._groups_dev), ptr(self._lr_dev),
                                                    self._chunks.data_ptr() + 8 * first, count, ptr(self._step),
                                                    ptr(self._tickets), _lib.current_stream()), "adam_step")
            return
        if part is not None:
            raise RuntimeError("MultiTensorAdam.step(part): more tensors than one launch's gradient table holds")
        tarr = self._tensors_host.numpy().view(_TENSOR_DT)
        for i, (p, grad, m, v, gi) in enumerate(tensors):
            tarr[i] = (p.data_ptr(), 0 if grad is None else grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), gi, 0)
        self._tensors_dev.copy_(self._tensors_host, non_blocking=True)
        check(L.instag_adam_step(ptr(self._tensors_dev), len(tensors), ptr(self._groups_dev), ptr(self._lr_dev),
                                 ptr(self._chunks), self._chunks.shape[0], ptr(self._step), _lib.current_stream()),
              "adam_step")

    def invalidate(self):
        """Call after replacing parameters / state tensors (densify, prune)."""
        self._layout_key = None

    # ---- checkpointing in torch.optim's format (scene/gaussian_model.py:127, :161 store / load optimizer.state_dict()) --
    def state_dict(self):
        index, groups = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                index.setdefault(id(p), len(index))
                ids.append(index[id(p)])
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        state = {}
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state.get(p)
                if st and "exp_avg" in st:
                    step = st.get("step")
                    state[index[id(p)]] = {"step": torch.zeros(()) if step is None else step.detach().reshape(()).clone(),
                                           "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        params = [p for g in self.param_groups for p in g["params"]]
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
        for idx, st in sd["state"].items():
            p = params[int(idx)]
            mine = self._ensure_state(p)
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            step = torch.as_tensor(st.get("step", 0.0), dtype=torch.float32, device=p.device).reshape(1)
            mine["step"] = step.clone()
        self.invalidate()


class CombinedAdam(MultiTensorAdam):
    """One launch for several MultiTensorAdam optimizers (the reference steps ``gaussians.optimizer`` and
    ``motion_optimizer`` back to back, train_face.py:781-788).  The member optimizers keep their own ``param_groups``
    and ``state`` (densify / prune / lr schedules keep working on them); this object only steps them together."""

    def __init__(self, optimizers, partition=None):
        """``partition(param) -> bool``: True = the parameter belongs to step("early"), False to step("late")."""
        self.optimizers = list(optimizers)
        self._dev = None
        self._step = None
        self._layout_key = None
        self._partition = partition
        self._extra_n, self._extra_vals, self._extra_off = 0, [], 0

    @property
    def param_groups(self):
        return [g for o in self.optimizers for g in o.param_groups]

    @property
    def state(self):
        raise AttributeError("CombinedAdam has no state of its own: use the member optimizers")

    def _state_of(self, p):
        for o in self.optimizers:
            for g in o.param_groups:
                for q in g["params"]:
                    if q is p:
                        return o.state[p]
        raise KeyError("parameter does not belong to a member optimizer")

    def zero_grad(self, set_to_none: bool = True):
        for o in self.optimizers:
            o.zero_grad(set_to_none)

    def invalidate(self):
        self._layout_key = None
        for o in self.optimizers:
            o.invalidate()
