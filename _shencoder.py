"""Backend module ``_shencoder``: the two entry points of the reference's pybind extension
(/root/reference/shencoder/src/bindings.cpp, shencoder/src/shencoder.h:8-9), same names and argument order, served
by libinstag_hip.so's C ABI.  With this file on the import path the reference's own
``shencoder/sphere_harmonics.py`` binds unchanged (``import _shencoder as _backend``, :9-10)."""
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
        raise RuntimeError(f"{name} must be a float32 tensor")
    return t


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    """shencoder.h:8 -- outputs [B, C*C] (and dy_dx [B, 3*C*C] when given) are written in place."""
    check(_lib.lib().instag_sh_encode_forward(ptr(_f32("inputs", inputs)), ptr(_f32("outputs", outputs)), int(B),
                                              int(D), int(C), ptr(_f32("dy_dx", dy_dx, optional=True)),
                                              _lib.current_stream()), "sh_encode_forward")


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    """shencoder.h:9 -- accumulates into the (zero-filled) grad_inputs [B, 3]."""
    check(_lib.lib().instag_sh_encode_backward(ptr(_f32("grad", grad)), ptr(_f32("inputs", inputs)), int(B), int(D),
                                               int(C), ptr(_f32("dy_dx", dy_dx)), ptr(_f32("grad_inputs", grad_inputs)),
                                               _lib.current_stream()), "sh_encode_backward")
