"""Per-frame conditioning codes of a motion network as one HIP workgroup per pass (csrc/audio.hip).

Replaces, on the device, the reference's chain
    enc_a = audio_att_net(audio_net(a).unsqueeze(0))          scene/motion_net.py:283-289 / :672-677
    enc_e = cat(exp_encode_net(e[:-1]), e[-1:])               scene/motion_net.py:297-299 / :684-686
(~11 conv1d / GEMM launches plus activations, three times that in backward) by one launch each way.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr

NPARAM = 26
_ARRIVAL_SLOTS = 64


def _ptr_array(tensors):
    arr = (C.c_void_p * NPARAM)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


SPLIT_BACKWARD = True        # False: the single-workgroup backward kernel (tests compare the two)

class _FrameCodes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, e, dims, arrivals, *params):
        L = _lib.lib()
        dim_in, mid, dim_aud = dims
        a = a.contiguous().float()
        params = tuple(None if p is None else p.contiguous() for p in params)
        dev = a.device
        n_saved = L.instag_frame_code_saved_floats(dim_in, mid, dim_aud)
        if n_saved < 0:
            raise RuntimeError("frame_codes: unsupported dimensions")
        enc_a = torch.empty(1, dim_aud, dtype=torch.float32, device=dev)
        enc_e = None
        if e is not None:
            e = e.contiguous().float()
            enc_e = torch.empty(6, dtype=torch.float32, device=dev)
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        check(L.instag_frame_code_forward(ptr(a), ptr(e), _ptr_array(params), ptr(enc_a), ptr(enc_e), ptr(saved),
                                          dim_in, mid, dim_aud, ptr(arrivals), _lib.current_stream()),
              "frame_code_forward")
        ctx.dims = dims
        ctx.has_e = e is not None
        ctx.save_for_backward(a, saved, *( [e] if e is not None else [] ), *[p for p in params if p is not None])
        ctx.param_mask = [p is not None for p in params]
        if enc_e is None:
            return enc_a, torch.empty(0, device=dev)
        return enc_a, enc_e

    @staticmethod
    def backward(ctx, d_enc_a, d_enc_e):
        L = _lib.lib()
        dim_in, mid, dim_aud = ctx.dims
        tensors = list(ctx.saved_tensors)
        a, saved = tensors[0], tensors[1]
        e = tensors[2] if ctx.has_e else None
        rest = iter(tensors[3 if ctx.has_e else 2:])
        params = [next(rest) if m else None for m in ctx.param_mask]
        if d_enc_a is None:
            d_enc_a = torch.zeros(1, dim_aud, dtype=torch.float32, device=a.device)
        d_enc_a = d_enc_a.contiguous().float()
        d_enc_e = d_enc_e.contiguous().float() if (ctx.has_e and d_enc_e is not None) else None
        grads = [None if p is None else torch.empty_like(p) for p in params]
        # eight workgroups (one per audio window), each with its own row of parameter gradients in `ws`
        ws = torch.empty(L.instag_frame_code_backward_workspace_bytes(dim_in, mid, dim_aud), dtype=torch.uint8,
                         device=a.device) if SPLIT_BACKWARD else None
        check(L.instag_frame_code_backward(ptr(a), ptr(e), _ptr_array(params), ptr(saved), ptr(d_enc_a), ptr(d_enc_e),
                                           _ptr_array(grads), dim_in, mid, dim_aud, ptr(ws),
                                           0 if ws is None else ws.numel(), _lib.current_stream()),
              "frame_code_backward")
        return (None, None, None, None, *grads)


def _module_params(field):
    """The 26 parameters in the C ABI's order, or None when the modules are not the stock architecture."""
    an, att = field.audio_net, field.audio_att_net
    try:
        convs = [an.encoder_conv[i] for i in (0, 2, 4, 6)]
        fcs = [an.encoder_fc1[i] for i in (0, 2)]
        aconvs = [att.attentionConvNet[i] for i in (0, 2, 4, 6, 8)]
        lin = att.attentionNet[0]
    except (IndexError, AttributeError, TypeError):
        return None
    mid = convs[0].out_channels
    chans = [(c.in_channels, c.out_channels) for c in convs]
    if chans != [(an.encoder_conv[0].in_channels, mid), (mid, mid), (mid, 64), (64, 64)]:
        return None
    if [(c.in_channels, c.out_channels) for c in aconvs] != [(att.dim_aud, 16), (16, 8), (8, 4), (4, 2), (2, 1)]:
        return None
    if att.seq_len != 8 or an.win_size != 16 or fcs[0].in_features != 64 or fcs[0].out_features != 64 \
            or fcs[1].out_features != att.dim_aud or any(m.bias is None for m in convs + fcs + aconvs + [lin]):
        return None
    out = []
    for m in convs + fcs + aconvs + [lin]:
        out += [m.weight, m.bias]
    if getattr(field, "exp_eye", False):
        net = field.exp_encode_net.net
        if len(net) != 2 or tuple(net[0].weight.shape) != (16, 5) or tuple(net[1].weight.shape) != (5, 16):
            return None
        out += [net[0].weight, net[1].weight]
    else:
        out += [None, None]
    return out


def supported(field, a, e) -> bool:
    if not (a.is_cuda and a.dim() == 3 and a.shape[0] == 8 and a.shape[2] == 16):
        return False
    if e is not None and (e.numel() != 6 or not getattr(field, "exp_eye", False)):
        return False
    if e is None and getattr(field, "exp_eye", False):
        return False
    params = _module_params(field)
    if params is None or a.shape[1] != field.audio_net.encoder_conv[0].in_channels:
        return False
    L = _lib.lib()
    return L.instag_frame_code_saved_floats(a.shape[1], field.audio_net.encoder_conv[0].out_channels,
                                            field.audio_att_net.dim_aud) > 0


def frame_codes(field, a, e):
    """-> (enc_a [1, dim_aud], enc_e [6] or None) for a motion network `field` (UMF or PMF)."""
    params = _module_params(field)
    dims = (int(a.shape[1]), int(field.audio_net.encoder_conv[0].out_channels), int(field.audio_att_net.dim_aud))
    # arrival counter of the forward's eight workgroups: one word per network (the universal and the personalised
    # field's branches run on different streams at the same time), zero between launches
    # (and per stream: frames streamed through several lanes at once must not share it either)
    # The words live in ONE table per (network, device), allocated outside any capture: a word first needed inside a
    # capture is a free slot of that table, never memory of the capture's private pool that would outlive its graph.
    # (Every stream of the package is a persistent registry stream, _lib.side_stream: handles are never reused.)
    capturing = torch.cuda.is_current_stream_capturing()
    tables = field.__dict__.setdefault("_frame_code_arrivals", {})
    entry = tables.get(a.device)
    if entry is None and not capturing:
        entry = tables[a.device] = (torch.zeros(_ARRIVAL_SLOTS, dtype=torch.int32, device=a.device), {})
    handle = torch.cuda.current_stream(a.device).cuda_stream
    if entry is not None and (handle in entry[1] or len(entry[1]) < _ARRIVAL_SLOTS):
        slot = entry[1].setdefault(handle, len(entry[1]))
        arrivals = entry[0][slot:slot + 1]
    else:
        arrivals = torch.zeros(1, dtype=torch.int32, device=a.device)      # owned by this call (and its capture) only
    enc_a, enc_e = _FrameCodes.apply(a, None if e is None else e.reshape(-1), dims, arrivals, *params)
    return enc_a, (enc_e if e is not None else None)
