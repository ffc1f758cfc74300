"""Image losses of the InsTaG train step (pure torch; run on whatever device the images are on).

Counterpart of /root/reference/utils/loss_utils.py: l1_loss :26-27, gaussian window :33-40,
ssim :42-72 (11x11 window, sigma 1.5, depthwise conv, C1=0.01^2, C2=0.03^2), normalize,
and utils/image_utils.py psnr.  Pinned by tests/golden/g3_losses.npz.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_WINDOWS = {}


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _window(size, channel, device, dtype):
    key = (size, channel, str(device), dtype)
    w = _WINDOWS.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2 * 1.5 ** 2)) for x in range(size)], dtype=torch.float32)
        g = (g / g.sum()).unsqueeze(1)
        w2d = (g @ g.t()).unsqueeze(0).unsqueeze(0)
        w = w2d.expand(channel, 1, size, size).contiguous().to(device=device, dtype=dtype)
        _WINDOWS[key] = w
    return w


def ssim(img1, img2, window_size=11, size_average=True):
    squeeze = img1.dim() == 3
    if squeeze:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    ch = img1.size(-3)
    w = _window(window_size, ch, img1.device, img1.dtype)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, w, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=ch) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


class _FusedL1SSIM(torch.autograd.Function):
    """(l1_mean, ssim_mean) of two [C,H,W] images in one HIP kernel each way (csrc/ssim.hip)."""

    @staticmethod
    def forward(ctx, img1, img2):
        from . import _lib
        from ._lib import check, ptr
        L = _lib.lib()
        a, b = img1.contiguous().float(), img2.contiguous().float()
        C, H, W = a.shape
        nb = L.instag_l1_ssim_num_partials(C, H, W)
        maps = torch.empty(3, C, H, W, dtype=torch.float32, device=a.device)
        parts = torch.empty(2, nb, dtype=torch.float32, device=a.device)
        check(L.instag_l1_ssim_forward(ptr(a), ptr(b), C, H, W, ptr(maps), ptr(parts[0]), ptr(parts[1]),
                                       _lib.current_stream()), "l1_ssim_forward")
        sums = parts.sum(dim=1) / float(C * H * W)
        ctx.save_for_backward(a, b, maps)
        return sums[1], sums[0]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        from . import _lib
        from ._lib import check, ptr
        a, b, maps = ctx.saved_tensors
        C, H, W = a.shape
        g_l1 = g_l1.contiguous().float()
        g_ssim = g_ssim.contiguous().float()
        d = torch.empty_like(a)
        check(_lib.lib().instag_l1_ssim_backward(ptr(a), ptr(b), ptr(maps), ptr(g_ssim), ptr(g_l1), C, H, W, ptr(d),
                                                 _lib.current_stream()), "l1_ssim_backward")
        return d, None


def l1_and_ssim(img1, img2):
    """Returns (l1_loss(img1, img2), ssim(img1, img2)); fused HIP kernels for [3,H,W] device images, the
    torch formulation otherwise (host-side tests).  ``img2`` is treated as a constant."""
    if img1.is_cuda and img1.dim() == 3:
        return _FusedL1SSIM.apply(img1, img2.detach())
    return l1_loss(img1, img2), ssim(img1, img2)
