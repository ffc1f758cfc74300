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


def normalize(input, mean=None, std=None):
    """utils/loss_utils.py:17-20: per-row standardisation whose denominator is padded by 1 % of the global std."""
    input_mean = torch.mean(input, dim=1, keepdim=True) if mean is None else mean
    input_std = torch.std(input, dim=1, keepdim=True) if std is None else std
    return (input - input_mean) / (input_std + 1e-2 * torch.std(input.reshape(-1)))


class _GeometryPrior(torch.autograd.Function):
    """geometry_prior_loss on the device: five launches instead of ~100 (csrc/prior.hip)."""

    @staticmethod
    def forward(ctx, normal, depth, gt_normal, gt_depth, face_mask, hair_mask, mouth_mask, use_depth, w_normal, w_depth):
        from . import _lib
        from ._lib import check, ptr
        L = _lib.lib()
        normal = normal.contiguous().float()
        H, W = normal.shape[-2:]
        dev = normal.device
        d = depth.contiguous().float() if use_depth else None
        gtn = gt_normal.contiguous().float()
        gtd = gt_depth.contiguous().float() if use_depth else None
        masks = [m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.contiguous().to(torch.uint8)
                 for m in (face_mask, hair_mask, mouth_mask)]
        ws = torch.empty(14 * H + 5, dtype=torch.float32, device=dev)      # stat 8H | parts 4H | rowb 2H | out 5
        stat, parts, rowb, out = ws[:8 * H], ws[8 * H:12 * H], ws[12 * H:14 * H], ws[14 * H:]
        check(L.instag_geometry_prior_forward(ptr(normal), ptr(d), ptr(gtn), ptr(gtd), ptr(masks[0]), ptr(masks[1]),
                                              ptr(masks[2]), H, W, int(use_depth), float(w_normal), float(w_depth),
                                              ptr(stat), ptr(parts), ptr(out), _lib.current_stream()),
              "geometry_prior_forward")
        ctx.save_for_backward(d, gtn, gtd, *masks, ws)
        ctx.cfg = (H, W, bool(use_depth), float(w_normal), float(w_depth), tuple(depth.shape) if use_depth else None)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        from ._lib import check, ptr
        L = _lib.lib()
        d, gtn, gtd, face, hair, mouth, ws = ctx.saved_tensors
        H, W, use_depth, w_normal, w_depth, dshape = ctx.cfg
        stat, rowb, out = ws[:8 * H], ws[12 * H:14 * H], ws[14 * H:]
        g = g.contiguous().float().reshape(1)
        d_normal = torch.empty(3, H, W, dtype=torch.float32, device=gtn.device)
        d_depth = torch.empty(dshape, dtype=torch.float32, device=gtn.device) if use_depth else None
        check(L.instag_geometry_prior_backward(ptr(g), ptr(d), ptr(gtn), ptr(gtd), ptr(face), ptr(hair), ptr(mouth), H, W,
                                               int(use_depth), w_normal, w_depth, ptr(stat), ptr(out), ptr(rowb),
                                               ptr(d_normal), ptr(d_depth), _lib.current_stream()),
              "geometry_prior_backward")
        return d_normal, d_depth, None, None, None, None, None, None, None, None


def geometry_prior_loss(normal, depth, gt_normal, gt_depth, face_mask, hair_mask, mouth_mask, use_depth=True,
                        w_normal=0.01, w_depth=1e-2):
    """Few-shot geometry priors of the face branch after warm_step + 2000 (train_face.py:458-504): the rendered
    normal against the monocular normal (1 - n_gt * n per channel, summed over channels, mean over head minus
    mouth) and, outside the 100 iterations after an opacity reset, the standardised rendered depth against the
    standardised monocular depth (mean |.| over face minus mouth).  normal [3,H,W], depth [1,H,W] or [H,W].
    Fused HIP kernels on the device (instag_geometry_prior_*), the torch formulation below otherwise."""
    if normal.is_cuda and normal.dim() == 3 and normal.shape[0] == 3:
        return _GeometryPrior.apply(normal, depth if use_depth else None, gt_normal, gt_depth if use_depth else None,
                                    face_mask, hair_mask, mouth_mask, bool(use_depth), w_normal, w_depth)
    return geometry_prior_loss_torch(normal, depth, gt_normal, gt_depth, face_mask, hair_mask, mouth_mask, use_depth,
                                     w_normal, w_depth)


def geometry_prior_loss_torch(normal, depth, gt_normal, gt_depth, face_mask, hair_mask, mouth_mask, use_depth=True,
                              w_normal=0.01, w_depth=1e-2):
    """Plain-torch statement of the same lines (host-side tests, parity reference of the fused kernels)."""
    head = face_mask | hair_mask
    m = (head ^ mouth_mask).to(normal.dtype)
    per_px = (1.0 - gt_normal * normal).sum(0)
    loss = w_normal * (per_px * m).sum() / m.sum()
    if use_depth:
        d = depth[0] if depth.dim() == 3 else depth
        sel = (face_mask ^ mouth_mask).to(d.dtype)       # masked means as sums: no host round trip (graph capture)
        loss = loss + w_depth * ((normalize(d) - normalize(gt_depth)).abs() * sel).sum() / sel.sum()
    return loss


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


# ---- the face branch's whole loss block ---------------------------------------------------------------------------
FLAG_HAIR_TO_BG, FLAG_ALPHA, FLAG_HAIR_ATTN, FLAG_LIPS, FLAG_MOUTH, FLAG_PLAIN = 1, 2, 4, 8, 16, 32


def face_loss_torch(image, gt, face_mask, hair_mask, mouth_mask, bg, alpha=None, attn=None, lips_rect=None,
                    extra=None, lambda_dssim=0.2, w_alpha=1e-3, w_attn=1e-4, w_extra=1e-5, hair_mask_iter=False):
    """Plain-torch statement of train_face.py:415-416, 426-456, 508-575 -> (loss, Ll1).  `alpha` / `attn` /
    `extra` None drops the corresponding warm-stage terms.  lips_rect = (r0, r1, c0, c1) indexes attn[1, r0:r1, c0:c1]."""
    head = face_mask | hair_mask
    bg3 = bg[:, None, None]
    keep = head & ~mouth_mask
    if hair_mask_iter:
        keep = keep & ~hair_mask
        image = torch.where(hair_mask[None], bg3.expand_as(image), image)
    gt_white = torch.where(keep[None], gt, bg3.expand_as(gt))
    Ll1 = l1_loss(image, gt_white)
    loss = Ll1 + lambda_dssim * (1.0 - ssim(image, gt_white))
    if extra is not None:
        loss = loss + w_extra * extra.sum()
    if alpha is not None:
        hm = head.to(alpha.dtype)
        loss = loss + w_alpha * (((1 - alpha) * hm).mean() + (alpha * (1 - hm)).mean())
    if attn is not None:
        if lips_rect is not None:
            r0, r1, c0, c1 = [int(v) for v in (lips_rect.tolist() if torch.is_tensor(lips_rect) else lips_rect)]
            loss = loss + w_attn * attn[1, r0:r1, c0:c1].mean()
        if not hair_mask_iter:
            hair = hair_mask.to(attn.dtype)
            cnt = hair.sum().clamp_min(1.0)
            loss = loss + w_attn * ((attn[1] * hair).sum() / cnt + (attn[0] * hair).sum() / cnt)
    return loss, Ll1


# A caller that runs backward right behind the loss and reads the loss VALUE only afterwards (the train steps) sets this
# around its loss call: the forward then launches the tile kernel only, and the backward launch also produces the
# scalars (csrc/ssim.hip, instag_face_loss_*_deferred) -- one launch less on the step's critical chain.  The returned
# loss / L1 tensors hold their values once backward has run.
DEFER_FINALIZE = False


class defer_finalize:
    def __enter__(self):
        global DEFER_FINALIZE
        self.prev, DEFER_FINALIZE = DEFER_FINALIZE, True

    def __exit__(self, *exc):
        global DEFER_FINALIZE
        DEFER_FINALIZE = self.prev
        return False


class _FusedFaceLoss(torch.autograd.Function):
    """(loss, Ll1) of the face branch in two launches forward, one backward (csrc/ssim.hip)."""

    @staticmethod
    def forward(ctx, image, alpha, attn, extra, gt, face_mask, hair_mask, mouth_mask, bg, lips_rect, cfg_tuple):
        import ctypes as C
        from . import _lib
        from ._lib import check, ptr
        L = _lib.lib()
        ctx.set_materialize_grads(False)
        image = image.contiguous().float()
        _, H, W = image.shape
        dev = image.device
        flags, w_dssim, w_alpha, w_hair, w_lips, w_extra = cfg_tuple
        cfg = _lib.FaceLossCfg(H, W, flags, w_dssim, w_alpha, w_hair, w_lips, w_extra)
        as_u8 = lambda m: None if m is None else (m.contiguous().view(torch.uint8) if m.dtype == torch.bool
                                                  else m.contiguous().to(torch.uint8))
        face_mask, hair_mask, mouth_mask = as_u8(face_mask), as_u8(hair_mask), as_u8(mouth_mask)
        gt, bg = gt.contiguous().float(), (None if bg is None else bg.contiguous().float())
        alpha = None if alpha is None else alpha.contiguous().float()
        attn = None if attn is None else attn.contiguous().float()
        extra_shape = None if extra is None else tuple(extra.shape)
        extra = None if extra is None else extra.reshape(-1).contiguous().float()
        lips_rect = None if lips_rect is None else lips_rect.contiguous().to(torch.int32)
        maps = torch.empty(3, 3, H, W, dtype=torch.float32, device=dev)
        parts = torch.empty(L.instag_face_loss_num_partials(H, W), dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)
        ctx.deferred = bool(DEFER_FINALIZE and any(ctx.needs_input_grad))
        if ctx.deferred:
            check(L.instag_face_loss_forward_deferred(C.byref(cfg), ptr(image), ptr(gt), ptr(face_mask),
                                                      ptr(hair_mask), ptr(mouth_mask), ptr(bg), ptr(alpha), ptr(attn),
                                                      ptr(lips_rect), ptr(extra), 0 if extra is None else extra.numel(),
                                                      ptr(maps), ptr(parts), _lib.current_stream()),
                  "face_loss_forward")
        else:
            check(L.instag_face_loss_forward(C.byref(cfg), ptr(image), ptr(gt), ptr(face_mask), ptr(hair_mask),
                                             ptr(mouth_mask), ptr(bg), ptr(alpha), ptr(attn), ptr(lips_rect), ptr(extra),
                                             0 if extra is None else extra.numel(), ptr(maps), ptr(parts), ptr(out),
                                             _lib.current_stream()), "face_loss_forward")
        ctx.cfg = cfg
        ctx.shapes = (alpha is not None, attn is not None, None if extra is None else extra_shape)
        ctx.has_lips = lips_rect is not None
        ctx.save_for_backward(image, gt, face_mask, hair_mask, mouth_mask, bg, maps, out,
                              *([lips_rect] if lips_rect is not None else []),
                              *([parts, extra] if ctx.deferred else []))
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_loss, g_l1):
        import ctypes as C
        from . import _lib
        from ._lib import check, ptr
        saved = ctx.saved_tensors
        image, gt, face_mask, hair_mask, mouth_mask, bg, maps, out = saved[:8]
        lips_rect = saved[8] if ctx.has_lips else None
        parts, extra = (saved[-2], saved[-1]) if ctx.deferred else (None, None)
        has_alpha, has_attn, has_extra = ctx.shapes
        cfg = ctx.cfg
        _, H, W = image.shape
        g_loss = None if g_loss is None else g_loss.contiguous().float()
        g_l1 = None if g_l1 is None else g_l1.contiguous().float()
        d_image = torch.empty_like(image)
        d_alpha = torch.empty(1, H, W, dtype=torch.float32, device=image.device) if has_alpha else None
        d_attn = torch.empty(3, H, W, dtype=torch.float32, device=image.device) if has_attn else None
        if ctx.deferred:
            check(_lib.lib().instag_face_loss_backward_deferred(
                C.byref(cfg), ptr(image), ptr(gt), ptr(face_mask), ptr(hair_mask), ptr(mouth_mask), ptr(bg),
                ptr(lips_rect), ptr(maps), ptr(parts), ptr(extra), 0 if extra is None else extra.numel(), ptr(out),
                ptr(g_loss), ptr(g_l1), ptr(d_image), ptr(d_alpha), ptr(d_attn), _lib.current_stream()),
                "face_loss_backward")
        else:
            check(_lib.lib().instag_face_loss_backward(C.byref(cfg), ptr(image), ptr(gt), ptr(face_mask),
                                                       ptr(hair_mask), ptr(mouth_mask), ptr(bg), ptr(lips_rect),
                                                       ptr(maps), ptr(out), ptr(g_loss), ptr(g_l1), ptr(d_image),
                                                       ptr(d_alpha), ptr(d_attn), _lib.current_stream()),
                  "face_loss_backward")
        d_extra = None
        if has_extra is not None and g_loss is not None:
            # d loss / d extra[i] = w_extra * g for every element (extra enters as a sum); w_extra == 1 needs no launch
            d_extra = (g_loss if cfg.w_extra == 1.0 else g_loss * cfg.w_extra).expand(has_extra)
        return d_image, d_alpha, d_attn, d_extra, None, None, None, None, None, None, None


def face_loss(image, gt, face_mask, hair_mask, mouth_mask, bg, alpha=None, attn=None, lips_rect=None, extra=None,
              lambda_dssim=0.2, w_alpha=1e-3, w_attn=1e-4, w_extra=1e-5, hair_mask_iter=False):
    """Loss block of the face branch -> (loss, Ll1); fused HIP kernels on the device, face_loss_torch otherwise."""
    if not (image.is_cuda and image.dim() == 3 and image.shape[0] == 3):
        return face_loss_torch(image, gt, face_mask, hair_mask, mouth_mask, bg, alpha, attn, lips_rect, extra,
                               lambda_dssim, w_alpha, w_attn, w_extra, hair_mask_iter)
    flags = (FLAG_HAIR_TO_BG if hair_mask_iter else 0) | (FLAG_ALPHA if alpha is not None else 0)
    if attn is not None:
        flags |= (FLAG_LIPS if lips_rect is not None else 0) | (0 if hair_mask_iter else FLAG_HAIR_ATTN)
    if lips_rect is not None and not torch.is_tensor(lips_rect):
        lips_rect = torch.tensor(list(lips_rect), dtype=torch.int32, device=image.device)
    cfg = (flags, float(lambda_dssim), float(w_alpha), float(w_attn), float(w_attn), float(w_extra))
    return _FusedFaceLoss.apply(image, alpha, attn, extra, gt, face_mask, hair_mask, mouth_mask, bg, lips_rect, cfg)


def plain_loss_fused(image, gt, lambda_dssim=0.2):
    """(Ll1 + lambda_dssim * (1 - ssim), Ll1) of two [3,H,W] device images (train_fuse_con.py:176-181) with the face
    branch's loss kernels in their plain mode: two launches forward, one backward, no scalar arithmetic launches."""
    cfg = (FLAG_PLAIN, float(lambda_dssim), 0.0, 0.0, 0.0, 0.0)
    return _FusedFaceLoss.apply(image, None, None, None, gt.detach(), None, None, None, None, None, cfg)


def mouth_loss_fused(image, alpha, gt, mouth_mask, lips_rect, bg, p_xyz=None, warm=True, lambda_dssim=0.2, p_raw=None):
    """Loss block of the mouth branch (train_mouth.py:186-221) on the device -> (loss, Ll1): the face branch's fused
    kernels in their mouth mode (csrc/ssim.hip F_MOUTH) -- two launches forward, one backward, instead of ~45 elementwise
    / reduce launches.  ``lips_rect`` = int32 [4] (row0, row1, col0, col1) on the device.  ``p_raw`` (optional): the
    alignment head's raw output [N, >=3] that ``p_xyz`` = p_raw[:, :3] * 1e-2 was made of (used instead of p_xyz)."""
    flags = FLAG_MOUTH | (FLAG_ALPHA if warm else 0)
    extra = None
    if warm and p_raw is not None and p_raw.is_cuda and p_raw.dim() == 2 and p_raw.shape[1] >= 3:
        # mean|p_xyz| with p_xyz = p[:, :3] * 1e-2 as partial sums the loss kernel adds up: one launch per pass instead
        # of the slice / scale / abs / mean chain and its five backward launches
        from .glue import abs_mean_partials
        # (the term's weight rides in the scale -- |x * 1e-2| * 1e-5 = |x * 1e-7| -- so the loss kernel's w_extra is 1
        # and the gradient of the partial sums is the loss gradient itself: no scaling launch in backward)
        extra = abs_mean_partials(p_raw, 3, 1e-2 * 1e-5)
        cfg = (flags, float(lambda_dssim), 1e-3, 0.0, 0.0, 1.0)
        return _FusedFaceLoss.apply(image, alpha if warm else None, None, extra, gt, None, None, mouth_mask, bg,
                                    lips_rect.to(torch.int32), cfg)
    elif warm and p_xyz is not None:
        extra = p_xyz.abs().mean().reshape(1)
    cfg = (flags, float(lambda_dssim), 1e-3, 0.0, 0.0, 1e-5)
    return _FusedFaceLoss.apply(image, alpha if warm else None, None, extra, gt, None, None, mouth_mask, bg,
                                lips_rect.to(torch.int32), cfg)
