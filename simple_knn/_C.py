"""``from simple_knn._C import distCUDA2`` (scene/gaussian_model.py:20) on MI355X, backed by libinstag_hip.so."""
import torch

from instag_amd import _lib
from instag_amd._lib import check, ptr


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [N,3] on the GPU -> [N] mean squared distance to the three nearest other points."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be a CUDA/HIP tensor (no CPU path)")
    pts = points.contiguous().float()
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError("distCUDA2: points must be [N,3]")
    out = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device)
    check(_lib.lib().instag_knn3_mean_dist2(ptr(pts), ptr(out), pts.shape[0], _lib.current_stream()), "distCUDA2")
    return out
