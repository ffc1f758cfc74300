#!/usr/bin/env python
"""Phase timing of the depth sort's radix passes (instag_debug_depth_sort): per-block 100 MHz stamps.
    python scripts/bench_sort.py [N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import _lib                                     # noqa: E402
from instag_amd._lib import check, ptr                          # noqa: E402
from instag_amd.diff_gauss import _make_args                    # noqa: E402
from tests.helpers import hip_settings, make_scene              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
a, sd = make_scene(N, 512, sh_degree=1)
s = hip_settings(sd)
g = {k: v.cuda().contiguous() for k, v in a.items()}
args, keep, _, _, _ = _make_args(s, g["means3D"], g["shs"], None, g["opacities"], g["scales"], g["rotations"], None,
                                 g["extra"])
L = _lib.lib()
geom = torch.empty(L.instag_raster_geom_bytes(N), dtype=torch.uint8, device="cuda")
nb = L.instag_debug_depth_sort_blocks(N)
stamps = torch.zeros(4, nb, 8, dtype=torch.int64, device="cuda")
order = torch.zeros(N, dtype=torch.int32, device="cuda")
for it in range(3):
    check(L.instag_debug_depth_sort(C.byref(args), ptr(geom), geom.numel(), ptr(stamps), ptr(order),
                                    _lib.current_stream()), "debug_depth_sort")
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.int64)
V = s.viewmatrix.cpu()
z = (a["means3D"] @ V[:3, 2] + V[3, 2]).numpy().astype(np.float32)
want = np.argsort(z.view(np.uint32), kind="stable")
print("order correct:", bool(np.array_equal(order.cpu().numpy(), want)), " blocks per pass:", nb)
names = ["load keys", "rank", "scans", "look-back", "-", "lds reorder + stores"]
for p in range(4):
    t0 = st[p, :, 0].min()
    print(f"pass {p}: block starts {0.01 * (st[p, :, 0] - t0).min():.2f}..{0.01 * (st[p, :, 0] - t0).max():.2f} us, "
          f"last end {0.01 * (st[p, :, 5].max() - t0):.2f} us")
    d = np.diff(st[p, :, :6], axis=1) * 0.01
    d = np.concatenate([d[:, :3], (st[p, :, 4:5] - st[p, :, 3:4]) * 0.01, (st[p, :, 5:6] - st[p, :, 4:5]) * 0.01], 1)
    for k, nm in enumerate(["load keys+hist", "rank", "local scans", "look-back", "lds reorder+stores"]):
        print(f"    {nm:22s} mean {d[:, k].mean():6.2f}  max {d[:, k].max():6.2f} us")
