"""Where the blend forward's time goes, tile by tile (diagnostic build only):
    INSTAG_EXTRA_FLAGS_raster_blend=-DBLEND_DBG python -m instag_amd.build && python scripts/probes/blend_tile_profile.py
Every tile's workgroup records start / end (100 MHz wall clock), the CU it ran on and how far it walked; printed:
the kernel's span, the tiles that end last, the busy time per CU, and the walk-length distribution."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd import _lib
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame

L = _lib.lib()
fn = L.instag_debug_blend_timing
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
dev = torch.device("cuda")
N, size = 100000, 512
tr = build_trainer(N, dev, sh_degree=1, seed=0, densify=False)
cams = toy_cameras(size)
frames = [make_frame(cams[k % len(cams)].to(dev), synthetic_frame(size, seed=k, device=dev)) for k in range(4)]
if os.environ.get("PROBE_GRAPH", "1") == "1":
    tr.enable_graph(frames[0], warmup_steps=2)        # captured steps: the rasterizer runs with walk hints
for k in range(6):
    tr.step(frames[k % 4])
torch.cuda.synchronize()
tiles = (size // 16) ** 2
for rep in range(2):
    tr.step(frames[rep])
    torch.cuda.synchronize()
    buf = np.zeros(tiles * 8, dtype=np.uint32)
    assert fn(buf.ctypes.data, buf.size) == 0
    d = buf.reshape(tiles, 8).astype(np.int64)
    t0 = (d[:, 0] | (d[:, 1] << 32)) * 10.0 / 1e3       # us
    t1 = (d[:, 2] | (d[:, 3] << 32)) * 10.0 / 1e3
    hw, xcc, last, length = d[:, 4], d[:, 5] & 0xF, d[:, 6], d[:, 7]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cu_id = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    pop = length > 0
    print(f"--- frame {rep}: span {t1.max():.1f} us; populated tiles {int(pop.sum())}; list len mean {length[pop].mean():.0f} "
          f"max {length.max()}; tile last contributor mean {last[pop].mean():.0f} max {last.max()}")
    print("start times: min %.1f  median %.1f  max %.1f" % (t0.min(), np.median(t0), t0.max()))
    order = np.argsort(-t1)[:12]
    print("tiles ending last: (tile, x, y, start, end, last_contributor, list, cu)")
    for t in order:
        print(f"   {t:5d} ({t % 32:2d},{t // 32:2d})  {t0[t]:7.1f} {t1[t]:7.1f}  {last[t]:5d} {length[t]:5d}  cu {cu_id[t]}")
    # duration against walk length
    dur = t1 - t0
    for lo, hi in ((1, 128), (128, 256), (256, 384), (384, 512), (512, 768), (768, 1024), (1024, 4096)):
        m = pop & (last >= lo) & (last < hi)
        if m.any():
            print(f"   last contributor in [{lo:4d},{hi:4d}): {int(m.sum()):4d} tiles, duration mean {dur[m].mean():6.1f} max {dur[m].max():6.1f} us, "
                  f"ns per entry {1e3 * dur[m].sum() / np.maximum(1, last[m]).sum():.0f}")
    # per-CU: tiles, summed walk, last end
    ids = np.unique(cu_id)
    rows = []
    for c in ids:
        m = cu_id == c
        rows.append((c, int(m.sum()), int(last[m].sum()), t1[m].max()))
    rows.sort(key=lambda r: -r[3])
    print(f"{len(ids)} distinct CU ids; per CU (id, tiles, summed walk, last end) -- the 8 latest and the 8 earliest:")
    for r in rows[:8] + rows[-8:]:
        print("   ", r)
    walks = np.array([r[2] for r in rows]); ends = np.array([r[3] for r in rows])
    print(f"summed walk per CU: mean {walks.mean():.0f} max {walks.max()} ; corr(walk sum, last end) = {np.corrcoef(walks, ends)[0, 1]:.2f}")
    # alive workgroups over time
    for t in (10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110):
        print(f"   t={t:3d} us: {int(((t0 <= t) & (t1 > t)).sum()):4d} workgroups running")
