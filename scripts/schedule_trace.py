#!/usr/bin/env python
"""Where does a density-control event's time go?  Runs the C3-phase step with density control on (or
schedule="reference" with argv[1] == "reference"), synchronising after every iteration, and prints the per-iteration wall
time around every event: the eager density-control iteration, the re-capture, the first replays of the new graph.

    python scripts/schedule_trace.py [reference] [N]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import diff_gauss  # noqa: E402
from instag_amd.scene_synth import synthetic_frame, toy_cameras  # noqa: E402
from instag_amd.train import build_trainer, make_frame  # noqa: E402

schedule = "reference" if "reference" in sys.argv[1:] else None
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
N = nums[0] if nums else 100000
dev = torch.device("cuda")
cams = toy_cameras(512)
tr = build_trainer(N, dev, densify=True, schedule=schedule)
frames = [make_frame(cams[k % len(cams)].to(dev), synthetic_frame(512, k, dev)) for k in range(8)]
tr.iteration = 550
tr.enable_graph(frames[0], keep_state=True)
torch.cuda.synchronize()
rows = []
for i in range(400):
    it = tr.iteration + 1
    due = tr._densify_due(it)
    rec0, d0, r0 = tr.recaptures, getattr(tr, "density_seconds", 0.0), getattr(tr, "recapture_seconds", 0.0)
    t0 = time.perf_counter()
    if "profile" in sys.argv and it in (700, 701):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        tr.step(frames[i % 8])
        pr.disable()
        print(f"---- cProfile of iteration {it} ({'density-control (eager)' if due else 're-capture'})")
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    else:
        tr.step(frames[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((it, due, tr.recaptures - rec0, 1e3 * (t1 - t0), 1e3 * (t2 - t0),
                 1e3 * (getattr(tr, "density_seconds", 0.0) - d0), 1e3 * (getattr(tr, "recapture_seconds", 0.0) - r0),
                 tr.g.num_points))
diff_gauss.set_capacity_plan(None)
steady = sorted(r[4] for r in rows if not r[1] and not r[2])
print(f"steady replayed iteration: median {steady[len(steady) // 2]:.3f} ms (host {sorted(r[3] for r in rows)[len(rows) // 2]:.3f} ms)")
print("iteration  due recap  host_ms  total_ms  density_ms  recapture_ms  gaussians")
for k, r in enumerate(rows):
    near = any(rows[j][1] or rows[j][2] for j in range(max(0, k - 1), min(len(rows), k + 1))) or (k > 1 and (rows[k - 2][1] or rows[k - 2][2]))
    if near:
        print(f"{r[0]:9d}  {int(r[1]):3d} {r[2]:5d}  {r[3]:7.3f}  {r[4]:8.3f}  {r[5]:10.3f}  {r[6]:12.3f}  {r[7]:9d}")
