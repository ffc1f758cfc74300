"""Cost of the data-parallel forms of the captured C3 step on ONE GPU: one graph vs the two-graph split that the
multi-rank path replays (graph A | [all-reduce, not issued here] | graph B: scale, hand back, statistics, optimizers)
vs the three-graph form (A' | [first bucket's all-reduce would start here] | A'' | [second bucket] | B).
    python scripts/probes/dp_split_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame

dev, size = torch.device("cuda"), 512
cams = toy_cameras(size)
frames = [make_frame(cams[i % len(cams)].to(dev), synthetic_frame(size, i, dev)) for i in range(8)]
for split in (False, True, "early", False, True, "early"):
    tr = build_trainer(100000, dev, sh_degree=1, seed=0, densify=False)
    tr.iteration = 3100
    tr.enable_graph(frames[0], split_for_allreduce=split)
    for i in range(10):
        tr.step(frames[i % 8])
    torch.cuda.synchronize()
    snap = tr.snapshot()
    tr.restore(snap)
    t0 = time.perf_counter()
    K = 60
    for i in range(K):
        tr.step(frames[i % 8])
    torch.cuda.synchronize()
    print(f"split={split}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step", flush=True)
    del tr
