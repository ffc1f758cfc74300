"""Which ingredient of the host-fed loop slows the replayed step down, and does it persist?  One process, variants in
sequence, the resident loop re-measured after each.   python scripts/probes/host_frames_probe2.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd import _lib, diff_gauss  # noqa: E402
from instag_amd.scene_synth import synthetic_frame, toy_cameras  # noqa: E402
from instag_amd.train import HostFrameFeeder, build_trainer, make_frame  # noqa: E402

dev = torch.device("cuda")
cams = toy_cameras(512)
tr = build_trainer(100000, dev)
frames = [make_frame(cams[k % len(cams)].to(dev), synthetic_frame(512, k, dev)) for k in range(8)]
host = [HostFrameFeeder.to_host(f) for f in frames]
stage = [frames[0].packed(dev), frames[0].packed(dev)]
tr.enable_graph(frames[0])
side = _lib.warmup_stream(dev)
ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()
main = torch.cuda.current_stream(dev)
small_host = torch.zeros(64, dtype=torch.uint8).pin_memory()
small_dev = torch.zeros(64, dtype=torch.uint8, device=dev)


def resident(n):
    for i in range(n):
        tr.step(frames[i % 8])


def events_only(n):
    for i in range(n):
        ev_a.record(main)
        side.wait_event(ev_a)
        ev_b.record(side)
        main.wait_event(ev_b)
        tr.step(frames[i % 8])


def copy_main(n):
    for i in range(n):
        stage[i % 2].copy_from(host[i % 8])          # H2D on the main stream, in front of the step
        tr.step(stage[i % 2])


def copy_side_nowait(n):
    for i in range(n):
        with torch.cuda.stream(side):
            stage[(i + 1) % 2].copy_from(host[(i + 1) % 8])     # H2D on the side stream, no ordering with the step at all
        tr.step(frames[i % 8])


def small_copy_side(n):
    for i in range(n):
        with torch.cuda.stream(side):
            small_dev.copy_(small_host, non_blocking=True)
        tr.step(frames[i % 8])


def kernel_side(n):
    for i in range(n):
        with torch.cuda.stream(side):
            small_dev.add_(1)
        tr.step(frames[i % 8])


resident(10)
torch.cuda.synchronize()
SNAP = tr.snapshot()


def measure(name, fn, steps=60):
    tr.restore(SNAP)            # the steps train: without this every later measurement walks longer tile lists
    fn(10)
    tr.restore(SNAP)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    print(f"{name:28s} {1e3 * (time.perf_counter() - t0) / steps:.4f} ms/step", flush=True)


def fed(n):
    feeder.run(tr.step, host, n)


feeder = HostFrameFeeder(frames[0], dev)
order = sys.argv[1:] or ["events_only", "kernel_side", "small_copy_side", "copy_main", "copy_side_nowait", "fed", "fed", "fed"]
measure("resident", resident)
for name in order:
    measure(name, globals()[name])
    measure("  resident after", resident)
diff_gauss.set_capacity_plan(None)
