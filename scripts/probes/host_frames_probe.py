"""Host-resident frames uploaded one step ahead (HostFrameFeeder) vs device-resident frames: step time of each, for a
timeline under rocprofv3 --kernel-trace --memory-copy-trace.   python scripts/probes/host_frames_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd import diff_gauss  # noqa: E402
from instag_amd.scene_synth import synthetic_frame, toy_cameras  # noqa: E402
from instag_amd.train import HostFrameFeeder, build_trainer, make_frame  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda")
cams = toy_cameras(512)
tr = build_trainer(100000, dev)
frames = [make_frame(cams[k % len(cams)].to(dev), synthetic_frame(512, k, dev)) for k in range(8)]
host = [HostFrameFeeder.to_host(f) for f in frames]
feeder = HostFrameFeeder(frames[0], dev)
tr.enable_graph(frames[0])


def resident(n):
    for i in range(n):
        tr.step(frames[i % 8])


def fed(n):
    feeder.run(tr.step, host, n)


for name, fn in (("resident", resident), ("host-fed", fed), ("resident", resident), ("host-fed", fed)):
    fn(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / steps:.4f} ms/step", flush=True)
diff_gauss.set_capacity_plan(None)
