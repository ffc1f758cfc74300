"""Diagnostic (build with INSTAG_EXTRA_FLAGS_raster_blend="-DBLEND_DBG2 ..."): replay a few captured C3 steps and print
what the forward blend's waits that gave up were waiting for (tile, segment, predecessor, who claimed it)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd import _lib, diff_gauss
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame

L = _lib.lib()
fn = L.instag_debug_blend_stalls
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
dev = torch.device("cuda")
pinned = torch.zeros(8 * 512 + 8, dtype=torch.int32).pin_memory()
seth = L.instag_debug_blend_stall_host
seth.restype, seth.argtypes = C.c_int, [C.c_void_p]
assert seth(C.c_void_p(pinned.data_ptr())) == 0      # (pinned memory is mapped at the same address on the device)
tr = build_trainer(100000, dev, sh_degree=1, seed=0, densify=False)
cams = toy_cameras(512)
frames = [make_frame(cams[k % len(cams)].to(dev), synthetic_frame(512, seed=k, device=dev)) for k in range(8)]
tr.enable_graph(frames[0], warmup_steps=2)
print("captured", flush=True)
import time
for k in range(int(os.environ.get("PROBE_STEPS", "3"))):
    t0 = time.perf_counter()
    tr.step(frames[k % 8])
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query() and time.perf_counter() - t0 < 3.0:
        time.sleep(0.001)
    late = not ev.query()
    print(f"step {k}: {1e3 * (time.perf_counter() - t0):.2f} ms" + (" -- NOT DONE after 3 s" if late else ""), flush=True)
    if late:
        for rep in range(3):
            buf = pinned.numpy().view(np.uint32).copy()
            n = int(buf[0])
            print(f"   while it hangs: {n} waits gave up", flush=True)
            for r in range(min(n, 24)):
                d = buf[8 + 8 * r: 16 + 8 * r]
                who = lambda w: (("helper" if w >> 31 else "tile's own") + f" block {(int(w) & 0x7FFFFFFF) - 1}") if w else "nobody"
                print(f"   tile {d[0]} seg {d[1]} waits for seg {d[2]} claimed by {who(d[3])}; waiter {who(d[4])}; nsegs {d[5]} "
                      f"claims {d[6]} first finished {d[7]}", flush=True)
            time.sleep(1.0)
        os._exit(3)
    buf = np.zeros(8 * 512 + 8, dtype=np.uint32)
    assert fn(buf.ctypes.data, buf.size) == 0
    n = int(buf[0])
    print(f"step {k}: {n} waits gave up so far; sort_stalls word {diff_gauss.sort_stalls()}", flush=True)
    for r in range(min(n, 12)):
        d = buf[8 + 8 * r: 16 + 8 * r]
        who = lambda w: ("helper" if w >> 31 else "tile's own") + f" block {(int(w) & 0x7FFFFFFF) - 1}" if w else "nobody"
        print(f"   tile {d[0]} seg {d[1]} waits for seg {d[2]} claimed by {who(d[3])}; waiter {who(d[4])}; nsegs {d[5]} "
              f"claims {d[6]} first finished {d[7]}", flush=True)
    if n:
        break
