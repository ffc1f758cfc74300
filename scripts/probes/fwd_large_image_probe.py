"""The segment-wise forward blend on an image with more tiles than the chip holds workgroups (C5: 300k Gaussians,
1024 x 1024 = 4,096 tiles): repeated calls of one capacity slot -- the walk hints are live from the second call on, helper
workgroups join -- must return the first call's images bit for bit, with no wait that gave up."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd import diff_gauss
from instag_amd.diff_gauss import GaussianRasterizer
from tests.helpers import hip_settings, make_scene

n, size = 300000, 1024
a, settings = make_scene(n, size, sh_degree=3, seed=0)
a["opacities"] = a["opacities"] * 0.25                  # faint: long walks, many shared tiles
g = {k: v.cuda() for k, v in a.items()}
st = hip_settings(settings)


def call():
    with torch.no_grad():
        return GaussianRasterizer(st)(means3D=g["means3D"], means2D=torch.zeros(n, 3, device="cuda"), shs=g["shs"],
                                      opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"],
                                      extra_attrs=g["extra"])


first = call()
R = diff_gauss.LAST_STATS["num_rendered"]
plan = diff_gauss.CapacityPlan([int(R * 1.1) + 64], "cuda")
diff_gauss.set_capacity_plan(plan)
try:
    for rep in range(6):
        plan.begin_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = call()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(x, y) for x, y in zip(outs, first))
        shared = int((plan.walk_hints[0][:(size // 16) ** 2] >= 24).sum())
        print(f"call {rep}: {1e3 * dt:.2f} ms, identical to the eager call: {same}, tiles marked shared for the next call: {shared}",
              flush=True)
        assert same
    print("overflow / give-ups:", plan.overflowed(), diff_gauss.sort_stalls())
finally:
    diff_gauss.set_capacity_plan(None)
