"""Which tensors does autograd still sum with an elementwise launch in the C3 step?  (shapes of aten::add / add_ calls
during one eager backward)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
dev = torch.device("cuda", 0)
tr = build_trainer(20000, dev, sh_degree=1, seed=0, densify=False)
fr = make_frame(toy_cameras(256)[0].to(dev), synthetic_frame(256, seed=0, device=dev))
for _ in range(2):
    tr.step(fr)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    tr.step(fr)
torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::sum", "aten::mul", "aten::copy_", "aten::fill_", "aten::zeros", "aten::zero_"):
        print(e.name, e.input_shapes)
