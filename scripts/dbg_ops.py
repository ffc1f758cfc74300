import sys, time; sys.path.insert(0,'.')
import torch
from torch.profiler import profile, ProfilerActivity
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
dev=torch.device('cuda')
tr=build_trainer(100000, dev)
cam=toy_cameras(512)[0].to(dev); fr=make_frame(cam, synthetic_frame(512, 0, dev))
for i in range(5): tr.step(fr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(5): tr.step(fr)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
