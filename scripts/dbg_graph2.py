import sys; sys.path.insert(0,'.')
import torch
from instag_amd import diff_gauss
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
dev=torch.device('cuda')
mode=sys.argv[1]
if mode=='prior':
    tr0=build_trainer(3000, dev, seed=1)
    cam=toy_cameras(128)[0].to(dev); fr=make_frame(cam, synthetic_frame(128,0,dev))
    for _ in range(3): tr0.step(fr)
    del tr0
tr=build_trainer(3000, dev, seed=1)
cam=toy_cameras(128)[0].to(dev); fr=make_frame(cam, synthetic_frame(128,0,dev))
g=tr.enable_graph(fr, warmup_steps=int(sys.argv[2]) if len(sys.argv)>2 else 2)
print('captured', flush=True)
for i in range(3): print(float(tr.step(fr)['loss']), flush=True)
print('overflow', g.check_overflow())
