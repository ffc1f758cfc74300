import sys, time; sys.path.insert(0,'.')
import torch
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
from instag_amd.renderer import render_motion
N=int(sys.argv[1]) if len(sys.argv)>1 else 100000
dev=torch.device('cuda')
def P(*a):
    torch.cuda.synchronize(); print(f"[{time.perf_counter()-T0:7.2f}s]", *a, flush=True)
T0=time.perf_counter()
tr=build_trainer(N, dev); P('trainer built')
cam=toy_cameras(512)[0].to(dev); fr=make_frame(cam, synthetic_frame(512, 0, dev)); P('frame')
g=tr.g
x=g.get_xyz
pm=g.neural_motion_grid(x, fr.talking_dict['auds'], fr.talking_dict['au_exp']); P('pmf fwd')
m=tr.motion_net(x+pm['p_xyz'], fr.talking_dict['auds'], fr.talking_dict['au_exp']); P('umf fwd')
pkg=render_motion(fr, g, tr.motion_net, None, tr.bg, return_attn=True, personalized=False, align=True); P('render_motion')
loss,_=tr.loss_fn(fr,pkg,True); P('loss', float(loss))
loss.backward(); P('backward')
tr.motion_optimizer.step(); P('motion opt'); g.optimizer.step(); P('gauss opt')
tr.motion_optimizer.zero_grad(); g.optimizer.zero_grad()
for i in range(3):
    t=time.perf_counter(); tr.step(fr); torch.cuda.synchronize(); print('step', i, time.perf_counter()-t, flush=True)
