import sys; sys.path.insert(0,'.')
import torch
from instag_amd import diff_gauss
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
piece=sys.argv[1]
dev=torch.device('cuda')
tr=build_trainer(3000, dev, seed=1)
cam=toy_cameras(128)[0].to(dev); fr=make_frame(cam, synthetic_frame(128,0,dev))
for _ in range(2): tr.step(fr)
R=diff_gauss.LAST_STATS['num_rendered']
plan=diff_gauss.CapacityPlan([R*2,R*2],dev); diff_gauss.set_capacity_plan(plan)
g=tr.g
def body():
    plan.begin_step()
    if piece=='grid':
        return tr.motion_net.encode_x(g.get_xyz, 0.15).sum()
    if piece=='audio':
        return tr.motion_net.encode_audio(fr.talking_dict['auds']).sum()
    if piece=='umf':
        m=tr.motion_net(g.get_xyz, fr.talking_dict['auds'], fr.talking_dict['au_exp']); return m['d_xyz'].sum()
    if piece in ('raster_fwd','raster_fwdbwd'):
        from instag_amd.renderer import render
        pk=render(fr, g, None, tr.bg); return pk['render'].sum()
    if piece in ('full_fwd','full_fwdbwd','full'):
        from instag_amd.renderer import render_motion
        pk=render_motion(fr, g, tr.motion_net, None, tr.bg, return_attn=True, personalized=False, align=True)
        loss,_=tr.loss_fn(fr,pk,True); tr._pk=pk; return loss
    if piece=='loss':
        from instag_amd.losses import l1_and_ssim
        x=torch.rand(3,128,128,device=dev,requires_grad=True); a,b=l1_and_ssim(x, fr.original_image); return a+b
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        l=body()
        if piece.endswith('bwd') or piece in ('full','umf','grid','audio','loss'): l.backward()
        if piece=='full': tr._stats_and_optimizers(tr._pk, False)
        tr._zero_grad()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print('warm ok', piece, flush=True)
gr=torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    l=body()
    if piece.endswith('bwd') or piece in ('full','umf','grid','audio','loss'): l.backward()
    if piece=='full': tr._stats_and_optimizers(tr._pk, False)
    tr._zero_grad()
print('captured', piece, flush=True)
gr.replay(); torch.cuda.synchronize(); print('replayed', piece, float(l), flush=True)
