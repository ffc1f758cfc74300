import sys, time; sys.path.insert(0,'.')
import torch
from instag_amd.gridencoder import GridEncoder, tri_plane_encode
FACE = dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=16, log2_hashmap_size=17, desired_resolution=256*0.15)
encs=[GridEncoder(**FACE).cuda() for _ in range(3)]
x=(torch.rand(100000,3,device='cuda')*0.2-0.1).requires_grad_(True)
w=torch.randn(100000,36,device='cuda')
for _ in range(3):
    y=tri_plane_encode(x,*encs,0.15); y.backward(w)
torch.cuda.synchronize()
y=tri_plane_encode(x,*encs,0.15)
s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): y.backward(w, retain_graph=True)
e.record(); torch.cuda.synchronize(); print('bwd us', s.elapsed_time(e)/20*1e3)
