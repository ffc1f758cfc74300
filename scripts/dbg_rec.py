import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.helpers import *
from tests.test_raster_gpu import run_oracle
from instag_amd.diff_gauss import debug_export, rasterize_forward
a, settings = make_scene(2000, 128, sh_degree=1)
outs_o, aux, _ = run_oracle(a, settings)
s = hip_settings(settings)
g = {k: v.cuda().contiguous() for k, v in a.items()}
outs, st = rasterize_forward(s, g["means3D"], g["shs"], None, g["opacities"], g["scales"], g["rotations"], None, g["extra"])
d = debug_export(st); torch.cuda.synchronize()
pre=aux['pre']; vis=pre['visible']; rec=d['rec2d'].cpu()
def ulp(a,b):
    ai=a.contiguous().view(torch.int32).long(); bi=b.contiguous().view(torch.int32).long()
    return (ai-bi).abs()
for name,sl,ref in [('xy',slice(0,2),pre['xy']),('conic',slice(2,5),pre['conic']),('rgb',slice(6,9),pre['rgb']),('depth',slice(9,10),pre['depth'][:,None]),('normal',slice(10,13),pre['normal'])]:
    u=ulp(rec[vis][:,sl], ref.detach()[vis])
    print(name, 'max ulp', u.max().item(), 'frac mismatch', (u>0).float().mean().item(), 'per col', (u>0).float().mean(0).tolist())
