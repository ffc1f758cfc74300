import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.helpers import *
from instag_amd.diff_gauss import debug_export, rasterize_forward
a, settings = make_scene(100000, 512, sh_degree=1)
s = hip_settings(settings)
g = {k: v.cuda().contiguous() for k, v in a.items()}
outs, st = rasterize_forward(s, g["means3D"], g["shs"], None, g["opacities"], g["scales"], g["rotations"], None, g["extra"])
d = debug_export(st); torch.cuda.synchronize()
nc=d['n_contrib'].float().view(32,16,32,16).permute(0,2,1,3).reshape(1024,256)
mx=nc.max(1).values.cpu().numpy(); rng=(d['ranges'][:,1]-d['ranges'][:,0]).cpu().numpy()
print('R',d['R'],'tiles with work',(rng>0).sum(),'list len: mean %.0f max %d'%(rng[rng>0].mean(), rng.max()))
print('n (max n_contrib) per tile: mean %.0f  p50 %.0f p90 %.0f p99 %.0f max %d   sum %d'%(mx[mx>0].mean(), np.percentile(mx[mx>0],50),np.percentile(mx[mx>0],90),np.percentile(mx[mx>0],99), mx.max(), mx.sum()))
print('mean n_contrib per pixel (active)', nc[nc>0].mean().item())
