import sys, os, torch, ctypes as C
sys.path.insert(0, "/root/repo")
from instag_amd import _lib, diff_gauss
from instag_amd.diff_gauss import GaussianRasterizer
from tests.helpers import hip_settings, make_scene
L = _lib.lib()
a, settings = make_scene(100000, 512, sh_degree=1, seed=0)
g = {k: v.cuda() for k, v in a.items()}
st = hip_settings(settings)
def step():
    with torch.no_grad():
        GaussianRasterizer(st)(means3D=g["means3D"], means2D=torch.zeros(100000,3,device="cuda"), shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"], extra_attrs=g["extra"])
for _ in range(3): step()
torch.cuda.synchronize()
L.instag_prof_enable(-1); L.instag_prof_reset()
for _ in range(20): step()
torch.cuda.synchronize()
ms, cnt = C.c_double(0), C.c_int64(0)
L.instag_prof_read(4, C.byref(ms), C.byref(cnt))
print("blend_fwd us", 1e3*ms.value/cnt.value)
