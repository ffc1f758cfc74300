"""Micro-benchmark of the tri-plane encoder kernels at the C3 shape (debug variants via INSTAG_TP_VARIANT)."""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import _lib
from instag_amd.gridencoder import GridEncoder, tri_plane_encode
L = _lib.lib()
N = 100000
enc = dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=16, log2_hashmap_size=17,
           desired_resolution=256 * 0.15, gridtype="hash", align_corners=False)
es = [GridEncoder(**enc).cuda() for _ in range(3)]
xyz = ((torch.rand(N, 3, device="cuda") * 2 - 1) * 0.1).requires_grad_(True)
g = torch.randn(N, 36, device="cuda")
def step():
    out = tri_plane_encode(xyz, es[0], es[1], es[2], 0.15)
    out.backward(g)
    xyz.grad = None
    for e in es: e.embeddings.grad = None
for _ in range(3): step()
torch.cuda.synchronize()
L.instag_prof_enable(-1); L.instag_prof_reset()
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
for name, kid in (("fwd", 7), ("bwd", 8)):
    ms, cnt = C.c_double(0), C.c_int64(0)
    L.instag_prof_read(kid, C.byref(ms), C.byref(cnt))
    print(f"variant {os.environ.get('INSTAG_TP_VARIANT', '0')}: triplane {name} {1e3 * ms.value / max(1, cnt.value):7.1f} us", flush=True)
