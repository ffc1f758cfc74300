"""Generic GridEncoder (the reference-ABI kernels) forward/backward timing: face planes, mouth planes, 3-D hash grid."""
import sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import _lib
from instag_amd.gridencoder import GridEncoder
L = _lib.lib()
N = 100000
CFGS = {
    "face-plane": dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=16, log2_hashmap_size=17, desired_resolution=256 * 0.15),
    "mouth-plane": dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=17, desired_resolution=384 * 0.15),
    "ngp3d": dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048),
}
for name, cfg in CFGS.items():
    enc = GridEncoder(**cfg).cuda()
    D = cfg["input_dim"]
    x = ((torch.rand(N, D, device="cuda") * 2 - 1) * 0.1).requires_grad_(True)
    g = torch.randn(N, enc.output_dim, device="cuda")
    def step():
        out = enc(x, bound=0.15)
        out.backward(g)
        x.grad = None; enc.embeddings.grad = None
    for _ in range(3): step()
    torch.cuda.synchronize()
    L.instag_prof_enable(-1); L.instag_prof_reset()
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize()
    res = []
    for nm, kid in (("fwd", 7), ("bwd", 8)):
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.instag_prof_read(kid, C.byref(ms), C.byref(cnt))
        res.append(f"{nm} {1e3 * ms.value / max(1, cnt.value):7.1f} us")
    L.instag_prof_enable(0)
    print(f"grid {name:12s} T={enc.embeddings.shape[0]:8d}: " + "  ".join(res), flush=True)
