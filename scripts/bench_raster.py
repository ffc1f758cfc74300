"""Rasterizer-only forward+backward timing for the raster configs of BASELINE.json (C2, C3 shape, C5)."""
import sys, os, time, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import _lib, diff_gauss
from instag_amd.diff_gauss import GaussianRasterizer
from tests.helpers import hip_settings, make_scene
L = _lib.lib()
for name, n, size, deg in (("C2", 50000, 512, 1), ("C3-raster", 100000, 512, 1), ("C5", 300000, 1024, 3)):
    a, settings = make_scene(n, size, sh_degree=deg, seed=0)
    g = {k: v.cuda().requires_grad_(k != "extra") for k, v in a.items()}
    st = hip_settings(settings)
    w = torch.randn(3, size, size, device="cuda")
    def step():
        m2 = torch.zeros(n, 3, device="cuda", requires_grad=True)
        outs = GaussianRasterizer(st)(means3D=g["means3D"], means2D=m2, shs=g["shs"], opacities=g["opacities"],
                                      scales=g["scales"], rotations=g["rotations"], extra_attrs=g["extra"])
        ((outs[0] * w).sum() + outs[3].sum()).backward()
        for v in g.values(): v.grad = None
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K * 1e3
    L.instag_prof_enable(-1); L.instag_prof_reset()
    for _ in range(K): step()
    torch.cuda.synchronize()
    ks = {}
    for nm, kid in (("preprocess", 0), ("duplicate", 1), ("sort", 2), ("ranges", 3), ("blend_fwd", 4), ("blend_bwd", 5), ("preprocess_bwd", 6)):
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.instag_prof_read(kid, C.byref(ms), C.byref(cnt))
        ks[nm] = round(1e3 * ms.value / K, 1)
    L.instag_prof_enable(0)
    print(f"{name}: N={n} {size}x{size} SH{deg} R={diff_gauss.LAST_STATS['num_rendered']} eager fwd+bwd {wall:.3f} ms/frame; kernel us/frame {ks} sum {sum(ks.values())/1e3:.3f} ms", flush=True)
