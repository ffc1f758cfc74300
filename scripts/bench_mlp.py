"""Micro-benchmark of the fused MLP kernels (forward / backward) at the C3 shapes."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd.mlp import fused_mlp

def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

import ctypes as C
from instag_amd import _lib
L = _lib.lib()

def prof(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    L.instag_prof_enable(-1); L.instag_prof_reset()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    out = {}
    for name, kid in (("fwd", 11), ("bwd", 12), ("wgrad", 13)):
        ms, cnt = C.c_double(0), C.c_int64(0)
        L.instag_prof_read(kid, C.byref(ms), C.byref(cnt))
        out[name] = (1e3 * ms.value / n, cnt.value // n)
    L.instag_prof_enable(0)
    return out

N = 100000
for (k0, h, o, nl) in [(74, 64, 11, 3), (74, 32, 11, 3), (36, 32, 32, 2), (36, 16, 6, 2), (36, 32, 6, 2)]:
    x = torch.randn(N, k0, device="cuda", requires_grad=True)
    dims = [k0] + [h] * (nl - 1) + [o]
    ws = [torch.randn(dims[i + 1], dims[i], device="cuda", requires_grad=True) * 0.1 for i in range(nl)]
    ws = [w.detach().requires_grad_(True) for w in ws]
    g = torch.randn(N, o, device="cuda")
    t_f = bench(lambda: fused_mlp(x, ws))
    def fb():
        y = fused_mlp(x, ws)
        y.backward(g)
        x.grad = None
        for w in ws: w.grad = None
    pr = prof(fb)
    print(f"{k0}->{h}x{nl-1}->{o}: fwd {t_f:7.1f} us | per step (HIP events): " +
          "  ".join(f"{k} {v[0]:6.1f} us/{v[1]} launches" for k, v in pr.items()), flush=True)
