"""Where the sigma-net forward's time goes: fixed cost (weight staging + launch), cost per tile round, cost of the
activation stores.  Usage: python scripts/mlp_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import _lib
from instag_amd._lib import check, ptr

dev = torch.device("cuda")
L = _lib.lib()
K0, H, O, NL = 74, 64, 11, 3
w1, w2, w3 = (torch.randn(H, K0, device=dev) * 0.1, torch.randn(H, H, device=dev) * 0.1,
              torch.randn(O, H, device=dev) * 0.1)


def t_fwd(N, store, iters=60):
    x = torch.randn(N, K0, device=dev)
    y = torch.empty(N, O, device=dev)
    a1 = torch.empty(N, H, device=dev) if store else None
    a2 = torch.empty(N, H, device=dev) if store else None
    s = _lib.current_stream()
    def run():
        check(L.instag_mlp_forward(ptr(x), ptr(w1), ptr(w2), ptr(w3), ptr(y), ptr(a1), ptr(a2), N, K0, H, O, NL, s), "f")
    for _ in range(10):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def t_bwd(N, iters=60):
    dy = torch.randn(N, O, device=dev)
    a1, a2 = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
    dz1, dz2, dx = torch.empty(N, H, device=dev), torch.empty(N, H, device=dev), torch.empty(N, K0, device=dev)
    s = _lib.current_stream()
    def run():
        check(L.instag_mlp_backward_add(ptr(dy), ptr(a1), ptr(a2), ptr(w1), ptr(w2), ptr(w3), ptr(dz1), ptr(dz2),
                                        ptr(dx), None, N, K0, H, O, NL, s), "b")
    for _ in range(10):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N in (32, 32 * 1024, 65536, 100000, 131072, 196608, 262144):
    print(f"N={N:7d} tiles={(N + 31) // 32:5d}  fwd store {t_fwd(N, True):6.1f} us   fwd no-store {t_fwd(N, False):6.1f} us"
          f"   bwd {t_bwd(N):6.1f} us", flush=True)
