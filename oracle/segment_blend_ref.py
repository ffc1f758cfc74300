"""TEST INFRASTRUCTURE -- never imported by the product path.

Front-to-back compositing of ONE pixel's depth-ordered list, two ways, in float32 numpy:

* ``serial``  -- the published forward loop the reference's rasterizer runs (call site gaussian_renderer/__init__.py:111-121;
  rules restated in oracle/rasterize_ref.py::blend): skip alpha < 1/255, clamp at 0.99, stop BEFORE the entry whose
  T (1 - alpha) would fall below 1e-4, T <- T (1 - alpha), C += c alpha T.
* ``by_segments`` -- the arithmetic of instag_amd/csrc/raster_blend.hip::blend_forward_claim_kernel: the list is cut
  into segments; inside a segment the recurrence runs on the LOCAL transmittance t (1 at the segment's first entry)
  with local weights alpha_k t_(k-1) and stops at the first entry with fl(P t_k) < 1e-4 (P = transmittance in front of
  the segment); a segment hands on t_seg (0 after a stop) and the next P is fl(P t_seg) (``chain_step``); the sums in
  front of the next segment are fl(P * local + previous) (one fma on the device: a double-rounded stand-in here).

The segments can be computed in ANY order once every t_seg is known -- which is what lets several workgroups walk one
tile.  tests/test_host_logic.py checks that both forms stop at the same entry and agree to float32 rounding.
"""
import numpy as np

F = np.float32
T_MIN = F(1e-4)
ALPHA_MIN = F(1.0 / 255.0)


def _alpha(raw):
    a = np.minimum(F(0.99), F(raw))
    return a if a >= ALPHA_MIN else F(0.0)


def serial(raw_alphas, colors):
    """-> (C [ch], T_final, n_contrib = index + 1 of the last contributing entry)."""
    T, C, last = F(1.0), np.zeros(colors.shape[1], dtype=F), 0
    for k, raw in enumerate(raw_alphas):
        a = _alpha(raw)
        if a == 0:
            continue
        test_T = F(T * F(F(1.0) - a))
        if test_T < T_MIN:
            break
        C = (C + colors[k] * F(a * T)).astype(F)
        T = test_T
        last = k + 1
    return C, T, last


def chain_step(P, alive, tseg):
    nxt = F(P * tseg)
    if alive:
        if nxt < T_MIN:
            alive = False
        else:
            P = nxt
    return P, alive


def walk_segment(raw_alphas, colors, P, alive):
    """One segment with the transmittance P in front of it -> (local sums, T after it, t_seg, last contributor in it)."""
    t, loc, last, stopped = F(1.0), np.zeros(colors.shape[1], dtype=F), 0, False
    if alive:
        for k, raw in enumerate(raw_alphas):
            a = _alpha(raw)
            tt = F(t * F(F(1.0) - a))
            if F(P * tt) < T_MIN:
                stopped = True
                break
            if a != 0:
                loc = (loc + colors[k] * F(a * t)).astype(F)
                last = k + 1
            t = tt
    return loc, F(P * t), (F(0.0) if (stopped or not alive) else t), last


def transmittance_only(raw_alphas):
    """t_seg as the transmittance-only pass posts it (no stop rule: the verdict for the chain is the same)."""
    t = F(1.0)
    for raw in raw_alphas:
        t = F(t * F(F(1.0) - _alpha(raw)))
    return t


def by_segments(raw_alphas, colors, seg_len, order=None, speculative=()):
    """Segments walked in ``order`` (default: front to back); those in ``speculative`` post their t_seg from the
    transmittance-only pass first, as a workgroup does whose predecessors are not posted yet."""
    n = len(raw_alphas)
    nseg = (n + seg_len - 1) // seg_len
    sl = [slice(s * seg_len, min(n, (s + 1) * seg_len)) for s in range(nseg)]
    posted = {s: transmittance_only(raw_alphas[sl[s]]) for s in speculative}
    # every walker needs P of ITS segment: the ordered product of what the segments in front of it post.  Segments not in
    # `speculative` post after their own walk, so they must come behind their predecessors in `order`.
    results = {}
    for s in (order if order is not None else range(nseg)):
        P, alive = F(1.0), True
        for q in range(s):
            P, alive = chain_step(P, alive, posted[q])
        loc, T_after, tseg, last = walk_segment(raw_alphas[sl[s]], colors[sl[s]], P, alive)
        results[s] = (P, alive, loc, T_after, last)
        posted.setdefault(s, tseg)
    # adding the tile up, front to back
    C, Tf, last = np.zeros(colors.shape[1], dtype=F), F(1.0), 0
    P, alive = F(1.0), True
    for s in range(nseg):
        Ps, alive_s, loc, T_after, ll = results[s]
        assert Ps == P and alive_s == alive, "every reader derives the same state from the posted products"
        if alive:
            C = (C + (P * loc).astype(F)).astype(F)
            Tf = T_after
            if ll:
                last = s * seg_len + ll
        P, alive = chain_step(P, alive, posted[s])
    return C, Tf, last
