"""GPU parity: HIP rasterizer (through the C ABI) vs the CPU oracle on the same seeded scenes.

Bars (BASELINE.json north_star): tile ids / bin counts / sort order bit-exact; RGB max-abs <= 1e-4;
depth / normal / alpha / extra <= 1e-4; gradients within 2e-3 of the gradient's max magnitude
(fp32, different summation order)."""
import numpy as np
import os

import pytest
import torch

from tests.helpers import hip_settings, leaf, make_scene, oracle_settings

pytestmark = pytest.mark.gpu


def run_oracle(a, settings, use_sh=True, use_cov=False, upstream=None):
    from oracle import rasterize_ref as R
    s = oracle_settings(settings)
    inp = {k: leaf(v) for k, v in a.items()}
    m2 = torch.zeros(inp["means3D"].shape[0], 3, requires_grad=True)
    cov = None
    if use_cov:
        cov0, _, _ = R.build_cov3d(a["scales"], a["rotations"], 1.0)
        cov = leaf(cov0)
    colors = None if use_sh else leaf(torch.sigmoid(a["shs"][:, 0, :]))
    outs, aux = R.rasterize(inp["means3D"], m2, inp["shs"] if use_sh else None, colors, inp["opacities"],
                            None if use_cov else inp["scales"], None if use_cov else inp["rotations"], cov,
                            inp["extra"], s, return_aux=True)
    inp["means2D"], inp["cov3D"], inp["colors"] = m2, cov, colors
    return outs, aux, inp


def run_hip(a, settings, use_sh=True, use_cov=False):
    from instag_amd.diff_gauss import GaussianRasterizer, _RasterizeGaussians
    from oracle import rasterize_ref as R
    s = hip_settings(settings)
    inp = {k: leaf(v, "cuda") for k, v in a.items()}
    m2 = torch.zeros(inp["means3D"].shape[0], 3, requires_grad=True, device="cuda")
    cov = None
    if use_cov:
        cov0, _, _ = R.build_cov3d(a["scales"], a["rotations"], 1.0)
        cov = leaf(cov0, "cuda")
    colors = None if use_sh else leaf(torch.sigmoid(a["shs"][:, 0, :]), "cuda")
    rast = GaussianRasterizer(raster_settings=s)
    outs = rast(means3D=inp["means3D"], means2D=m2, shs=inp["shs"] if use_sh else None, colors_precomp=colors,
                opacities=inp["opacities"], scales=None if use_cov else inp["scales"],
                rotations=None if use_cov else inp["rotations"], cov3Ds_precomp=cov, extra_attrs=inp["extra"])
    inp["means2D"], inp["cov3D"], inp["colors"] = m2, cov, colors
    return outs, inp


CASES = [
    pytest.param(2000, 128, 1, id="C1-2k-128"),
    pytest.param(6000, 200, 3, id="6k-200-sh3-ragged"),
    pytest.param(20000, 256, 1, id="20k-256"),
]


@pytest.mark.parametrize("n,size,deg", CASES + [pytest.param(6000, 160, 1, id="6k-160-deep-scene")])
def test_binning_bit_exact(n, size, deg):
    from instag_amd.diff_gauss import debug_export, rasterize_forward, _f32c
    a, settings = make_scene(n, size, sh_degree=deg)
    if size == 160:
        # a scene four times as deep: view-space z on both sides of 0.5 and of 1.0, so the depth keys do NOT share their
        # top byte and the depth sort's last pass really sorts (in the other cases every key has the top byte 0x3F and
        # that pass takes its identity shortcut, csrc/raster_sort.hip)
        a["means3D"] = a["means3D"] * 4.0
        z = a["means3D"] @ settings["viewmatrix"][:3, 2] + settings["viewmatrix"][3, 2]
        assert float(z.min()) < 0.5 < 1.0 < float(z.max()), (float(z.min()), float(z.max()))
    else:
        z = a["means3D"] @ settings["viewmatrix"][:3, 2] + settings["viewmatrix"][3, 2]
        assert 0.5 <= float(z.min()) and float(z.max()) < 2.0
    outs_o, aux, _ = run_oracle(a, settings)
    s = hip_settings(settings)
    g = {k: v.cuda().contiguous() for k, v in a.items()}
    outs, st = rasterize_forward(s, g["means3D"], g["shs"], None, g["opacities"], g["scales"], g["rotations"],
                                 None, g["extra"])
    d = debug_export(st)
    torch.cuda.synchronize()
    pre, binning = aux["pre"], aux["binning"]
    assert torch.equal(d["radii"].cpu(), pre["radii"])
    assert np.array_equal(d["tiles_touched"].cpu().numpy().astype(np.int64), binning["tiles_touched"])
    assert binning["R"] < binning["candidates"]          # exact tile culling removed some rectangle tiles
    assert d["R"] == binning["R"]
    # the bounding rectangle = the PUBLISHED binning (oracle cull="rect"): min tile, width, height bit for bit, so the
    # rectangle lists the kernels cull FROM are the published ones; that culling changes no result is
    # tests/test_oracle_golden.py::test_exact_tile_culling_changes_nothing
    vis_r = pre["radii"] > 0
    rect_o = pre["rect"]                                # min x, min y, max x, max y (exclusive)
    rect_h = d["rect"].cpu()
    assert torch.equal(rect_h[vis_r, 0], rect_o[vis_r, 0]) and torch.equal(rect_h[vis_r, 1], rect_o[vis_r, 1])
    assert torch.equal(rect_h[vis_r, 2], (rect_o[:, 2] - rect_o[:, 0])[vis_r])
    assert torch.equal(rect_h[vis_r, 3], (rect_o[:, 3] - rect_o[:, 1])[vis_r])
    assert int((rect_h[vis_r, 2] * rect_h[vis_r, 3]).sum()) == binning["candidates"] == int(pre["tiles_touched"].sum())
    assert np.array_equal(d["keys"].cpu().numpy().view(np.uint64), binning["keys"])
    assert np.array_equal(d["point_list"].cpu().numpy(), binning["point_list"])
    assert np.array_equal(d["ranges"].cpu().numpy(), binning["ranges"])
    vis = pre["visible"]
    rec = d["rec2d"].cpu()
    # per-Gaussian floats produced by the contraction-free TU: bit-exact
    assert torch.equal(rec[vis, 0:2], pre["xy"].detach()[vis])
    assert torch.equal(rec[vis, 2:5], pre["conic"].detach()[vis])
    assert torch.equal(rec[vis, 9], pre["depth"].detach()[vis])
    assert torch.equal(rec[vis, 6:9], pre["rgb"].detach()[vis])
    assert torch.equal(rec[vis, 10:13], pre["normal"].detach()[vis])
    # n_contrib may differ only where exp() rounding flips a threshold
    nc = d["n_contrib"].cpu()
    assert (nc != aux["n_contrib"]).float().mean().item() < 1e-3


@pytest.mark.parametrize("n,size,deg", CASES)
def test_forward_images(n, size, deg):
    a, settings = make_scene(n, size, sh_degree=deg)
    outs_o, aux, _ = run_oracle(a, settings)
    outs, _ = run_hip(a, settings)
    names = ["image", "depth", "normal", "alpha", "radii", "extra"]
    for name, o, h in zip(names, outs_o, outs):
        if name == "radii":
            assert torch.equal(h.cpu(), o)
            continue
        err = (h.detach().cpu() - o.detach()).abs().max().item()
        assert err <= 1e-4, f"{name}: max abs err {err}"


@pytest.fixture(autouse=True, params=["tile", "segment"])
def forward_blend_kernel(request, monkeypatch):
    """Every test of this file runs with each of the two forward blend kernels forced (csrc/raster_blend.hip: by default
    the instance count picks one)."""
    monkeypatch.setenv("INSTAG_BLEND_FWD", request.param)
    return request.param


PARITY_LOG = {}


def _record(case, name, **numbers):
    """Measured parity errors of this test session -> gpurun_out/parity_errors.json (copied to profiles/ per round)."""
    import json
    import os
    PARITY_LOG.setdefault(case, {})[name] = {k: float(v) for k, v in numbers.items()}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_errors.json"), "w") as f:
        json.dump(PARITY_LOG, f, indent=1, sort_keys=True)


# Gradient bars, against the oracle evaluated in DOUBLE precision on the fp32 binning (oracle precision="fp64"):
#   (1) max |dg| <= 5e-5 * max |g| per tensor                       (measured: <= 2.8e-5, profiles/r02_parity_errors.json);
#   (2) entries with |g| > 1e-3 * max |g|: relative error <= 1e-3 at the 99.9th percentile (measured <= 6.2e-4) and
#       <= 5e-3 at the worst entry (measured <= 2.7e-3): __expf / v_rcp_f32 in the blend recurrence and fp32 summation
#       over lists of up to ~2,600 entries are what is left once the oracle's own rounding is out of the comparison.
G_MAX, G_REL_Q999, G_REL_WORST = 5e-5, 1e-3, 5e-3


def _grad_check(go, gh, name, case="", max_tol=G_MAX):
    go, gh = go.detach().double(), gh.detach().cpu().double()
    scale = go.abs().max().item()
    d = (go - gh).abs()
    err = d.max().item()
    big = go.abs() > 1e-3 * scale
    rel = (d[big] / go.abs()[big]) if bool(big.any()) else torch.zeros(1, dtype=torch.float64)
    q999 = torch.quantile(rel[:2_000_000], 0.999).item()
    _record(case, name, max_abs_over_max=err / max(scale, 1e-300), rel_q999=q999, rel_worst=rel.max().item(),
            grad_max=scale, entries_checked_rel=int(big.sum()))
    assert err <= max_tol * scale + 1e-9, f"{name}: max err {err} vs scale {scale} ({err / scale:.2e})"
    assert q999 <= G_REL_Q999, f"{name}: 99.9th percentile of the relative error {q999:.2e}"
    assert rel.max().item() <= G_REL_WORST, f"{name}: worst relative error {rel.max().item():.2e}"


GRAD_CASES = [
    pytest.param(3000, 128, True, False, 3, id="3k-128-sh2-scale_rot"),
    pytest.param(3000, 128, False, False, 3, id="3k-128-colors_precomp"),
    pytest.param(3000, 128, True, True, 3, id="3k-128-cov3D_precomp"),
    pytest.param(6000, 200, True, False, 0, id="6k-200-sh3-ragged"),
    pytest.param(20000, 256, True, False, 0, id="20k-256-sh1"),
]


@pytest.mark.parametrize("n,size,use_sh,use_cov,seed", GRAD_CASES)
def test_backward_gradients(n, size, use_sh, use_cov, seed, request):
    """All-output upstream gradient (image, depth, normal, alpha, extra weighted by unit normals) -> every input
    gradient of the HIP backward against the fp64 oracle; forward images against it too (<= 1e-4)."""
    deg = {3000: 2 if use_sh else 0, 6000: 3, 20000: 1}[n]
    a, settings = make_scene(n, size, sh_degree=deg, seed=seed)
    from oracle import rasterize_ref as R
    s = oracle_settings(settings)
    inp_o = {k: leaf(v) for k, v in a.items()}
    m2 = torch.zeros(n, 3, requires_grad=True)
    cov = None
    if use_cov:
        cov = leaf(R.build_cov3d(a["scales"], a["rotations"], 1.0)[0])
    colors = None if use_sh else leaf(torch.sigmoid(a["shs"][:, 0, :]))
    outs_o, aux = R.rasterize(inp_o["means3D"], m2, inp_o["shs"] if use_sh else None, colors, inp_o["opacities"],
                              None if use_cov else inp_o["scales"], None if use_cov else inp_o["rotations"], cov,
                              inp_o["extra"], s, return_aux=True, precision="fp64")
    inp_o["means2D"], inp_o["cov3D"], inp_o["colors"] = m2, cov, colors
    outs_h, inp_h = run_hip(a, settings, use_sh, use_cov)
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(o.shape, generator=g) if o.is_floating_point() else None for o in outs_o]
    loss_o = sum((o * w.double()).sum() for o, w in zip(outs_o, ws) if w is not None)
    loss_h = sum((o * w.cuda()).sum() for o, w in zip(outs_h, ws) if w is not None)
    loss_o.backward()
    loss_h.backward()
    case = request.node.callspec.id
    for name, o, h in zip(("image", "depth", "normal", "alpha", "radii", "extra"), outs_o, outs_h):
        if o.is_floating_point():
            err = (h.detach().cpu().double() - o.detach()).abs().max().item()
            _record(case, "fwd_" + name, max_abs=err)
            assert err <= 1e-5, f"{name}: max abs err {err}"       # north_star's bar is 1e-4; measured <= 3.5e-6
    keys = ["means3D", "means2D", "opacities", "extra"]
    keys += ["shs"] if use_sh else ["colors"]
    keys += ["cov3D"] if use_cov else ["scales", "rotations"]
    for k in keys:
        assert inp_h[k].grad is not None, k
        _grad_check(inp_o[k].grad, inp_h[k].grad, k, case)


def test_backward_deterministic():
    a, settings = make_scene(4000, 160, sh_degree=1, seed=7)
    grads = []
    for _ in range(2):
        outs_h, inp_h = run_hip(a, settings)
        (outs_h[0].sum() + outs_h[1].sum() + outs_h[2].sum() + outs_h[3].sum()).backward()
        grads.append({k: v.grad.clone() for k, v in inp_h.items() if v is not None and v.grad is not None})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def test_edge_cases():
    from instag_amd.diff_gauss import GaussianRasterizer
    a, settings = make_scene(500, 100, sh_degree=0)
    s = hip_settings(settings)
    # everything behind the camera -> background only, zero radii, finite zero gradients
    far = {k: v.clone() for k, v in a.items()}
    far["means3D"] = far["means3D"] + torch.tensor([0.0, 0.0, 5.0])
    inp = {k: leaf(v, "cuda") for k, v in far.items()}
    m2 = torch.zeros(500, 3, device="cuda", requires_grad=True)
    img, depth, normal, alpha, radii, extra = GaussianRasterizer(s)(
        means3D=inp["means3D"], means2D=m2, shs=inp["shs"], opacities=inp["opacities"], scales=inp["scales"],
        rotations=inp["rotations"], extra_attrs=inp["extra"])
    assert int(radii.sum()) == 0
    assert torch.allclose(img, s.bg[:, None, None].expand_as(img))
    assert float(alpha.detach().abs().max()) == 0.0
    img.sum().backward()
    assert float(inp["means3D"].grad.abs().max()) == 0.0
    # argument validation mirrors the published rasterizer
    with pytest.raises(Exception):
        GaussianRasterizer(s)(means3D=inp["means3D"], means2D=m2, opacities=inp["opacities"],
                              scales=inp["scales"], rotations=inp["rotations"])
    with pytest.raises(Exception):
        GaussianRasterizer(s)(means3D=inp["means3D"], means2D=m2, shs=inp["shs"], opacities=inp["opacities"])
    with pytest.raises(RuntimeError):
        GaussianRasterizer(s)(means3D=inp["means3D"].cpu(), means2D=m2, shs=inp["shs"], opacities=inp["opacities"],
                              scales=inp["scales"], rotations=inp["rotations"])


def test_compositing_identity_full_size():
    """Size-independent property at config-2 scale: render = sum_i w_i c_i + (1 - alpha) * bg, so rendering
    with two backgrounds differs by exactly (1-alpha)*(bg1-bg0); extra(=1) equals alpha."""
    from instag_amd.diff_gauss import GaussianRasterizer
    a, settings = make_scene(50000, 512, sh_degree=1)
    g = {k: v.cuda() for k, v in a.items()}
    m2 = torch.zeros(50000, 3, device="cuda")
    outs = []
    for bg in ((0.0, 1.0, 0.0), (1.0, 0.0, 1.0)):
        st = dict(settings)
        st["bg"] = torch.tensor(bg)
        outs.append(GaussianRasterizer(hip_settings(st))(
            means3D=g["means3D"], means2D=m2, shs=g["shs"], opacities=g["opacities"], scales=g["scales"],
            rotations=g["rotations"], extra_attrs=g["extra"]))
    (i0, d0, n0, a0, r0, e0), (i1, d1, n1, a1, r1, e1) = outs
    assert torch.equal(a0, a1) and torch.equal(d0, d1) and torch.equal(r0, r1)
    dbg = torch.tensor([1.0, -1.0, 1.0], device="cuda")[:, None, None]
    assert float(((i1 - i0) - (1 - a0) * dbg).abs().max()) <= 2e-6
    assert float((e0 - a0).abs().max()) <= 2e-6
    assert float(a0.min()) >= 0.0 and float(a0.max()) <= 1.0


def test_capacity_mode_matches_sync_mode():
    """Sync-free (hipGraph-capturable) forward/backward == the two-stage path; overflow is flagged, never faults."""
    from instag_amd import diff_gauss
    a, settings = make_scene(5000, 192, sh_degree=1, seed=9)
    outs_ref, inp_ref = run_hip(a, settings)
    (outs_ref[0].sum() + outs_ref[1].sum() + outs_ref[3].sum()).backward()
    R = diff_gauss.LAST_STATS["num_rendered"]
    for cap, expect_overflow in ((int(R * 1.5) + 100, False), (R, False), (R // 2, True)):
        plan = diff_gauss.CapacityPlan([cap], "cuda")
        diff_gauss.set_capacity_plan(plan)
        try:
            plan.begin_step()
            outs, inp = run_hip(a, settings)
            (outs[0].sum() + outs[1].sum() + outs[3].sum()).backward()
            torch.cuda.synchronize()
        finally:
            diff_gauss.set_capacity_plan(None)
        assert plan.needed() == [R]
        assert bool(plan.overflowed()) == expect_overflow
        if not expect_overflow:
            for o, r in zip(outs, outs_ref):
                assert torch.equal(o, r)
            for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
                assert torch.equal(inp[k].grad, inp_ref[k].grad), k
        else:
            assert all(torch.isfinite(v.grad).all() for v in inp.values() if v is not None and v.grad is not None)


def test_graphed_train_step_matches_eager(monkeypatch):
    """The hipGraph-replayed train step performs the same arithmetic as the eager step."""
    from instag_amd import diff_gauss
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import build_trainer, make_frame
    dev = torch.device("cuda")
    cams = toy_cameras(128)
    frames = [make_frame(cams[i].to(dev), synthetic_frame(128, i, dev)) for i in range(3)]
    losses = {}
    params = {}
    for mode in ("eager", "graph", "graph-two-optimizer-launches", "graph-split", "graph-early"):
        tr = build_trainer(3000, dev, seed=1)
        # "graph-two-optimizer-launches": one graph, statistics + the optimizer step of SH / opacity / scale / rotation
        # on a side stream beside the motion fields' backward, positions + networks at the end (off by default)
        monkeypatch.setenv("INSTAG_EARLY_OPTIMIZER", "1" if mode == "graph-two-optimizer-launches" else "0")
        try:
            if mode != "eager":
                # 2 eager + 2 capacity-mode steps on frame 0; "graph-split" = the two-graph form used with several
                # ranks (forward+backward | gradient exchange | statistics+optimizers); "graph-early" = the
                # three-graph form whose first bucket (SH, opacity, scale, rotation gradients, final before the motion
                # fields' backward runs) is exchanged beside that backward
                tr.enable_graph(frames[0], warmup_steps=2,
                                split_for_allreduce={"graph-split": True, "graph-early": "early"}.get(mode, False))
                assert tr._graph.early_optimizer == (mode == "graph-two-optimizer-launches")
                if mode == "graph-early":
                    g = tr._graph
                    assert g.graph_a2 is not None and g.graph_b is not None
                    names = {k for k, q in tr.g._p.items() if any(q is e for e in g._params_early)}
                    assert names == {"f_dc", "f_rest", "opacity", "scaling", "rotation"}, names
                    assert any(q is tr.g._p["xyz"] for q in g._params)
                    assert g._bucket_early.numel() == 20 * tr.g.num_points
            else:
                for _ in range(4):
                    tr.step(frames[0])
            ls = []
            for i in range(6):
                ls.append(float(tr.step(frames[i % 3])["loss"]))
            if mode != "eager":
                assert tr._graph is not None and not tr._graph.check_overflow()
        finally:
            diff_gauss.set_capacity_plan(None)
        losses[mode] = ls
        params[mode] = tr.g.get_xyz.detach().clone()
    assert tr.iteration == 10
    for mode in ("graph", "graph-two-optimizer-launches", "graph-split", "graph-early"):
        for a_, b_ in zip(losses["eager"], losses[mode]):
            assert abs(a_ - b_) <= 1e-5 * max(1.0, abs(a_)), (mode, losses["eager"], losses[mode])
        assert float((params["eager"] - params[mode]).abs().max()) <= 1e-5


@pytest.mark.parametrize("full,fuse", [(False, True), (False, False), (True, None)],
                         ids=["rgb-main-pass-fused", "rgb-main-pass-side-by-side", "all-channel-main-pass"])
def test_aux_colors_match_second_pass(full, fuse, monkeypatch):
    """aux_colors rides along the main pass: image and gradients equal a second rasterizer call on detached geometry
    (gaussian_renderer/__init__.py:253-268).  With an rgb-only main pass the backward of both images is ONE blend
    launch; with depth / normal gradients the aux image gets its own launch inside the same C call."""
    import torch
    from instag_amd import diff_gauss
    from instag_amd.diff_gauss import GaussianRasterizer
    monkeypatch.setattr(diff_gauss, "FUSE_AUX_BACKWARD", fuse)
    N, size = 3000, 96
    a, sd = make_scene(N, size, sh_degree=1, seed=7)
    a = {k: a[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    settings = hip_settings(sd)
    g = torch.Generator().manual_seed(2)
    aux0 = torch.rand(N, 3, generator=g)
    w_img, w_aux = torch.randn(3, size, size, generator=g).cuda(), torch.randn(3, size, size, generator=g).cuda()

    def leaves():
        d = {k: v.cuda().clone().requires_grad_(True) for k, v in a.items()}
        d["aux"] = aux0.cuda().clone().requires_grad_(True)
        d["m2d"] = torch.zeros(N, 3, device="cuda", requires_grad=True)
        return d

    rast = GaussianRasterizer(settings)
    ones = torch.ones(N, 1, device="cuda")
    # reference: two calls
    r = leaves()
    out_main = rast(means3D=r["means3D"], means2D=r["m2d"], shs=r["shs"], opacities=r["opacities"],
                    scales=r["scales"], rotations=r["rotations"], extra_attrs=ones)
    out_attn = rast(means3D=r["means3D"].detach(), means2D=r["m2d"], colors_precomp=r["aux"],
                    opacities=r["opacities"].detach(), scales=r["scales"].detach(), rotations=r["rotations"].detach(),
                    extra_attrs=ones)
    extra_loss = (lambda o: (o[1] * w_img[:1]).sum() + (o[2] * w_aux).sum()) if full else (lambda o: 0.0)
    ((out_main[0] * w_img).sum() + out_main[3].sum() + (out_attn[0] * w_aux).sum() + extra_loss(out_main)).backward()
    # fused: one call
    f = leaves()
    outs = rast(means3D=f["means3D"], means2D=f["m2d"], shs=f["shs"], opacities=f["opacities"], scales=f["scales"],
                rotations=f["rotations"], extra_attrs=ones, aux_colors=f["aux"])
    assert len(outs) == 7
    ((outs[0] * w_img).sum() + outs[3].sum() + (outs[6] * w_aux).sum() + extra_loss(outs)).backward()
    assert torch.equal(outs[0], out_main[0])
    assert float((outs[6] - out_attn[0]).abs().max()) <= 1e-6
    for k in r:
        gr, gf = r[k].grad, f[k].grad
        assert gr is not None and gf is not None, k
        scale = max(float(gr.abs().max()), 1e-6)
        assert float((gr - gf).abs().max()) <= 2e-5 * scale, (k, float((gr - gf).abs().max()), scale)


def test_deferred_aux_join_keeps_means2d_gradient():
    """Inside deferred_grads() the aux image's backward outlives the rasterizer's backward; all gradients unchanged."""
    from instag_amd.deferred import deferred_grads
    from instag_amd.diff_gauss import GaussianRasterizer
    N, size = 2000, 64
    a, sd = make_scene(N, size, sh_degree=1, seed=3)
    settings = hip_settings(sd)
    g = torch.Generator().manual_seed(4)
    aux0 = torch.rand(N, 3, generator=g).cuda()
    w_img, w_aux = torch.randn(3, size, size, generator=g).cuda(), torch.randn(3, size, size, generator=g).cuda()
    ones = torch.ones(N, 1, device="cuda")

    def run(deferred):
        d = {k: a[k].cuda().clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        d["aux"] = aux0.clone().requires_grad_(True)
        d["m2d"] = torch.zeros(N, 3, device="cuda", requires_grad=True)
        outs = GaussianRasterizer(settings)(means3D=d["means3D"], means2D=d["m2d"], shs=d["shs"],
                                            opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"],
                                            extra_attrs=ones, aux_colors=d["aux"])
        loss = (outs[0] * w_img).sum() + (outs[6] * w_aux).sum()
        if deferred:
            with deferred_grads("cuda"):
                loss.backward()
        else:
            loss.backward()
        torch.cuda.synchronize()
        return {k: v.grad.clone() for k, v in d.items()}

    ref, got = run(False), run(True)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k


FULL = [
    pytest.param(50000, 512, 1, id="C2-50k-512-sh1"),
    pytest.param(100000, 512, 1, id="C3-100k-512-sh1"),
    pytest.param(300000, 1024, 3, id="C5-300k-1024-sh3"),
]


@pytest.mark.parametrize("n,size,deg", FULL)
def test_full_size_properties(n, size, deg):
    """BASELINE.json's full-size configurations through size-independent properties (no oracle at these sizes):
    sorted (tile, depth) keys and consistent tile ranges; compositing identity under a background change;
    extra(=1) == alpha; bitwise reproducible backward; gradients linear in the upstream image gradient;
    sync-free capacity mode == two-stage mode."""
    from instag_amd import diff_gauss
    from instag_amd.diff_gauss import GaussianRasterizer, rasterize_forward
    a, settings = make_scene(n, size, sh_degree=deg, seed=2)
    g = {k: v.cuda() for k, v in a.items()}
    gen = torch.Generator().manual_seed(7)
    w1, w2 = torch.randn(3, size, size, generator=gen).cuda(), torch.randn(3, size, size, generator=gen).cuda()

    def run(bg, w):
        st = dict(settings)
        st["bg"] = torch.tensor(bg)
        leaves = {k: g[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        m2 = torch.zeros(n, 3, device="cuda", requires_grad=True)
        outs = GaussianRasterizer(hip_settings(st))(
            means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], opacities=leaves["opacities"],
            scales=leaves["scales"], rotations=leaves["rotations"], extra_attrs=g["extra"])
        (outs[0] * w).sum().backward()
        leaves["means2D"] = m2
        return outs, {k: v.grad for k, v in leaves.items()}

    o0, g1 = run((0.0, 1.0, 0.0), w1)
    R = diff_gauss.LAST_STATS["num_rendered"]
    assert R > n                                                       # the scene is not degenerate
    o1, _ = run((1.0, 0.0, 1.0), w1)
    (i0, d0, n0, a0, r0, e0), (i1, d1, n1, a1, r1, e1) = o0, o1
    assert torch.equal(a0, a1) and torch.equal(d0, d1) and torch.equal(r0, r1) and torch.equal(n0, n1)
    dbg = torch.tensor([1.0, -1.0, 1.0], device="cuda")[:, None, None]
    assert float(((i1 - i0) - (1 - a0) * dbg).abs().max()) <= 2e-6
    assert float((e0 - a0).abs().max()) <= 2e-6
    assert float(a0.min()) >= 0.0 and float(a0.max()) <= 1.0 and bool(torch.isfinite(i0).all())

    # reproducible, and linear in the upstream gradient
    _, g1b = run((0.0, 1.0, 0.0), w1)
    for k in g1:
        assert torch.equal(g1[k], g1b[k]), k
    _, g2 = run((0.0, 1.0, 0.0), w2)
    _, g12 = run((0.0, 1.0, 0.0), w1 + w2)
    for k in g1:
        ref = g1[k] + g2[k]
        scale = max(float(ref.abs().max()), 1e-12)
        assert float((g12[k] - ref).abs().max()) <= 2e-4 * scale, k
        assert bool(torch.isfinite(g12[k]).all()), k

    # binning state: keys sorted by (tile, depth), ranges partition the list
    st = hip_settings(settings)
    with torch.no_grad():
        _, state = rasterize_forward(st, g["means3D"], g["shs"], None, g["opacities"], g["scales"], g["rotations"],
                                     None, g["extra"])
    dbgx = diff_gauss.debug_export(state)
    keys = dbgx["keys"]
    assert dbgx["R"] == R and bool((keys[1:] >= keys[:-1]).all())
    rng = dbgx["ranges"].long()
    nonempty = rng[:, 1] > rng[:, 0]
    assert int((rng[nonempty, 1] - rng[nonempty, 0]).sum()) == R
    tiles_of_keys = (keys >> 32)
    starts = rng[nonempty, 0]
    assert bool((tiles_of_keys[starts] == torch.nonzero(nonempty).flatten()).all())

    # capacity mode
    plan = diff_gauss.CapacityPlan([int(R * 1.2) + 64], "cuda")
    diff_gauss.set_capacity_plan(plan)
    try:
        plan.begin_step()
        oc, gc = run((0.0, 1.0, 0.0), w1)
        torch.cuda.synchronize()
    finally:
        diff_gauss.set_capacity_plan(None)
    assert plan.needed() == [R] and not plan.overflowed()
    for x, y in zip(oc, o0):
        assert torch.equal(x, y)
    for k in g1:
        assert torch.equal(gc[k], g1[k]), k


@pytest.mark.parametrize("deg", [0, 1, 3])
def test_split_sh_storage_matches_concatenated(deg):
    """shs=(dc, rest) == shs=cat(dc, rest): bit-identical images and gradients (no concatenation on the device)."""
    from instag_amd.diff_gauss import GaussianRasterizer
    N, size = 3000, 96
    a, sd = make_scene(N, size, sh_degree=deg, seed=5)
    st = hip_settings(sd)
    g = torch.Generator().manual_seed(1)
    w = torch.randn(3, size, size, generator=g).cuda()
    M = a["shs"].shape[1]

    def run(split):
        d = {k: a[k].cuda().clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
        dc = a["shs"][:, :1].cuda().clone().requires_grad_(True)
        rest = a["shs"][:, 1:].cuda().clone().requires_grad_(True)
        m2 = torch.zeros(N, 3, device="cuda", requires_grad=True)
        shs = (dc, rest) if split else torch.cat([dc, rest], dim=1)
        outs = GaussianRasterizer(st)(means3D=d["means3D"], means2D=m2, shs=shs, opacities=d["opacities"],
                                      scales=d["scales"], rotations=d["rotations"],
                                      extra_attrs=torch.ones(N, 1, device="cuda"))
        ((outs[0] * w).sum() + outs[3].sum()).backward()
        grads = {k: v.grad for k, v in d.items()}
        grads.update(dc=dc.grad, rest=rest.grad if M > 1 else None, m2=m2.grad)
        return outs, grads

    o_c, g_c = run(False)
    o_s, g_s = run(True)
    for x, y in zip(o_s, o_c):
        assert torch.equal(x, y)
    for k in g_c:
        if g_c[k] is None:
            continue
        assert g_s[k] is not None and torch.equal(g_s[k], g_c[k]), k


def test_stalled_look_back_is_reported_not_silent():
    """A binning look-back whose predecessor never publishes gives up after a bounded spin -- and says so: the sticky
    device counter makes the next synchronising rasterizer call raise (and CapacityPlan.overflowed() in capacity
    mode), instead of a silently mis-sorted tile list (csrc/raster_sort.hip g_sort_stalls)."""
    from instag_amd import _lib, diff_gauss
    from instag_amd.diff_gauss import GaussianRasterizer
    L = _lib.lib()
    dev = "cuda"
    assert diff_gauss.sort_stalls(clear=True) == 0
    a, settings = make_scene(500, 64)
    inp = {k: v.to(dev) for k, v in a.items()}

    def render():
        m2 = torch.zeros(500, 3, device=dev)
        return GaussianRasterizer(hip_settings(settings, dev))(
            means3D=inp["means3D"], means2D=m2, shs=inp["shs"], opacities=inp["opacities"], scales=inp["scales"],
            rotations=inp["rotations"], extra_attrs=inp["extra"])
    render()                                                     # healthy: no stall
    assert diff_gauss.sort_stalls() == 0
    try:
        _lib.check(L.instag_debug_scan_stall_probe(_lib.current_stream()), "probe")   # block 1 waits for a block 0 that never runs
        assert diff_gauss.sort_stalls() == 1
        with pytest.raises(RuntimeError, match="look-back gave up"):
            render()
        plan = diff_gauss.CapacityPlan([4096], torch.device(dev))
        with pytest.raises(RuntimeError, match="gave up"):
            plan.overflowed()
    finally:
        assert diff_gauss.sort_stalls(clear=True) >= 1
    assert diff_gauss.sort_stalls() == 0
    render()


def test_dp_two_ranks_share_one_gpu():
    """Data-parallel train step with two ranks on this one card (gloo): replicas stay bit-identical, and the
    two-graph form used with several ranks agrees with the eager form (tests/dp_worker.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29531", os.path.join(root, "tests", "dp_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=root)
    if r.returncode != 0:
        sys.stderr.write("---- dp_worker stdout ----\n" + r.stdout[-4000:] + "\n---- dp_worker stderr ----\n" + r.stderr[-12000:])
    assert r.returncode == 0 and "DP-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_full_c3_train_step_100k_512():
    """Config C3 at FULL size through the whole train step (not only its raster pass): 100k Gaussians, 512x512, SH 1,
    PMF + UMF, image + attention map, loss block, backward, statistics, Adam -- eager launches against the replayed
    hipGraph over three frames: same losses, same parameters, no capacity overflow, finite everywhere; and the
    gradients of a repeated frame are bitwise reproducible."""
    from instag_amd import diff_gauss
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import C3_PHASE, build_trainer, make_frame
    dev = torch.device("cuda")
    N, size = 100000, 512
    cams = toy_cameras(size)
    frames = [make_frame(cams[i].to(dev), synthetic_frame(size, i, dev)) for i in range(3)]
    losses, params = {}, {}
    for mode in ("eager", "graph"):
        tr = build_trainer(N, dev, seed=0)
        try:
            if mode == "graph":
                g = tr.enable_graph(frames[0], warmup_steps=2)        # 2 eager + 2 capacity-mode steps on frame 0
                assert g.capacity >= diff_gauss.LAST_STATS["num_rendered"] > N
            else:
                for _ in range(4):
                    tr.step(frames[0])
            ls = [float(tr.step(frames[i % 3])["loss"]) for i in range(6)]
            if mode == "graph":
                assert tr._graph is not None and tr._graph.check_overflow() == []
                need = tr._graph.plan.needed()
                # (two planned slots; the attention map rides along the main pass, so one rasterizer call per step)
                assert need[0] > N and all(r <= g.capacity for r in need), need
        finally:
            diff_gauss.set_capacity_plan(None)
        assert tr.iteration == 10 and all(np.isfinite(ls)), ls
        losses[mode] = ls
        params[mode] = torch.cat([p.detach().reshape(-1) for p in tr._all_params()])
        assert bool(torch.isfinite(params[mode]).all())
        if mode == "eager":
            # bitwise reproducible gradients: the same frame twice, no optimizer step in between
            grads = []
            for _ in range(2):
                tr._forward_backward(frames[1], C3_PHASE)
                grads.append([None if p.grad is None else p.grad.clone() for p in tr._all_params()])
                tr._zero_grad()
            some = 0
            for a_, b_ in zip(*grads):
                assert (a_ is None) == (b_ is None)
                if a_ is not None:
                    assert torch.equal(a_, b_)
                    some += int(a_.abs().max() > 0)
            assert some >= 10
        del tr
    for a_, b_ in zip(losses["eager"], losses["graph"]):
        assert abs(a_ - b_) <= 1e-5 * max(1.0, abs(a_)), (losses["eager"], losses["graph"])
    assert float((params["eager"] - params["graph"]).abs().max()) <= 1e-5


@pytest.mark.parametrize("shared", [False, True], ids=["own-workgroup", "shared-tiles"])
def test_long_walks_cross_many_segments(shared, monkeypatch):
    """(shared-tiles: every tile of three or more segments is walked by four workgroups claiming segments, the form
    the forward blend uses for tiles that walked far in the previous call of a captured step.)
    Dense, faint scene on a small image: the tiles' lists are thousands of entries long and rays walk far beyond one
    128-entry segment, so the backward pass starts most of its workgroups from the per-segment state the forward pass
    left behind (csrc/raster_blend.hip: seg_state) -- every gradient against the fp64 oracle, with the same bars as
    test_backward_gradients."""
    from instag_amd import diff_gauss
    from oracle import rasterize_ref as R
    monkeypatch.setenv("INSTAG_BLEND_FWD_SHARE_ALL", "1" if shared else "0")
    n, size = 12000, 96
    a, settings = make_scene(n, size, sh_degree=1, seed=11)
    a["opacities"] = a["opacities"] * 0.12            # faint: a ray needs hundreds of Gaussians to saturate
    s = oracle_settings(settings)
    inp_o = {k: leaf(v) for k, v in a.items()}
    m2 = torch.zeros(n, 3, requires_grad=True)
    outs_o = R.rasterize(inp_o["means3D"], m2, inp_o["shs"], None, inp_o["opacities"], inp_o["scales"],
                         inp_o["rotations"], None, inp_o["extra"], s, precision="fp64")
    inp_o["means2D"] = m2
    diff_gauss.KEEP_LAST_STATE = True
    try:
        outs_h, inp_h = run_hip(a, settings)
        walked = diff_gauss.debug_export(diff_gauss.LAST_STATS.pop("state"))["n_contrib"]
    finally:
        diff_gauss.KEEP_LAST_STATE = False
    assert int(walked.max()) > 4 * 128, f"scene too shallow for this test: longest walk {int(walked.max())}"
    g = torch.Generator().manual_seed(2)
    ws = [torch.randn(o.shape, generator=g) if o.is_floating_point() else None for o in outs_o]
    sum((o * w.double()).sum() for o, w in zip(outs_o, ws) if w is not None).backward()
    sum((o * w.cuda()).sum() for o, w in zip(outs_h, ws) if w is not None).backward()
    for name, o, h in zip(("image", "depth", "normal", "alpha", "radii", "extra"), outs_o, outs_h):
        if o.is_floating_point():
            assert (h.detach().cpu().double() - o.detach()).abs().max().item() <= 1e-5, name
    for k in ("means3D", "means2D", "opacities", "extra", "shs", "scales", "rotations"):
        _grad_check(inp_o[k].grad, inp_h[k].grad, k, "long-walks")


def test_shared_tiles_give_the_same_bits(monkeypatch):
    """A tile walked by its own workgroup and the same tile shared between four workgroups (segments claimed in order,
    transmittance-only passes where a predecessor is not posted yet, one workgroup adding the segments up): the images,
    the per-pixel state and every gradient bit for bit -- the arithmetic is per segment and does not depend on who
    computed what (csrc/raster_blend.hip: blend_forward_claim_kernel)."""
    from instag_amd import diff_gauss
    results = []
    for shared in ("0", "1", "1"):
        monkeypatch.setenv("INSTAG_BLEND_FWD_SHARE_ALL", shared)
        a, settings = make_scene(12000, 96, sh_degree=1, seed=11)
        a["opacities"] = a["opacities"] * 0.12
        diff_gauss.KEEP_LAST_STATE = True
        try:
            outs, inp = run_hip(a, settings)
            state = diff_gauss.debug_export(diff_gauss.LAST_STATS.pop("state"))
        finally:
            diff_gauss.KEEP_LAST_STATE = False
        g = torch.Generator().manual_seed(2)
        sum((o * torch.randn(o.shape, generator=g).cuda()).sum() for o in outs if o.is_floating_point()).backward()
        results.append(([o.detach().clone() for o in outs], state["n_contrib"].clone(),
                        {k: v.grad.clone() for k, v in inp.items() if v is not None and v.grad is not None}))
    assert diff_gauss.sort_stalls() == 0
    for other in results[1:]:
        for x, y in zip(results[0][0], other[0]):
            assert torch.equal(x, y)
        assert torch.equal(results[0][1], other[1])
        assert results[0][2].keys() == other[2].keys()
        for k in results[0][2]:
            assert torch.equal(results[0][2][k], other[2][k]), k


def test_walk_hints_of_a_capacity_slot_change_nothing():
    """Repeated calls of one capacity slot: from the second call on the slot's walk hints are live and helper workgroups
    join the tiles that walked far (csrc/raster_blend.hip; more tiles -- 36 x 36 -- than the chip holds workgroups of this
    kernel, so some of the tiles' own workgroups start late).  Every call returns the eager call's images bit for bit and
    no wait for a post gives up."""
    from instag_amd import diff_gauss
    from instag_amd.diff_gauss import GaussianRasterizer
    n, size = 40000, 576
    a, settings = make_scene(n, size, sh_degree=1, seed=21)
    a["opacities"] = a["opacities"] * 0.15
    g = {k: v.cuda() for k, v in a.items()}
    st = hip_settings(settings)

    def call():
        with torch.no_grad():
            return GaussianRasterizer(st)(means3D=g["means3D"], means2D=torch.zeros(n, 3, device="cuda"), shs=g["shs"],
                                          opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"],
                                          extra_attrs=g["extra"])

    first = call()
    R = diff_gauss.LAST_STATS["num_rendered"]
    plan = diff_gauss.CapacityPlan([int(R * 1.1) + 64], "cuda")
    diff_gauss.set_capacity_plan(plan)
    try:
        for rep in range(4):
            plan.begin_step()
            outs = call()
            for x, y in zip(outs, first):
                assert torch.equal(x, y), rep
        assert plan.overflowed() == [] and diff_gauss.sort_stalls() == 0
        if os.environ.get("INSTAG_BLEND_FWD") == "segment":
            assert int((plan.walk_hints[0][:(size // 16) ** 2] >= 24).sum()) > 50      # (the test did exercise shared tiles)
    finally:
        diff_gauss.set_capacity_plan(None)


def test_second_backward_and_masked_gradient_reuse_the_state():
    """Backward twice over ONE forward state (retain_graph) with different upstream gradients, the second one zero on
    most of the image: the gradient rows of the first pass must not leak into the second (the reducing kernel clears the
    row flags it consumes; tiles without gradient write no rows).  Reference = the same backward on a fresh forward."""
    a, settings = make_scene(6000, 160, sh_degree=1, seed=4)
    g = torch.Generator().manual_seed(9)
    w1 = torch.randn(3, 160, 160, generator=g).cuda()
    w2 = torch.zeros(3, 160, 160)
    w2[:, 40:72, 56:120] = torch.randn(3, 32, 64, generator=g)          # a few tiles only
    w2 = w2.cuda()

    def grads(inp):
        out = {k: v.grad.clone() for k, v in inp.items() if v is not None and v.grad is not None}
        for v in inp.values():
            if v is not None:
                v.grad = None
        return out

    outs, inp = run_hip(a, settings)
    (outs[0] * w1).sum().backward(retain_graph=True)
    first = grads(inp)
    (outs[0] * w2).sum().backward()
    second = grads(inp)
    outs_f, inp_f = run_hip(a, settings)
    (outs_f[0] * w2).sum().backward()
    fresh = grads(inp_f)
    assert first.keys() == second.keys() == fresh.keys()
    for k in fresh:
        assert torch.equal(second[k], fresh[k]), k
    assert not torch.equal(first["means3D"], second["means3D"])
