"""GPU tests of the train-step callers around the hot path: the face branch's iteration-dependent phases and
density control (train_face.py:340-350, 426-575, 667-788), the mouth branch (train_mouth.py:106-293) and the fuse
stage (train_fuse_con.py:75-245).  Each fused / restructured loss block is compared with the plain-torch statement of
the reference's lines on the same rendered outputs; gradients through the same HIP rasterizer."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


class SmallOpt:
    """OptimizationParams with every schedule boundary pulled into the first dozen iterations."""
    iterations = 14
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 45000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.003
    rotation_lr = 0.001
    percent_dense = 0.005
    lambda_dssim = 0.2
    densification_interval = 3
    opacity_reset_interval = 6
    densify_from_iter = 2
    densify_until_iter = 11
    densify_grad_threshold = 0.00002


def _frames(size, n, dev, **kw):
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import make_frame
    cams = toy_cameras(size)
    return [make_frame(cams[i].to(dev), synthetic_frame(size, i, dev, **kw)) for i in range(n)]


def _detached(obj):
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: _detached(v) for k, v in dict(obj.items()).items()}
    return obj


def _grads(tr):
    out = {k: p.grad.detach().clone() for k, p in tr.g._p.items() if p.grad is not None}
    for name, net in (("umf", tr.motion_net), ("pmf", tr.g.neural_motion_grid)):
        for n_, p_ in net.named_parameters():
            if p_.grad is not None:
                out[f"{name}.{n_}"] = p_.grad.detach().clone()
    return out


@pytest.mark.parametrize("it", [500, 1500, 3004, 3003, 5001, 6050])
def test_face_phase_step_matches_torch_loss_statement(it):
    """Every phase of the face branch: the trainer's loss (fused loss block + prior terms) and the gradients it sends
    to the Gaussians and both motion fields == the reference's lines written with plain torch ops (boolean-mask
    indexing, in-place painting) behind the same render."""
    from instag_amd.deferred import deferred_grads
    from instag_amd.losses import face_loss_torch, normalize
    from instag_amd.renderer import render_motion
    from instag_amd.train import build_trainer, face_phase
    dev = torch.device("cuda")
    frame = _frames(96, 1, dev, priors=True)[0]
    phase = face_phase(it)
    tr = build_trainer(3000, dev, seed=2)
    pkg, loss, l1 = tr._forward_backward(frame, phase)
    got = _grads(tr)
    tr._zero_grad()

    pkg2 = render_motion(frame, tr.g, tr.motion_net, None, tr.bg, return_attn=True, personalized=False,
                         align=phase.align)
    td = frame.talking_dict
    face, hair, mouth = td["face_mask"], td["hair_mask"], td["mouth_mask"]
    extra = alpha = attn = lips = None
    if phase.warm:
        m, pm = pkg2["motion"], pkg2["p_motion"]
        extra = (m["d_xyz"].abs().mean() + m["d_rot"].abs().mean() + m["d_opa"].abs().mean()
                 + m["d_scale"].abs().mean() + pm["p_xyz"].abs().mean())
        alpha, attn, lips = pkg2["alpha"], pkg2["attn"], td["lips_rect"]
    want, want_l1 = face_loss_torch(pkg2["render"], frame.original_image, face, hair, mouth, tr.bg, alpha=alpha,
                                    attn=attn, lips_rect=lips, extra=extra, hair_mask_iter=phase.hair_mask_iter)
    if phase.priors:
        head = face + hair
        want = want + 0.01 * (1 - td["normal"] * pkg2["normal"]).sum(0)[head ^ mouth].mean()
        if phase.prior_depth:
            sel = face ^ mouth
            want = want + 1e-2 * (normalize(pkg2["depth"][0])[sel] - normalize(td["depth"])[sel]).abs().mean()
    with deferred_grads(dev):
        want.backward()
    ref = _grads(tr)
    assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want))), (float(loss), float(want))
    assert abs(float(l1) - float(want_l1)) <= 2e-6
    assert set(got) == set(ref), set(got) ^ set(ref)
    for k in ref:
        scale = float(ref[k].abs().max())
        err = float((got[k] - ref[k]).abs().max())
        assert err <= 2e-4 * scale + 1e-9, (k, err, scale)
    if phase.priors:
        # the normal / depth images carried gradient: the full (all-channel) blend backward ran
        assert float(got["rotation"].abs().max()) > 0


def test_reference_schedule_density_control_order():
    """schedule="reference": statistics -> densify / prune / opacity reset -> optimizers (train_face.py:667-788).  In
    a density-control iteration the rebuilt Gaussians take no Adam step (they carry no gradient), the motion field
    does; every captured graph is dropped when the parameter set changes and -- graph mode staying on -- the next
    iteration captures its step again by itself, without consuming iterations."""
    from instag_amd import diff_gauss
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.motion_net import MotionNetwork, PersonalizedMotionNetwork
    from instag_amd.scene_synth import synthetic_gaussians
    from instag_amd.train import FaceTrainer
    dev = torch.device("cuda")
    torch.manual_seed(3)
    args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    g = GaussianModel(1, neural_motion_grid=PersonalizedMotionNetwork(args=args).to(dev))
    g.load_raw(synthetic_gaussians(3000, sh_degree=1, seed=3), dev)
    tr = FaceTrainer(g, MotionNetwork(args=args).to(dev), torch.tensor([0.0, 1.0, 0.0], device=dev), opt=SmallOpt,
                     densify=True, seed=0, schedule="reference")
    frames = _frames(96, 3, dev, priors=True)
    try:
        counts = []
        for i in range(1, 13):
            due = tr._densify_due(i)
            if i == 5:
                tr.enable_graph(frames[0], warmup_steps=1)      # iterations 5..7 are spent inside enable_graph
                assert tr._graph is not None and tr.iteration == 7
                continue
            if i in (6, 7):
                continue
            w_before = next(tr.motion_net.sigma_net.parameters()).detach().clone()
            out = tr.step(frames[i % 3])
            assert tr.iteration == i
            assert torch.isfinite(out["loss"]), i
            counts.append(tr.g.num_points)
            assert not torch.equal(w_before, next(tr.motion_net.sigma_net.parameters()).detach()), i
            if due:
                assert tr._graph is None, "the graph must be dropped when the parameter set is rebuilt"
                st = tr.g.optimizer.state[tr.g._p["xyz"]]
                assert st["exp_avg"].shape == tr.g._p["xyz"].shape
            elif i >= 8:
                # replayed: iteration 8 from the graph enable_graph captured, 10 / 11 from graphs step() captured itself
                # for the rebuilt parameter set (and the phase of that iteration)
                assert tr._graph is not None and tr._graph.phase == out["phase"] and not tr._graph.check_overflow(), i
                assert tr._graph.static.original_image.shape[-1] == 96
            assert tr.g.xyz_gradient_accum.shape[0] == tr.g.num_points == tr.g.max_radii2D.shape[0]
        assert len(set(counts)) > 1, counts          # densify / prune changed N at least once
        assert tr.recaptures >= 1
        # iteration 12 is a density-control iteration (12 % 3 == 0): run eagerly, every graph dropped
        assert tr._graph is None
    finally:
        diff_gauss.set_capacity_plan(None)


def test_density_control_graph_mode_matches_eager():
    """A run that crosses density-control events in graph mode (captured steps dropped at every event and captured
    again by step() itself, no iteration consumed, training state untouched by the captures) == the same run launched
    eagerly: same Gaussian counts after every event, same losses, same parameters."""
    from instag_amd import diff_gauss
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.motion_net import MotionNetwork, PersonalizedMotionNetwork
    from instag_amd.scene_synth import synthetic_gaussians
    from instag_amd.train import FaceTrainer
    dev = torch.device("cuda")
    frames = _frames(96, 3, dev, priors=True)
    # (no opacity reset inside the run: three iterations after one, the opacity prune would remove every Gaussian)
    Opt = type("Opt", (SmallOpt,), {"iterations": 1000, "densify_until_iter": 1000, "opacity_reset_interval": 1000,
                                    "densification_interval": 4})
    runs = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        args = SimpleNamespace(audio_extractor="deepspeech", type="face")
        g = GaussianModel(1, neural_motion_grid=PersonalizedMotionNetwork(args=args).to(dev))
        g.load_raw(synthetic_gaussians(3000, sh_degree=1, seed=3), dev)
        tr = FaceTrainer(g, MotionNetwork(args=args).to(dev), torch.tensor([0.0, 1.0, 0.0], device=dev), opt=Opt,
                         densify=True, seed=0, schedule="reference")
        try:
            if mode == "graph":
                tr.enable_graph(frames[0], warmup_steps=1, keep_state=True)      # no iteration consumed
                assert tr.iteration == 0
            losses, counts = [], []
            for i in range(1, 23):
                out = tr.step(frames[i % 3])
                assert tr.iteration == i
                losses.append(float(out["loss"]))
                counts.append(tr.g.num_points)
            if mode == "graph":
                assert tr.recaptures >= 4, tr.recaptures
            assert min(counts) > 500, counts
        finally:
            diff_gauss.set_capacity_plan(None)
        runs[mode] = (losses, counts, tr.g.get_xyz.detach().clone(),
                      next(tr.motion_net.sigma_net.parameters()).detach().clone())
    (le, ce, xe, we), (lg, cg, xg, wg) = runs["eager"], runs["graph"]
    assert ce == cg, (ce, cg)
    for a_, b_ in zip(le, lg):
        assert abs(a_ - b_) <= 1e-4 * max(1.0, abs(a_)), (le, lg)
    assert float((xe - xg).abs().max()) <= 1e-5 and float((we - wg).abs().max()) <= 1e-4


def _mouth_setup(dev, n_face=1500, n_mouth=900, seed=4):
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
    torch.manual_seed(seed)
    face_args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    mouth_args = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
    pc_face = GaussianModel(1, PersonalizedMotionNetwork(args=face_args).to(dev)).create_random(n_face, dev, seed=1)
    face_net = MotionNetwork(args=face_args).to(dev)
    pc = GaussianModel(1, PersonalizedMotionNetwork(args=mouth_args).to(dev)).create_random(n_mouth, dev, seed=2)
    net = MouthMotionNetwork(args=mouth_args).to(dev)
    return pc_face, face_net, pc, net


def test_mouth_trainer_phases_and_freeze():
    """MouthTrainer over its three phases (no alignment -> alignment -> warm -> late): loss == the reference's lines on
    the same render, density control keeps the optimizer state consistent, and after bg_iter the geometry and the
    motion field stop moving while the colours keep learning."""
    from instag_amd.losses import l1_loss, ssim
    from instag_amd.train_stages import MouthTrainer, mouth_phase
    dev = torch.device("cuda")
    pc_face, face_net, pc, net = _mouth_setup(dev)
    bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
    # prune threshold 0.05 + 0.25 it / densify_until_iter stays below the initial opacity (0.1), the opacity reset (10)
    # comes after the last densification (9): otherwise the compressed schedule would prune every Gaussian
    MouthOpt = type("MouthOpt", (SmallOpt,), {"opacity_reset_interval": 10, "densify_until_iter": 200})
    tr = MouthTrainer(pc, net, pc_face, face_net, bg, opt=MouthOpt, densify=True, seed=0, warm_step=3, bg_iter=10)
    frames = _frames(96, 3, dev)
    assert mouth_phase(2, SmallOpt, 3).warm is False and mouth_phase(4, SmallOpt, 3).warm
    assert mouth_phase(10, SmallOpt, 3, 10).late is False and mouth_phase(11, SmallOpt, 3, 10).late
    assert mouth_phase(9001).late and not mouth_phase(9000).late             # the reference's bg_iter = 9000

    # loss statement (warm phase) on the trainer's own render
    frame = frames[0]
    pkg, loss, l1 = tr.forward(frame, mouth_phase(4, SmallOpt, 3, 10), k=12)
    td = frame.talking_dict
    mouth = td["mouth_mask"]
    xmin, xmax, ymin, ymax = td["lips_rect"].tolist()
    lips = torch.zeros_like(mouth)
    lips[xmin:xmax, ymin:ymax] = True
    gt = frame.original_image
    gt_green = gt * mouth + bg[:, None, None] * ~mouth
    img = pkg["render"].detach().clone()
    img[:, (lips ^ mouth)] = bg[:, None]
    alpha = pkg["alpha"].detach()
    want_l1 = l1_loss(img, gt_green)
    want = want_l1 + 0.2 * (1.0 - ssim(img, gt_green)) + 1e-5 * pkg["p_motion"]["p_xyz"].detach().abs().mean() \
        + 1e-3 * (((1 - alpha) * lips).mean() + (alpha * ~lips).mean())
    assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want))), (float(loss), float(want))
    assert abs(float(l1) - float(want_l1)) <= 2e-6
    loss.backward()
    assert pc._xyz.grad is not None and all(p.grad is None for p in face_net.parameters())
    tr.motion_optimizer.zero_grad(set_to_none=True)
    tr.g.optimizer.zero_grad(set_to_none=True)

    ks = []
    for i in range(1, 11):                       # iterations 1..10: every phase before bg_iter, density control on
        out = tr.step(frames[i % 3])
        assert torch.isfinite(out["loss"]) and not out["phase"].late, i
        ks.append(out["k"])
        assert tr.g.xyz_gradient_accum.shape[0] == tr.g.num_points
        assert tr.g.optimizer.state[tr.g._p["xyz"]]["exp_avg"].shape == tr.g._p["xyz"].shape
    assert all(10 <= k <= 50 for k in ks) and len(set(ks)) > 1
    xyz0, sc0 = tr.g._p["xyz"].detach().clone(), tr.g._p["scaling"].detach().clone()
    fdc0 = tr.g._p["f_dc"].detach().clone()
    w0 = next(net.sigma_net.parameters()).detach().clone()
    tr.densify = False                           # (the reference's schedule has no density control after bg_iter)
    for i in range(11, 14):                      # bg_iter = 10: black background, geometry + motion field frozen
        out = tr.step(frames[i % 3])
        assert out["phase"].late and out["phase"].warm and torch.isfinite(out["loss"])
    assert torch.equal(xyz0, tr.g._p["xyz"].detach()) and torch.equal(sc0, tr.g._p["scaling"].detach())
    assert torch.equal(w0, next(net.sigma_net.parameters()).detach())
    assert not torch.equal(fdc0, tr.g._p["f_dc"].detach())


def test_fuse_trainer_step():
    """FuseTrainer: composition == train_fuse_con.py:106-121 on the two renders, frozen parameters stay put, the
    trainable ones (face colours + opacity, mouth colours) move, and the loss goes down on a repeated frame."""
    from instag_amd.losses import l1_loss, ssim
    from instag_amd.train_stages import FuseTrainer
    dev = torch.device("cuda")
    pc_face, face_net, pc_mouth, mouth_net = _mouth_setup(dev, n_face=3000, n_mouth=800, seed=6)
    bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
    tr = FuseTrainer(pc_face, face_net, pc_mouth, mouth_net, bg)
    frame = _frames(96, 1, dev, background=True)[0]
    out, loss, l1 = tr.forward(frame)
    fa, ma = out["face"]["alpha"], out["mouth"]["alpha"]
    scene_bg = frame.talking_dict["background"]
    mouth_image = out["mouth"]["render"] - bg[:, None, None] * (1.0 - ma) + scene_bg * (1.0 - ma)
    image = out["face"]["render"] - bg[:, None, None] * (1.0 - fa) + mouth_image * (1.0 - fa)
    assert float((out["image"] - image).abs().max()) <= 1e-6
    want = l1_loss(image, frame.original_image) + 0.2 * (1.0 - ssim(image, frame.original_image))
    assert abs(float(loss) - float(want)) <= 2e-6
    loss.backward()
    assert pc_face._p["xyz"].grad is None and pc_mouth._p["opacity"].grad is None
    assert pc_face._p["opacity"].grad is not None and pc_mouth._p["f_dc"].grad is not None
    assert all(p.grad is None for p in face_net.parameters()) and all(p.grad is None for p in mouth_net.parameters())
    tr.g.optimizer.zero_grad(set_to_none=True)
    tr.g_mouth.optimizer.zero_grad(set_to_none=True)

    frozen = {("f", k): pc_face._p[k].detach().clone() for k in FuseTrainer.FROZEN_FACE}
    frozen.update({("m", k): pc_mouth._p[k].detach().clone() for k in FuseTrainer.FROZEN_MOUTH})
    op0, fdc0, mdc0 = (pc_face._p["opacity"].detach().clone(), pc_face._p["f_dc"].detach().clone(),
                       pc_mouth._p["f_dc"].detach().clone())
    losses = [float(tr.step(frame)["loss"]) for _ in range(12)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0], losses
    for (which, k), v in frozen.items():
        cur = (pc_face if which == "f" else pc_mouth)._p[k].detach()
        assert torch.equal(v, cur), (which, k)
    assert not torch.equal(op0, pc_face._p["opacity"].detach())
    assert not torch.equal(fdc0, pc_face._p["f_dc"].detach()) and not torch.equal(mdc0, pc_mouth._p["f_dc"].detach())


def test_stage_trainers_graph_matches_eager():
    """Whole-step hipGraph of the mouth and fuse steps == the eager steps (same seeds, same frames, same random k
    sequence): losses and parameters agree after the replays; the mouth graph is dropped on a density-control
    iteration and the step falls back to eager launches."""
    from instag_amd import diff_gauss
    from instag_amd.train_stages import FuseTrainer, MouthTrainer
    dev = torch.device("cuda")
    bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
    frames = _frames(96, 3, dev, background=True)
    NoDensify = type("NoDensify", (SmallOpt,), {"iterations": 100000})

    def run(kind, graph):
        pc_face, face_net, pc_mouth, mouth_net = _mouth_setup(dev, n_face=2000, n_mouth=900, seed=7)
        if kind == "mouth":
            # warm_step = 0: every iteration is in the same phase (the graph's warm-up steps run in the phase it captures)
            tr = MouthTrainer(pc_mouth, mouth_net, pc_face, face_net, bg, opt=NoDensify, densify=False, seed=3,
                              warm_step=0, bg_iter=1000)
        else:
            tr = FuseTrainer(pc_face, face_net, pc_mouth, mouth_net, bg, opt=NoDensify)
        try:
            if graph:
                tr.enable_graph(frames[0], warmup_steps=2)          # 4 real steps on frame 0
                assert tr.iteration == 4 and tr._graph is not None
            else:
                for _ in range(4):
                    tr.step(frames[0])
            losses = [float(tr.step(frames[i % 3])["loss"]) for i in range(6)]
            if graph:
                assert tr._graph is not None and not tr._graph.check_overflow()
        finally:
            diff_gauss.set_capacity_plan(None)
        vec = torch.cat([tr.g._p["f_dc"].detach().reshape(-1), tr.g._p["opacity"].detach().reshape(-1),
                         tr.g._p["xyz"].detach().reshape(-1)])
        return losses, vec, tr

    for kind in ("mouth", "fuse"):
        le, ve, _ = run(kind, False)
        lg, vg, tr = run(kind, True)
        assert tr.iteration == 10
        # The mouth field's 186 KB planes take the generic grid encoder, whose backward flushes its LDS sums with global
        # float atomics (csrc/grid.hip): the summation order, hence the last bits of the table gradient, differ from
        # run to run, and ten Adam steps amplify that -- 2.6e-5 on a parameter has been observed between two runs of
        # the SAME mode.  (The face path, all fixed-order sums, is compared at 1e-5 in test_raster_gpu.py.)
        for a_, b_ in zip(le, lg):
            assert abs(a_ - b_) <= 1e-4 * max(1.0, abs(a_)), (kind, le, lg)
        assert float((ve - vg).abs().max()) <= 2e-4, kind

    # density control on: the captured step is abandoned on the iteration that densifies
    pc_face, face_net, pc_mouth, mouth_net = _mouth_setup(dev, n_face=2000, n_mouth=900, seed=8)
    Opt = type("Opt", (SmallOpt,), {"iterations": 100000, "densify_until_iter": 200, "opacity_reset_interval": 1000})
    tr = MouthTrainer(pc_mouth, mouth_net, pc_face, face_net, bg, opt=Opt, densify=True, seed=3, warm_step=0, bg_iter=1000)
    try:
        tr.enable_graph(frames[0], warmup_steps=2)                  # iterations 1..4
        out = tr.step(frames[1])                                    # 5: replayed
        assert tr._graph is not None and torch.isfinite(out["loss"])
        n0 = tr.g.num_points
        out = tr.step(frames[2])                                    # 6 % 3 == 0: density control, eager
        assert tr._graph is None and torch.isfinite(out["loss"])
        assert tr.g.xyz_gradient_accum.shape[0] == tr.g.num_points
        out = tr.step(frames[0])
        assert torch.isfinite(out["loss"]) and (tr.g.num_points != n0 or True)
    finally:
        diff_gauss.set_capacity_plan(None)


def test_face_schedule_replays_one_graph_per_phase():
    """schedule="reference" around the hair iterations (six of seven iterations paint the hair to background,
    train_face.py:340): one captured step per phase, step() replays whichever matches the iteration, and a replayed
    step's loss == the eager loss of the same phase on the same state."""
    from instag_amd import diff_gauss
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.motion_net import MotionNetwork, PersonalizedMotionNetwork
    from instag_amd.scene_synth import synthetic_gaussians
    from instag_amd.train import FacePhase, FaceTrainer, face_phase
    dev = torch.device("cuda")
    torch.manual_seed(11)
    args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    g = GaussianModel(1, neural_motion_grid=PersonalizedMotionNetwork(args=args).to(dev))
    g.load_raw(synthetic_gaussians(3000, sh_degree=1, seed=5), dev)
    tr = FaceTrainer(g, MotionNetwork(args=args).to(dev), torch.tensor([0.0, 1.0, 0.0], device=dev), densify=False,
                     schedule="reference")
    frames = _frames(96, 3, dev)
    tr.iteration = 3100
    hair, plain = FacePhase(hair_mask_iter=True), FacePhase()
    assert face_phase(3108) == plain and face_phase(3109) == hair          # 3108 = 7 * 444
    try:
        tr.enable_graph(frames[0], warmup_steps=1, phase=hair)             # iterations 3101..3103
        tr.enable_graph(frames[0], warmup_steps=1, phase=plain)            # iterations 3104..3106
        assert tr.iteration == 3106 and set(tr._graph_cache) == {hair, plain}
        seen = set()
        for it in range(3107, 3118):
            frame = frames[it % 3]
            phase = face_phase(it)
            diff_gauss.set_capacity_plan(None)
            _, want, _ = tr._forward_backward(frame, phase)                # eager loss of this phase on this state
            want = float(want)
            tr._zero_grad()
            out = tr.step(frame)
            assert tr.iteration == it and out["phase"] == phase
            assert tr._graph is tr._graph_cache[phase], it                 # replayed, not launched eagerly
            assert not tr._graph.check_overflow()
            assert abs(float(out["loss"]) - want) <= 1e-5 * max(1.0, abs(want)), (it, float(out["loss"]), want)
            seen.add(phase)
        assert seen == {hair, plain}
        tr._drop_graph()
        assert tr._graph is None and not tr._graph_cache
    finally:
        diff_gauss.set_capacity_plan(None)


def _plain_render_motion(frame, pc, motion_net, bg, personalized, align):
    """gaussian_renderer/__init__.py:188-283 written with plain torch ops and the reference's IN-PLACE arithmetic on the
    motion dictionary (d_xyz += ..., d_xyz *= p_scale), two or three separate rasterizer calls."""
    import math
    from instag_amd.diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    s = GaussianRasterizationSettings(
        image_height=frame.image_height, image_width=frame.image_width, tanfovx=math.tan(frame.FoVx * 0.5),
        tanfovy=math.tan(frame.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=frame.world_view_transform,
        projmatrix=frame.full_proj_transform, sh_degree=pc.active_sh_degree, campos=frame.camera_center,
        prefiltered=False, debug=False)
    rast = GaussianRasterizer(s)
    aud, exp = frame.talking_dict["auds"], frame.talking_dict["au_exp"]
    xyz = pc.get_xyz
    p = pc.neural_motion_grid(pc.get_xyz, aud, exp) if (personalized or align) else None
    if align:
        xyz = xyz + p["p_xyz"]
    m = dict(motion_net(xyz, aud, exp).items())            # materialised entries, like the reference's dict
    d_xyz, d_scale, d_rot = m["d_xyz"], m["d_scale"], m["d_rot"]
    if personalized:
        d_xyz += p["d_xyz"]
        d_scale += p["d_scale"]
        d_rot += p["d_rot"]
    if align:
        d_xyz *= p["p_scale"]
    means3D = pc.get_xyz + d_xyz
    opacity = pc.get_opacity
    scales = pc.scaling_activation(pc._scaling + d_scale)
    rots = pc.rotation_activation(pc._rotation + d_rot)
    m2 = torch.zeros_like(pc.get_xyz, requires_grad=True)
    ones = torch.ones_like(opacity)
    img, depth, normal, alpha, radii, extra = rast(means3D=means3D, means2D=m2, shs=pc.get_features, opacities=opacity,
                                                   scales=scales, rotations=rots, extra_attrs=ones)

    def attn_of(preds):
        col = torch.cat([preds["ambient_aud"], preds["ambient_eye"], torch.zeros_like(preds["ambient_eye"])], dim=-1)
        return rast(means3D=means3D.detach(), means2D=m2, colors_precomp=col, opacities=opacity.detach(),
                    scales=scales.detach(), rotations=rots.detach(), extra_attrs=ones)[0]
    return dict(render=img, depth=depth, normal=normal, alpha=alpha, radii=radii, motion=m, attn=attn_of(m),
                p_attn=attn_of(p) if personalized else None, viewspace_points=m2, p_motion=p)


def _plain_fuse_inference(frame, pc, net, pcm, netm, bg, personalized, scene_bg, k=10):
    """synthesize_fuse.py:46-74 with plain torch ops and separate rasterizer calls: render_motion(align=True) ->
    render_motion_mouth_con(align=True, inference=True) -> composite.  At inference the mouth branch reads
    ``motion_net_face.cache`` (gaussian_renderer/__init__.py:362-363), which in the reference is the very dictionary
    render_motion has just updated in place (d_xyz += p.d_xyz; d_xyz *= p_scale, :207-217): here ``face["motion"]``."""
    import math
    from instag_amd.diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    with torch.no_grad():
        face = _plain_render_motion(frame, pc, net, bg, personalized, True)
        cache = face["motion"]
        s = GaussianRasterizationSettings(
            image_height=frame.image_height, image_width=frame.image_width, tanfovx=math.tan(frame.FoVx * 0.5),
            tanfovy=math.tan(frame.FoVy * 0.5), bg=bg, scale_modifier=1.0, viewmatrix=frame.world_view_transform,
            projmatrix=frame.full_proj_transform, sh_degree=pcm.active_sh_degree, campos=frame.camera_center,
            prefiltered=False, debug=False)
        aud = frame.talking_dict["auds"]
        p = pcm.neural_motion_grid(pcm.get_xyz, aud)                                   # :349-350
        xyz = pcm.get_xyz + p["p_xyz"]                                                 # :352-353
        dy = cache["d_xyz"][..., 1]                                                    # :362-363, 366
        motion_max = dy.topk(k, 0, True, True)[0]
        motion_min = dy.topk(k, 0, False, True)[0]
        move = torch.stack([motion_max[-1], motion_min[-1], motion_max[-1] - motion_min[-1]]).reshape(1, 3) * 1e2
        m = dict(netm(xyz, aud, move).items())
        d_xyz = m["d_xyz"]
        if personalized:
            d_xyz += p["d_xyz"]                                                        # :385-390
        opacity = pcm.get_opacity
        mr, _, _, ma, _, _ = GaussianRasterizer(s)(
            means3D=pcm.get_xyz + d_xyz, means2D=torch.zeros_like(pcm.get_xyz), shs=pcm.get_features, opacities=opacity,
            scales=pcm.get_scaling, rotations=pcm.rotation_activation(pcm._rotation), extra_attrs=torch.ones_like(opacity))
        mouth_image = mr + scene_bg * (1.0 - ma)                                       # synthesize_fuse.py:66
        image = face["render"] + mouth_image * (1.0 - face["alpha"])                   # :70
        return image.clamp(0, 1)                                                       # :72


@pytest.mark.parametrize("personalized", [False, True], ids=["align", "personalized+align"])
def test_fuse_inference_matches_plain_torch(personalized):
    """SURVEY 8(f)4 value test: FuseRenderer (eager, one hipGraph per frame, three frames per replay on three lanes) ==
    the reference's inference loop transcribed with plain torch ops (synthesize_fuse.py:46-74), including the mouth
    branch reading the face field's cache (gaussian_renderer/__init__.py:362-372)."""
    from instag_amd import diff_gauss
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.infer import FuseRenderer
    from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
    dev = torch.device("cuda")
    torch.manual_seed(21)
    face_args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    mouth_args = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
    pc = GaussianModel(1, PersonalizedMotionNetwork(args=face_args).to(dev)).create_random(3000, dev, seed=1)
    pcm = GaussianModel(1, PersonalizedMotionNetwork(args=mouth_args).to(dev)).create_random(800, dev, seed=2)
    net, netm = MotionNetwork(args=face_args).to(dev), MouthMotionNetwork(args=mouth_args).to(dev)
    with torch.no_grad():
        # random-init fields move nothing visible: scale the output layers up so the jaw feature, both displacements and
        # the alignment actually shape the images
        for mod in (net.sigma_net, netm.sigma_net, pc.neural_motion_grid.sigma_net, pc.neural_motion_grid.align_net,
                    pcm.neural_motion_grid.sigma_net, pcm.neural_motion_grid.align_net):
            mod.net[-1].weight.mul_(30.0)
    frames = _frames(96, 3, dev)
    bg = torch.zeros(3, device=dev)
    scene_bgs = [torch.rand(3, 96, 96, device=dev) for _ in frames]
    want = [_plain_fuse_inference(f, pc, net, pcm, netm, bg, personalized, sb) for f, sb in zip(frames, scene_bgs)]
    assert float((want[0] - want[1]).abs().max()) > 1e-2                      # the frames differ
    static = _plain_fuse_inference(frames[0], pc, net, pcm, netm, bg, personalized, scene_bgs[1])
    assert float((static - want[0]).abs().max()) > 1e-3                       # the background shows through
    r = FuseRenderer(pc, net, pcm, netm, bg, personalized=personalized)
    worst = 0.0
    try:
        for f, sb, w in zip(frames, scene_bgs, want):
            worst = max(worst, float((r.render(f, sb) - w).abs().max()))
        r.enable_graph(frames[0])
        for f, sb, w in zip(frames, scene_bgs, want):
            got = r.render(f, sb)
            assert not r.check_overflow()
            worst = max(worst, float((got - w).abs().max()))
        r.close()
        r.enable_graph(frames[0], frames_per_replay=3)
        got = r.render_batch(frames, scene_bgs)
        assert not r.check_overflow()
        for k, w in enumerate(want):
            worst = max(worst, float((got[k] - w).abs().max()))
        order = (2, 0, 1, 1, 0)
        five = r.render_batch([frames[j] for j in order], [scene_bgs[j] for j in order])
        for k, j in enumerate(order):
            worst = max(worst, float((five[k] - want[j]).abs().max()))
    finally:
        r.close()
        diff_gauss.set_capacity_plan(None)
    assert worst <= 2e-6, worst


@pytest.mark.parametrize("personalized,align", [(True, True), (True, False), (False, True), (False, False)],
                         ids=["personalized+align", "personalized", "align(fused operator)", "plain"])
def test_render_motion_branches_match_plain_torch(personalized, align):
    """render_motion in every (personalized, align) combination == the reference's lines transcribed with plain torch
    ops: images, both attention maps, the motion dictionary AFTER the reference's in-place updates (what the
    regularisers of train_face.py:508-514 read) and motion_net.cache (what the mouth branch reads at inference,
    gaussian_renderer/__init__.py:362-363), plus gradients of a loss over image + attention maps + d_xyz."""
    from instag_amd.renderer import render_motion
    from instag_amd.train import build_trainer
    dev = torch.device("cuda")
    frame = _frames(96, 1, dev, priors=True)[0]
    tr = build_trainer(2500, dev, seed=4)
    g = torch.Generator().manual_seed(9)
    w_img, w_att = torch.randn(3, 96, 96, generator=g).to(dev), torch.randn(3, 96, 96, generator=g).to(dev)

    def loss_of(pkg):
        ls = (pkg["render"] * w_img).sum() + pkg["alpha"].sum() + (pkg["attn"] * w_att).sum() \
            + 10.0 * pkg["motion"]["d_xyz"].abs().mean() + pkg["motion"]["d_rot"].abs().mean()
        if personalized:
            ls = ls + (pkg["p_attn"] * w_att).sum()
        return ls

    want = _plain_render_motion(frame, tr.g, tr.motion_net, tr.bg, personalized, align)
    loss_of(want).backward()
    ref = _grads(tr)
    ref_m2d = want["viewspace_points"].grad.clone()
    tr._zero_grad()
    # (values only from here on: a live autograd graph of the plain pass would keep the parameters' gradient accumulators,
    # bound to the default stream, into the backward of render_motion, whose per-frame branches run on side streams)
    want = _detached(want)
    got = render_motion(frame, tr.g, tr.motion_net, None, tr.bg, return_attn=True, personalized=personalized,
                        align=align)
    cache_dxyz = tr.motion_net.cache["d_xyz"]
    loss_of(got).backward()
    mine = _grads(tr)
    for k in ("render", "alpha", "depth", "normal", "attn") + (("p_attn",) if personalized else ()):
        assert float((got[k] - want[k]).abs().max()) <= 2e-6, k
    assert torch.equal(got["radii"], want["radii"])
    for k in ("d_xyz", "d_rot", "d_scale"):
        assert float((got["motion"][k] - want["motion"][k]).abs().max()) <= 1e-7, k
    assert not cache_dxyz.requires_grad
    assert float((cache_dxyz - want["motion"]["d_xyz"].detach()).abs().max()) <= 1e-7
    assert set(mine) == set(ref), set(mine) ^ set(ref)
    for k in ref:
        scale = float(ref[k].abs().max())
        assert float((mine[k] - ref[k]).abs().max()) <= 2e-4 * scale + 1e-9, k
    m2d = got["viewspace_points"].grad
    assert float((m2d - ref_m2d).abs().max()) <= 2e-4 * float(ref_m2d.abs().max()) + 1e-9


def test_static_render_matches_direct_rasterizer_call():
    """render() (gaussian_renderer/__init__.py:37-133): the dictionary it returns == a direct rasterizer call on the
    model's activated parameters; override_color switches to colors_precomp."""
    import math
    from instag_amd.diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from instag_amd.renderer import render
    from instag_amd.train import build_trainer
    dev = torch.device("cuda")
    frame = _frames(96, 1, dev)[0]
    tr = build_trainer(2500, dev, seed=6)
    pc = tr.g
    s = GaussianRasterizationSettings(
        image_height=frame.image_height, image_width=frame.image_width, tanfovx=math.tan(frame.FoVx * 0.5),
        tanfovy=math.tan(frame.FoVy * 0.5), bg=tr.bg, scale_modifier=1.0, viewmatrix=frame.world_view_transform,
        projmatrix=frame.full_proj_transform, sh_degree=pc.active_sh_degree, campos=frame.camera_center,
        prefiltered=False, debug=False)
    for override in (None, torch.rand(pc.get_xyz.shape[0], 3, device=dev)):
        pkg = render(frame, pc, None, tr.bg, override_color=override)
        m2 = torch.zeros_like(pc.get_xyz, requires_grad=True)
        img, depth, normal, alpha, radii, extra = GaussianRasterizer(s)(
            means3D=pc.get_xyz, means2D=m2, shs=pc.get_features if override is None else None,
            colors_precomp=override, opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation,
            extra_attrs=torch.ones_like(pc.get_opacity))
        assert torch.equal(pkg["render"], img) and torch.equal(pkg["depth"], depth) and torch.equal(pkg["alpha"], alpha)
        assert torch.equal(pkg["normal"], normal) and torch.equal(pkg["radii"], radii)
        assert torch.equal(pkg["visibility_filter"], radii > 0)
        pkg["render"].sum().backward()
        img.sum().backward()
        assert torch.equal(pkg["viewspace_points"].grad, m2.grad)
        tr._zero_grad()


@pytest.mark.parametrize("warm", [False, True])
def test_fused_mouth_loss_matches_plain_torch(warm):
    """Mouth-branch loss block as the fused kernels' mouth mode == instag_amd.train_stages.mouth_loss (the plain-torch
    statement of train_mouth.py:186-221, itself checked against the reference's lines on the CPU): loss, L1 and the
    gradients of image, alpha and p_xyz."""
    from instag_amd.losses import mouth_loss_fused
    from instag_amd.train_stages import _lips_mask, mouth_loss
    g = torch.Generator().manual_seed(4)
    H, W = 150, 170
    image0, alpha0 = torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g)
    gt = torch.rand(3, H, W, generator=g).cuda()
    mouth = torch.zeros(H, W, dtype=torch.bool)
    mouth[60:110, 50:130] = torch.rand(50, 80, generator=g) > 0.3
    mouth = mouth.cuda()
    rect = torch.tensor([70, 120, 40, 120], dtype=torch.int32, device="cuda")
    bg = torch.tensor([0.0, 1.0, 0.0], device="cuda")
    p0 = torch.randn(3000, 3, generator=g) * 1e-2

    def run(fused):
        image, alpha, p = (t.clone().cuda().requires_grad_(True) for t in (image0, alpha0, p0))
        if fused:
            loss, l1 = mouth_loss_fused(image, alpha, gt, mouth, rect, bg, p, warm=warm)
        else:
            loss, l1 = mouth_loss(image, alpha, gt, mouth, _lips_mask(mouth, rect), bg, p, warm=warm)
        (loss + 0.5 * l1).backward()
        return loss.detach(), l1.detach(), image.grad, alpha.grad, p.grad

    ref, got = run(False), run(True)
    assert abs(float(ref[0]) - float(got[0])) <= 2e-6 and abs(float(ref[1]) - float(got[1])) <= 2e-6
    assert float((ref[2] - got[2]).abs().max()) <= 2e-5 * float(ref[2].abs().max())
    if warm:
        assert float((ref[3] - got[3]).abs().max()) <= 1e-6 * float(ref[3].abs().max()) + 1e-12
        assert float((ref[4] - got[4]).abs().max()) <= 1e-6 * float(ref[4].abs().max()) + 1e-12
    else:
        assert got[3] is None or float(got[3].abs().max()) == 0.0
