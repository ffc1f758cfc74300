"""GPU parity: fused glue operators vs the plain PyTorch formulations they replace."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, name, tol=2e-5):
    err = float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
    scale = max(1.0, float(b.detach().abs().max()))
    assert err <= tol * scale, f"{name}: err {err} scale {scale}"


def test_motion_glue_matches_torch():
    from instag_amd.glue import motion_glue
    g = torch.Generator().manual_seed(0)
    N = 5003
    leafs = dict(enc_x=torch.randn(N, 36, generator=g), aud=torch.randn(N, 32, generator=g),
                 eye=torch.randn(N, 6, generator=g), enc_a=torch.randn(1, 32, generator=g),
                 enc_e=torch.randn(6, generator=g))
    leafs["eye"][:7] = -1.0          # rows whose relu output is all zero: norm 0 -> zero gradient
    wh, wa = torch.randn(N, 74, generator=g), torch.randn(N, 2, generator=g)

    def ref(t):
        eye_att = torch.relu(t["eye"])
        h = torch.cat([t["enc_x"], t["enc_a"].repeat(N, 1) * t["aud"], t["enc_e"] * eye_att], dim=-1)
        return h, torch.cat([t["aud"].norm(dim=-1, keepdim=True), eye_att.norm(dim=-1, keepdim=True)], dim=-1)

    td = {k: v.double().requires_grad_(True) for k, v in leafs.items()}
    h_r, a_r = ref(td)
    ((h_r * wh.double()).sum() + (a_r * wa.double()).sum()).backward()
    th = {k: v.cuda().requires_grad_(True) for k, v in leafs.items()}
    h_h, a_h = motion_glue(th["enc_x"], th["aud"], th["eye"], th["enc_a"], th["enc_e"])
    assert a_h.shape == (N, 3) and float(a_h[:, 2].abs().max()) == 0.0      # (aud, eye, 0)
    ((h_h * wh.cuda()).sum() + (a_h[:, :2] * wa.cuda()).sum()).backward()
    _close(h_h, h_r, "h_in")
    _close(a_h[:, :2], a_r, "amb")
    for k in leafs:
        _close(th[k].grad, td[k].grad, "d_" + k, tol=1e-4)


def test_deform_activate_matches_torch():
    from instag_amd.glue import deform_activate
    g = torch.Generator().manual_seed(1)
    N = 4001
    leafs = dict(xyz=torch.randn(N, 3, generator=g) * 0.1, scaling=torch.randn(N, 3, generator=g) * 2 - 4,
                 rotation=torch.randn(N, 4, generator=g), opacity=torch.randn(N, 1, generator=g) * 2,
                 h=torch.randn(N, 11, generator=g), p=torch.randn(N, 6, generator=g) * 3)
    ws = [torch.randn(N, k, generator=g) for k in (3, 3, 4, 1)]

    def ref(t):
        d_xyz = t["h"][:, :3] * 1e-2 * (torch.tanh(t["p"][:, 3:] / 5) * 0.25 + 1)
        return (t["xyz"] + d_xyz, torch.nn.functional.softplus(t["scaling"] + t["h"][:, 8:11]),
                torch.nn.functional.normalize(t["rotation"] + t["h"][:, 3:7]), torch.sigmoid(t["opacity"]))

    td = {k: v.double().requires_grad_(True) for k, v in leafs.items()}
    sum((o * w.double()).sum() for o, w in zip(ref(td), ws)).backward()
    th = {k: v.cuda().requires_grad_(True) for k, v in leafs.items()}
    outs = deform_activate(th["xyz"], th["scaling"], th["rotation"], th["opacity"], th["h"], th["p"])
    sum((o * w.cuda()).sum() for o, w in zip(outs, ws)).backward()
    for o, r, n in zip(outs, ref(td), ("means3D", "scales", "rotations", "opacity")):
        _close(o, r, n)
    for k in leafs:
        _close(th[k].grad, td[k].grad, "d_" + k, tol=1e-4)


def test_motion_l1_reg_matches_torch():
    from instag_amd.glue import motion_l1_reg
    g = torch.Generator().manual_seed(2)
    h, p = torch.randn(7777, 11, generator=g), torch.randn(7777, 6, generator=g)

    def ref(h, p):
        return ((h[:, :3] * 1e-2).abs().mean() + h[:, 3:7].abs().mean() + h[:, 7:8].abs().mean()
                + h[:, 8:11].abs().mean() + (p[:, :3] * 1e-2).abs().mean())

    hd, pd = h.double().requires_grad_(True), p.double().requires_grad_(True)
    (3.0 * ref(hd, pd)).backward()
    hh, ph = h.cuda().requires_grad_(True), p.cuda().requires_grad_(True)
    r = motion_l1_reg(hh, ph)
    (3.0 * r).backward()
    assert abs(float(r) - float(ref(hd, pd))) < 1e-6
    _close(hh.grad, hd.grad, "dh", tol=1e-6)
    _close(ph.grad, pd.grad, "dp", tol=1e-6)


def test_multi_tensor_adam_matches_torch():
    """One-launch Adam/AdamW == torch.optim.Adam / AdamW over several steps, groups, learning-rate changes."""
    from instag_amd.optim import MultiTensorAdam
    g = torch.Generator().manual_seed(3)
    shapes = [(10000, 3), (10000, 1, 3), (10000, 3, 3), (64, 74), (11, 64), (9464, 1), (5,)]
    base = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * (0.1 if k % 2 else 3.0) for s in shapes] for k in range(6)]

    def run(make, adamw):
        ps = [torch.nn.Parameter(b.clone().cuda()) for b in base]
        groups = [{"params": [ps[0]], "lr": 1.6e-4}, {"params": [ps[1], ps[2]], "lr": 2.5e-3},
                  {"params": ps[3:5], "lr": 5e-4, "weight_decay": 0.01 if adamw else 0.0},
                  {"params": ps[5:], "lr": 5e-3}]
        opt = make(groups)
        for k, gs in enumerate(grads):
            for i, (p, gr) in enumerate(zip(ps, gs)):
                p.grad = None if (i == 6 and k < 2) else gr.clone().cuda()     # a parameter that starts without grad
            opt.param_groups[0]["lr"] = 1.6e-4 * (0.9 ** k)
            if hasattr(opt, "set_lrs"):
                opt.set_lrs()
            opt.step()
        return [p.detach().cpu() for p in ps]

    for adamw in (False, True):
        if adamw:
            ref = run(lambda gr: torch.optim.AdamW(gr, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0), True)
            got = run(lambda gr: MultiTensorAdam(gr, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, decoupled=True), True)
        else:
            ref = run(lambda gr: torch.optim.Adam(gr, lr=0.0, eps=1e-15), False)
            got = run(lambda gr: MultiTensorAdam(gr, lr=0.0, betas=(0.9, 0.999), eps=1e-15), False)
        for a, b in zip(got, ref):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("split", [True, False], ids=["bwd-8wg", "bwd-1wg"])
@pytest.mark.parametrize("extractor,net", [("deepspeech", "umf"), ("esperanto", "pmf")])
def test_frame_codes_match_torch_modules(extractor, net, split, monkeypatch):
    """AudioNet + AudioAttNet + expression MLP as fused kernels vs the nn.Module chain (fp64, CPU); backward with one
    workgroup per audio window (default) and as a single workgroup."""
    import copy
    from types import SimpleNamespace
    from instag_amd import audio as A
    monkeypatch.setattr(A, "SPLIT_BACKWARD", split)
    from instag_amd.motion_net import MotionNetwork, PersonalizedMotionNetwork, audio_in_dim

    class NoEncoder(torch.nn.Module):          # the tri-plane encoders play no part in the per-frame branch
        def __init__(self, **kw):
            super().__init__()
            self.output_dim = 12

    torch.manual_seed(3)
    args = SimpleNamespace(audio_extractor=extractor, type="face")
    cls = MotionNetwork if net == "umf" else PersonalizedMotionNetwork
    ref = cls(args=args, encoder_cls=NoEncoder).double()
    with torch.no_grad():
        for p in ref.parameters():             # biases and weights large enough to exercise both LeakyReLU sides
            p.mul_(2.0)
    dev = copy.deepcopy(ref).float().cuda()
    a = torch.randn(8, audio_in_dim(extractor), 16)
    e = torch.rand(6)
    wa, we = torch.randn(1, 32), torch.randn(6)

    enc_a_r, enc_e_r = ref.encode_frame(a.double(), e.double())
    ((enc_a_r * wa.double()).sum() + (enc_e_r * we.double()).sum()).backward()
    assert A.supported(dev, a.cuda(), e.cuda())
    enc_a_h, enc_e_h = dev.encode_frame(a.cuda(), e.cuda())
    ((enc_a_h * wa.cuda()).sum() + (enc_e_h * we.cuda()).sum()).backward()
    _close(enc_a_h, enc_a_r, "enc_a")
    _close(enc_e_h, enc_e_r, "enc_e")
    names = [n for n, _ in ref.named_parameters()
             if n.startswith(("audio_net", "audio_att_net", "exp_encode_net"))]
    assert len(names) == 26
    pr, ph = dict(ref.named_parameters()), dict(dev.named_parameters())
    for n in names:
        assert ph[n].grad is not None, n
        _close(ph[n].grad, pr[n].grad, "d_" + n, tol=5e-5)


@pytest.mark.parametrize("hair_mask_iter,size", [(False, (96, 80)), (True, (70, 53))])
def test_face_loss_matches_torch(hair_mask_iter, size):
    """Fused loss block (gt_white composition, L1 + DSSIM, alpha / attention / lips terms) vs torch fp64."""
    from instag_amd.losses import face_loss, face_loss_torch
    H, W = size
    g = torch.Generator().manual_seed(11)
    image, gt = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    alpha, attn = torch.rand(1, H, W, generator=g), torch.rand(3, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    face = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.3 * H) ** 2
    hair = (((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.4 * H) ** 2) & (yy < 0.4 * H) & ~face
    mouth = ((yy - 0.65 * H) ** 2 + (xx - W / 2) ** 2) < (0.08 * H) ** 2
    bg = torch.tensor([0.0, 1.0, 0.0])
    lips = torch.tensor([int(0.55 * H), int(0.75 * H), int(0.3 * W), int(0.7 * W)], dtype=torch.int32)
    extra = torch.tensor(0.37)

    def run(fn, dev, dt):
        leaves = [t.to(dev, dt).requires_grad_(True) for t in (image, alpha, attn, extra)]
        loss, l1 = fn(leaves[0], gt.to(dev, dt), face.to(dev), hair.to(dev), mouth.to(dev), bg.to(dev, dt),
                      alpha=leaves[1], attn=leaves[2], lips_rect=lips.to(dev), extra=leaves[3],
                      hair_mask_iter=hair_mask_iter)
        (loss + 0.5 * l1).backward()
        return loss, l1, [t.grad for t in leaves]

    loss_r, l1_r, g_r = run(face_loss_torch, "cpu", torch.float64)
    loss_h, l1_h, g_h = run(face_loss, "cuda", torch.float32)
    assert abs(float(loss_h) - float(loss_r)) < 2e-6, (float(loss_h), float(loss_r))
    assert abs(float(l1_h) - float(l1_r)) < 2e-6
    for name, a, b in zip(("d_image", "d_alpha", "d_attn", "d_extra"), g_h, g_r):
        err = float((a.double().cpu() - b).abs().max())
        assert err <= 2e-5 * max(float(b.abs().max()), 1e-4), f"{name}: {err} vs {float(b.abs().max())}"
    if hair_mask_iter:
        assert float(g_h[0][:, hair.cuda()].abs().max()) == 0.0

    # the scalar stage deferred into the backward launch (what the train steps use): the same bits -- values once
    # backward has run, gradients always
    from instag_amd.losses import defer_finalize

    def deferred(*a, **kw):
        with defer_finalize():
            return face_loss(*a, **kw)
    loss_d, l1_d, g_d = run(deferred, "cuda", torch.float32)
    assert float(loss_d) == float(loss_h) and float(l1_d) == float(l1_h)
    for a, b in zip(g_d, g_h):
        assert torch.equal(a, b)
    # without a gradient to compute nothing is deferred: the value is there at once
    with torch.no_grad(), defer_finalize():
        loss_n, l1_n = face_loss(image.cuda(), gt.cuda(), face.cuda(), hair.cuda(), mouth.cuda(), bg.cuda(),
                                 alpha=alpha.cuda(), attn=attn.cuda(), lips_rect=lips.cuda(), extra=extra.cuda(),
                                 hair_mask_iter=hair_mask_iter)
    assert float(loss_n) == float(loss_h) and float(l1_n) == float(l1_h)


def test_densify_stats_matches_torch():
    from instag_amd.glue import densify_stats
    g = torch.Generator().manual_seed(5)
    N = 4099
    vs = torch.randn(N, 3, generator=g)
    radii = torch.randint(-1, 40, (N,), generator=g, dtype=torch.int32)
    mr, acc, den = torch.rand(N, generator=g) * 30, torch.rand(N, 1, generator=g), torch.rand(N, 1, generator=g).round()
    vis = radii > 0
    mr_r = torch.where(vis, torch.max(mr, radii.float()), mr)
    acc_r = acc + vs[:, :2].norm(dim=-1, keepdim=True) * vis[:, None]
    den_r = den + vis[:, None].float()
    mr_h, acc_h, den_h = mr.cuda(), acc.cuda(), den.cuda()
    densify_stats(vs.cuda(), radii.cuda(), mr_h, acc_h, den_h)
    assert torch.equal(mr_h.cpu(), mr_r) and torch.equal(den_h.cpu(), den_r)
    _close(acc_h, acc_r, "grad_accum", tol=1e-6)


@pytest.mark.parametrize("n", [1, 2, 3, 257, 5000])
def test_dist_cuda2_matches_bruteforce(n):
    """simple_knn._C.distCUDA2: mean squared distance to the 3 nearest other points (fp64 brute force on the CPU)."""
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(n)
    pts = torch.rand(n, 3, generator=g) * 0.2 - 0.1
    if n >= 257:
        pts[5] = pts[9]                      # duplicate points: distance 0 to each other
    d = torch.cdist(pts.double(), pts.double()) ** 2
    d.fill_diagonal_(float("inf"))
    k = min(3, n - 1)
    ref = d.topk(k, largest=False).values.mean(dim=1) if k > 0 else torch.zeros(n, dtype=torch.float64)
    got = distCUDA2(pts.cuda()).cpu().double()
    assert got.shape == (n,)
    assert float((got - ref).abs().max()) <= 1e-6 * max(float(ref.abs().max()), 1e-12) + 1e-12


def test_create_from_pcd_uses_knn_scales():
    from instag_amd.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(0)
    pts, col = torch.rand(500, 3, generator=g) * 0.2 - 0.1, torch.rand(500, 3, generator=g)
    gm = GaussianModel(1).create_from_pcd(pts, col, spatial_lr_scale=1.0)
    d = torch.cdist(pts.double(), pts.double()) ** 2
    d.fill_diagonal_(float("inf"))
    ref = torch.log(torch.sqrt(d.topk(3, largest=False).values.mean(1).clamp_min(1e-7)))
    assert float((gm._scaling[:, 0].detach().cpu().double() - ref).abs().max()) < 1e-5
    assert gm.get_xyz.shape == (500, 3) and gm.get_features.shape == (500, 4, 3)
    assert float((gm.get_features[:, 0].detach().cpu() * 0.28209479177387814 + 0.5 - col).abs().max()) < 1e-6


def test_mouth_branch_gpu_matches_cpu_modules():
    """MouthMotionNetwork on the device (HIP grid encoders + fused MLPs) == the same module on the CPU with the oracle
    grid encoder; render_motion_mouth_con runs end to end and reaches every mouth parameter with a gradient."""
    import copy
    from types import SimpleNamespace
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
    from instag_amd.renderer import render_motion_mouth_con
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import make_frame
    from oracle.grid_torch import GridEncoder as CpuGrid
    torch.manual_seed(5)
    args = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
    cpu = MouthMotionNetwork(args=args, encoder_cls=CpuGrid)
    with torch.no_grad():
        for n_, p_ in cpu.named_parameters():
            if n_.endswith("embeddings"):
                p_.copy_(torch.randn(p_.shape) * 0.1)
    dev = MouthMotionNetwork(args=args).cuda()
    dev.load_state_dict(cpu.state_dict())
    x = torch.rand(4000, 3) * 0.2 - 0.1
    a, move = torch.randn(8, 29, 16), torch.tensor([[0.3, -0.2, 0.5]])
    oc, od = cpu(x, a, move), dev(x.cuda(), a.cuda(), move.cuda())
    for k in oc:
        _close(od[k], oc[k], k, tol=2e-5)
    # (the outputs go before the next forward: alive, they would keep this forward's autograd graph -- and the gradient
    # accumulators of `dev`'s parameters, bound to THIS call's streams -- into a backward whose audio branch runs on a
    # side stream; PyTorch warns about exactly that, and tests/conftest.py turns the warning into an error)
    del od, oc

    # end to end: face model + face field feed the jaw-movement feature, mouth model is rendered
    size = 96
    frame = make_frame(toy_cameras(size)[0].to("cuda"), synthetic_frame(size, 0, "cuda"))
    face_args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    pc_face = GaussianModel(1).create_random(1500, "cuda", seed=1)
    face_net = MotionNetwork(args=face_args).cuda()
    pc = GaussianModel(1, PersonalizedMotionNetwork(args=args).cuda()).create_random(800, "cuda", seed=2)
    bg = torch.tensor([0.0, 0.0, 0.0], device="cuda")
    pkg = render_motion_mouth_con(frame, pc, dev, pc_face, face_net, None, bg, align=True, k=10)
    assert pkg["render"].shape == (3, size, size) and bool(torch.isfinite(pkg["render"]).all())
    (pkg["render"].sum() + pkg["alpha"].sum()).backward()
    for n_, p_ in list(dev.named_parameters()):
        if not n_.startswith("aud_ch_att_net"):          # unused by the reference's forward as well
            assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n_
    assert pc._xyz.grad is not None and float(pc._xyz.grad.abs().max()) > 0
    assert all(p_.grad is None for p_ in face_net.parameters())          # the movement feature carries no gradient

    # fuse stage: face over mouth over the scene background; with a zero bg_color it is plain "over" compositing
    # (the first render's package goes first: its autograd graph would keep the mouth parameters' gradient accumulators
    # bound to this stream, while render_fuse may run the mouth pass -- and its backward -- on a stream of its own)
    del pkg
    from instag_amd.renderer import render_fuse
    pc_face.neural_motion_grid = PersonalizedMotionNetwork(args=face_args).cuda()
    scene_bg = torch.rand(3, size, size, device="cuda")
    out = render_fuse(frame, pc_face, face_net, pc, dev, None, bg, scene_background=scene_bg)
    fa, ma = out["face"]["alpha"], out["mouth"]["alpha"]
    expect = out["face"]["render"] + (out["mouth"]["render"] + scene_bg * (1 - ma)) * (1 - fa)
    assert float((out["image"] - expect).abs().max()) <= 1e-6
    out["image"].sum().backward()
    assert any(p_.grad is not None for p_ in face_net.parameters())


def test_fuse_renderer_graph_matches_eager():
    """Inference path: the captured forward-only frame == the eager one, frame after frame."""
    from types import SimpleNamespace
    from instag_amd import diff_gauss
    from instag_amd.gaussian_model import GaussianModel
    from instag_amd.infer import FuseRenderer
    from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import make_frame
    torch.manual_seed(9)
    size = 96
    face_args = SimpleNamespace(audio_extractor="deepspeech", type="face")
    mouth_args = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
    pc = GaussianModel(1, PersonalizedMotionNetwork(args=face_args).cuda()).create_random(3000, "cuda", seed=1)
    pcm = GaussianModel(1, PersonalizedMotionNetwork(args=mouth_args).cuda()).create_random(800, "cuda", seed=2)
    net, netm = MotionNetwork(args=face_args).cuda(), MouthMotionNetwork(args=mouth_args).cuda()
    cams = toy_cameras(size)
    frames = [make_frame(cams[i].to("cuda"), synthetic_frame(size, i, "cuda")) for i in range(3)]
    bg = torch.zeros(3, device="cuda")
    r = FuseRenderer(pc, net, pcm, netm, bg)
    try:
        eager = [r.render(f).clone() for f in frames]
        r.enable_graph(frames[0])
        for f, e in zip(frames, eager):
            got = r.render(f)
            assert not r.check_overflow()
            assert torch.equal(got, e)
        # streaming: three frames per replay, each on a lane of its own inside ONE graph (synthesize_fuse.py:34-92
        # renders frame after frame; SURVEY 8(f)4 "batch frames per launch"); a group of five = one full + one padded
        r.close()
        r.enable_graph(frames[0], frames_per_replay=3)
        got = r.render_batch(frames)
        assert got.shape[0] == 3 and not r.check_overflow()
        for k, e in enumerate(eager):
            assert torch.equal(got[k], e), k
        five = r.render_batch([frames[2], frames[0], frames[1], frames[1], frames[0]])
        for k, j in enumerate((2, 0, 1, 1, 0)):
            assert torch.equal(five[k], eager[j]), k
        assert torch.equal(r.render(frames[1]), eager[1])
    finally:
        r.close()
        diff_gauss.set_capacity_plan(None)


def test_deform_activate_with_fused_regulariser():
    """deform_activate(reg_weight): partial sums == weight * the five mean-|.| terms, gradients == deform + regulariser."""
    from instag_amd.glue import deform_activate
    g = torch.Generator().manual_seed(3)
    N = 3001
    vals = dict(xyz=torch.randn(N, 3, generator=g), scaling=torch.randn(N, 3, generator=g),
                rotation=torch.randn(N, 4, generator=g), opacity=torch.randn(N, 1, generator=g),
                h=torch.randn(N, 11, generator=g), p=torch.randn(N, 6, generator=g))
    wm, ws, wr, wo = (torch.randn(N, k, generator=g) for k in (3, 3, 4, 1))
    w = 0.37

    def ref(t):
        ps = torch.tanh(t["p"][:, 3:] / 5) * 0.25 + 1
        means = t["xyz"] + (t["h"][:, :3] * 1e-2) * ps
        scales = torch.nn.functional.softplus(t["scaling"] + t["h"][:, 8:11])
        rots = torch.nn.functional.normalize(t["rotation"] + t["h"][:, 3:7])
        op = torch.sigmoid(t["opacity"])
        # d_xyz as the reference's dictionary holds it after `d_xyz *= p_scale` (gaussian_renderer/__init__.py:217)
        reg = (((t["h"][:, :3] * 1e-2) * ps).abs().mean() + t["h"][:, 3:7].abs().mean() + t["h"][:, 7:8].abs().mean()
               + t["h"][:, 8:11].abs().mean() + (t["p"][:, :3] * 1e-2).abs().mean())
        return means, scales, rots, op, w * reg

    td = {k: v.double().requires_grad_(True) for k, v in vals.items()}
    m, s, r, o, reg = ref(td)
    ((m * wm.double()).sum() + (s * ws.double()).sum() + (r * wr.double()).sum() + (o * wo.double()).sum()
     + 2.5 * reg).backward()
    th = {k: v.cuda().requires_grad_(True) for k, v in vals.items()}
    mh, sh, rh, oh, parts = deform_activate(th["xyz"], th["scaling"], th["rotation"], th["opacity"], th["h"], th["p"],
                                            reg_weight=w)
    ((mh * wm.cuda()).sum() + (sh * ws.cuda()).sum() + (rh * wr.cuda()).sum() + (oh * wo.cuda()).sum()
     + 2.5 * parts.sum()).backward()
    assert abs(float(parts.sum()) - float(reg)) <= 1e-6 * max(1.0, abs(float(reg)))
    for k in vals:
        _close(th[k].grad, td[k].grad, "d_" + k, tol=2e-5)


def test_multi_tensor_adam_state_dict_roundtrip():
    from instag_amd.optim import MultiTensorAdam
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        return [torch.nn.Parameter(torch.randn(300, 7, device="cuda")), torch.nn.Parameter(torch.randn(41, device="cuda"))]
    grads = [[torch.randn(300, 7, device="cuda"), torch.randn(41, device="cuda")] for _ in range(4)]
    pa = make()
    oa = MultiTensorAdam([{"params": [pa[0]], "lr": 1e-2}, {"params": [pa[1]], "lr": 3e-3}], eps=1e-15)
    for g in grads[:2]:
        for p, gg in zip(pa, g):
            p.grad = gg.clone()
        oa.step()
    pb = make()
    with torch.no_grad():
        for p, q in zip(pb, pa):
            p.copy_(q)
    ob = MultiTensorAdam([{"params": [pb[0]], "lr": 1.0}, {"params": [pb[1]], "lr": 1.0}], eps=1e-15)
    ob.load_state_dict(oa.state_dict())
    assert [g["lr"] for g in ob.param_groups] == [1e-2, 3e-3]
    for g in grads[2:]:
        for ps, o in ((pa, oa), (pb, ob)):
            for p, gg in zip(ps, g):
                p.grad = gg.clone()
            o.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p, q)


@pytest.mark.parametrize("use_depth,size", [(True, (96, 80)), (False, (70, 53)), (True, (33, 517))])
def test_geometry_prior_matches_torch(use_depth, size):
    """Fused normal / depth prior (csrc/prior.hip) == the plain-torch statement of train_face.py:458-504 with
    utils/loss_utils.py:17-20, value and gradients (fp64 reference)."""
    from instag_amd.losses import geometry_prior_loss, geometry_prior_loss_torch
    H, W = size
    g = torch.Generator().manual_seed(H * 1000 + W)
    normal = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    depth = 0.8 + 0.1 * torch.rand(1, H, W, generator=g)
    gt_normal = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    gt_depth = 0.5 + 0.3 * torch.rand(H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    face = ((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.35 * min(H, W)) ** 2
    hair = (yy < 0.3 * H) & ~face
    mouth = ((yy - 0.65 * H) ** 2 + (xx - W / 2) ** 2) < (0.08 * min(H, W)) ** 2

    n64 = normal.double().requires_grad_(True)
    d64 = depth.double().requires_grad_(True)
    want = geometry_prior_loss_torch(n64, d64, gt_normal.double(), gt_depth.double(), face, hair, mouth, use_depth)
    (want * 3.0).backward()

    nd = normal.cuda().requires_grad_(True)
    dd = depth.cuda().requires_grad_(True)
    got = geometry_prior_loss(nd, dd, gt_normal.cuda(), gt_depth.cuda(), face.cuda(), hair.cuda(), mouth.cuda(),
                              use_depth)
    (got * 3.0).backward()
    assert abs(float(got) - float(want)) <= 2e-5 * max(1.0, abs(float(want))), (float(got), float(want))
    n_scale = float(n64.grad.abs().max())
    assert float((nd.grad.cpu().double() - n64.grad).abs().max()) <= 1e-4 * n_scale
    if use_depth:
        scale = float(d64.grad.abs().max())
        err = float((dd.grad.cpu().double() - d64.grad).abs().max())
        assert err <= 2e-3 * scale, (err, scale)
    else:
        assert dd.grad is None


@pytest.mark.parametrize("hidden,with_amb", [(64, True), (32, True), (64, False)])
def test_glue_sigma_backward_kernel_matches_the_two_operators(hidden, with_amb):
    """glue + sigma_net as one autograd node (sigma_net's backward with the glue backward as its epilogue, the [N,74]
    input gradient never stored) == motion_glue followed by fused_mlp: outputs and every gradient."""
    from instag_amd import glue
    from instag_amd.mlp import fused_mlp
    from instag_amd.motion_net import MLP
    torch.manual_seed(5)
    N = 5003
    net = MLP(74, 11, hidden, 3).cuda()
    assert glue.glue_sigma_supported(torch.empty(N, 36, device="cuda"), torch.empty(N, 32, device="cuda"),
                                     torch.empty(N, 6, device="cuda"), net)
    base = [torch.randn(N, 36), torch.randn(N, 32), torch.randn(N, 6), torch.randn(32), torch.randn(6)]
    gy, gamb = torch.randn(N, 11).cuda(), torch.randn(N, 3).cuda()
    gamb[:, 2] = 0

    def run(fused):
        ins = [t.clone().cuda().requires_grad_(True) for t in base]
        for p in net.parameters():
            p.grad = None
        if fused:
            y, amb = glue.glue_sigma(*ins, net)
        else:
            h_in, amb = glue.motion_glue(*ins)
            y = fused_mlp(h_in, [l.weight for l in net.net])
        loss = (y * gy).sum() + ((amb * gamb).sum() if with_amb else 0.0)
        loss.backward()
        return [y.detach(), amb.detach()] + [t.grad for t in ins] + [p.grad.clone() for p in net.parameters()]

    ref, got = run(False), run(True)
    names = ["y", "amb", "d_enc_x", "d_aud", "d_eye_pre", "d_enc_a", "d_enc_e", "dW1", "dW2", "dW3"]
    for n_, r, g in zip(names, ref, got):
        scale = float(r.abs().max())
        assert float((r - g).abs().max()) <= 2e-5 * scale + 1e-6, (n_, float((r - g).abs().max()), scale)


@pytest.mark.parametrize("hidden", [64, 32])
@pytest.mark.parametrize("in_block", [True, False])
def test_glue_sigma_unstored_input_rows_give_the_same_bits(hidden, in_block, monkeypatch):
    """sigma_net's input rows not written by the forward, assembled again by the first layer's weight gradient
    (instag_linear_weight_grad_batched_glue; outside a deferred block by torch) == rows stored by the forward: the same
    products in the same order, so every output and gradient bit for bit."""
    from instag_amd import glue
    from instag_amd.deferred import deferred_grads
    from instag_amd.motion_net import MLP
    torch.manual_seed(6)
    N = 7001
    net = MLP(74, 11, hidden, 3).cuda()
    base = [torch.randn(N, 36), torch.randn(N, 32), torch.randn(N, 6), torch.randn(32), torch.randn(6)]
    gy, gamb = torch.randn(N, 11).cuda(), torch.randn(N, 3).cuda()

    def run(virtual):
        monkeypatch.setattr(glue, "VIRTUAL_INPUT", virtual)
        ins = [t.clone().cuda().requires_grad_(True) for t in base]
        for p in net.parameters():
            p.grad = None
        y, amb = glue.glue_sigma(*ins, net)
        loss = (y * gy).sum() + (amb * gamb).sum()
        if in_block:
            with deferred_grads("cuda"):
                loss.backward()
        else:
            loss.backward()
        return [y.detach(), amb.detach()] + [t.grad for t in ins] + [p.grad.clone() for p in net.parameters()]

    stored, virtual = run(False), run(True)
    for i, (a, b) in enumerate(zip(stored, virtual)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("n,k", [(1, 1), (37, 50), (4096, 50), (4097, 10), (100000, 50), (400003, 64)])
def test_extreme_values_match_topk(n, k):
    """csrc/select.hip (the jaw-movement feature's two selections) == torch.topk values, largest and smallest."""
    from instag_amd.renderer import _extreme_values
    g = torch.Generator().manual_seed(n)
    v = torch.randn(n, generator=g)
    if n > 100:
        v[7] = v[11]                                 # ties
    top, bottom = _extreme_values(v.cuda(), k)
    kk = min(k, n)
    assert torch.equal(top.cpu(), v.topk(kk, largest=True, sorted=True).values)
    assert torch.equal(bottom.cpu(), v.topk(kk, largest=False, sorted=True).values)


@pytest.mark.parametrize("kind", ["constant", "many-per-chunk", "short-tail", "all-negative"])
def test_extreme_values_adversarial_inputs(kind):
    """The two-launch selection (512-value chunks, then one workgroup over the candidates >= the best chunk's k-th value)
    on inputs that defeat its pruning: every value tied at the threshold, 63 large values in every chunk (all of them pass
    the threshold), a last chunk shorter than k, and a vector without positive values."""
    from instag_amd.renderer import _extreme_values
    g = torch.Generator().manual_seed(3)
    n, k = 100000, 50
    if kind == "constant":
        v = torch.full((n,), 0.25)
    elif kind == "many-per-chunk":
        v = torch.randn(n, generator=g) * 1e-3
        idx = (torch.arange(n) % 512) < 49                       # 49 large values in every 512-value chunk
        v[idx] = 10.0 + torch.rand(int(idx.sum()), generator=g)
        v[~idx & ((torch.arange(n) % 512) < 98)] -= 10.0         # ... and 49 small ones
    elif kind == "short-tail":
        n = 512 * 9 + 7                                           # the last chunk holds 7 values
        v = torch.randn(n, generator=g)
        v[-7:] = torch.tensor([9.0, -9.0, 8.0, -8.0, 7.0, -7.0, 6.0])
    else:
        v = -torch.rand(n, generator=g) - 1.0
    top, bottom = _extreme_values(v.cuda(), k)
    assert torch.equal(top.cpu(), v.topk(k, largest=True, sorted=True).values)
    assert torch.equal(bottom.cpu(), v.topk(k, largest=False, sorted=True).values)


def test_mouth_activate_matches_plain_torch():
    """glue.mouth_activate = the mouth render's elementwise tail (gaussian_renderer/__init__.py:404-420 with
    scene/motion_net.py:446-452) -- values and all six gradients against the torch expressions."""
    from instag_amd.glue import mouth_activate
    torch.manual_seed(3)
    n = 3001
    mk = lambda *s: torch.randn(*s, device="cuda").requires_grad_(True)
    xyz, scaling, rotation, opacity, h, hs = mk(n, 3), mk(n, 3), mk(n, 4), mk(n, 1), mk(n, 7), mk(n, 1)
    w = [torch.randn(n, c, device="cuda") for c in (3, 3, 4, 1)]
    scale = (1e-2 / 5, 1e-2, 1e-2 / 5)

    def plain():
        d = (h[..., :3] * torch.tensor(scale, device="cuda")) * torch.sigmoid(hs) * 2
        return (xyz + d, torch.nn.functional.softplus(scaling), torch.nn.functional.normalize(rotation),
                torch.sigmoid(opacity))

    leaves = (xyz, scaling, rotation, opacity, h, hs)
    grads = []
    outs = []
    for fn in (plain, lambda: mouth_activate(xyz, scaling, rotation, opacity, h, hs, scale)):
        for t in leaves:
            t.grad = None
        o = fn()
        sum((a * b).sum() for a, b in zip(o, w)).backward()
        outs.append([t.detach().clone() for t in o])
        grads.append([t.grad.clone() for t in leaves])
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
    assert float(grads[1][4][:, 3:].abs().max()) == 0.0          # the predicted rotation is not applied


def test_abs_mean_partials_match_torch():
    from instag_amd.glue import abs_mean_partials
    torch.manual_seed(5)
    p = torch.randn(20011, 6, device="cuda")
    p[7, 1] = 0.0                                                   # sign(0) = 0
    p.requires_grad_(True)
    want = (p[:, :3] * 1e-2).abs().mean()
    (want * 3.0).backward()
    g_want = p.grad.clone()
    p.grad = None
    got = abs_mean_partials(p, 3, 1e-2).sum()
    (got * 3.0).backward()
    assert abs(float(got) - float(want)) <= 2e-6 * float(want)
    assert float((p.grad - g_want).abs().max()) <= 1e-12 + 1e-6 * float(g_want.abs().max())
    assert float(p.grad[:, 3:].abs().max()) == 0.0


def test_mouth_glue_matches_cat_and_repeat():
    from instag_amd.glue import mouth_glue
    torch.manual_seed(6)
    n = 20003
    enc_x = torch.randn(n, 36, device="cuda").requires_grad_(True)
    enc_a = torch.randn(1, 32, device="cuda").requires_grad_(True)
    move = torch.randn(1, 3, device="cuda")
    w1, w2 = torch.randn(n, 71, device="cuda"), torch.randn(n, 39, device="cuda")

    def plain():
        m = move.repeat(n, 1)
        return torch.cat([enc_x, enc_a.repeat(n, 1), m], -1), torch.cat([enc_x, m], -1)

    res = []
    for fn in (plain, lambda: mouth_glue(enc_x, enc_a, move)):
        enc_x.grad = enc_a.grad = None
        a, b = fn()
        ((a * w1).sum() + (b * w2).sum()).backward()
        res.append((a.detach().clone(), b.detach().clone(), enc_x.grad.clone(), enc_a.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float((res[0][2] - res[1][2]).abs().max()) <= 1e-6 * float(res[0][2].abs().max())
    assert float((res[0][3] - res[1][3]).abs().max()) <= 2e-5 * float(res[0][3].abs().max())


@pytest.mark.parametrize("n", [900, 100000])
def test_jaw_feature_matches_topk(n):
    from instag_amd.renderer import _jaw_feature
    torch.manual_seed(8)
    h = torch.randn(n, 11, device="cuda")
    dy = h[:, 1] * 1e-2
    for k in (1, 10, 50):
        mx, mn = dy.topk(k).values[-1], dy.topk(k, largest=False).values[-1]
        want = torch.stack([mx, mn, mx - mn]).reshape(1, 3) * 1e2
        assert torch.equal(_jaw_feature(h, 1, 1e-2, k), want)
        kd = torch.tensor([k], dtype=torch.int64, device="cuda")
        assert torch.equal(_jaw_feature(h, 1, 1e-2, kd), want)


@pytest.mark.parametrize("with_scene", [False, True])
def test_fuse_compose_matches_plain_torch(with_scene):
    from instag_amd.glue import fuse_compose
    torch.manual_seed(9)
    H, W = 67, 131
    mk = lambda *s: torch.rand(*s, device="cuda").requires_grad_(True)
    face, mouth, af, am = mk(3, H, W), mk(3, H, W), mk(1, H, W), mk(1, H, W)
    bg = torch.tensor([0.0, 1.0, 0.0], device="cuda")
    scene = torch.rand(3, H, W, device="cuda") if with_scene else None
    w1, w2 = torch.randn(3, H, W, device="cuda"), torch.randn(3, H, W, device="cuda")

    def plain():
        bg3 = bg[:, None, None]
        sb = scene if scene is not None else torch.zeros_like(face)
        mi = mouth - bg3 * (1.0 - am) + sb * (1.0 - am)
        return face - bg3 * (1.0 - af) + mi * (1.0 - af), mi

    res = []
    for fn in (plain, lambda: fuse_compose(face, af, mouth, am, bg, scene)):
        for t in (face, mouth, af, am):
            t.grad = None
        img, mi = fn()
        ((img * w1).sum() + (mi * w2).sum()).backward()
        res.append([img.detach().clone(), mi.detach().clone()] + [t.grad.clone() for t in (face, mouth, af, am)])
    for a, b in zip(res[0], res[1]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


def test_plain_loss_fused_matches_torch():
    """losses.plain_loss_fused (the fuse stage's whole-frame L1 + DSSIM, train_fuse_con.py:176-181) against the torch
    formulation pinned by tests/golden/g3_losses.npz: values and the image gradient."""
    from instag_amd.losses import l1_loss, plain_loss_fused, ssim
    torch.manual_seed(10)
    img = torch.rand(3, 96, 160, device="cuda").requires_grad_(True)
    gt = torch.rand(3, 96, 160, device="cuda")
    want_l1 = l1_loss(img, gt)
    want = want_l1 + 0.2 * (1.0 - ssim(img, gt))
    want.backward()
    g_want = img.grad.clone()
    img.grad = None
    loss, l1 = plain_loss_fused(img, gt, 0.2)
    loss.backward()
    assert abs(float(loss) - float(want)) <= 2e-6 * float(want) and abs(float(l1) - float(want_l1)) <= 2e-6 * float(want_l1)
    assert float((img.grad - g_want).abs().max()) <= 2e-5 * float(g_want.abs().max())
