"""GPU parity: HIP grid / SH encoders (through the C ABI) vs the CPU oracles and golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FACE = dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=16, log2_hashmap_size=17,
            desired_resolution=256 * 0.15)
MOUTH = dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=64, log2_hashmap_size=17,
             desired_resolution=384 * 0.15)
NGP3D = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=14,
             desired_resolution=512)
TILED = dict(input_dim=3, num_levels=4, level_dim=4, base_resolution=8, log2_hashmap_size=12,
             desired_resolution=64, gridtype="tiled", align_corners=True, interpolation="smoothstep")


def _pair(cfg, seed=0):
    from instag_amd.gridencoder import GridEncoder
    from oracle.grid_ref import GridEncoderRef
    ref_cfg = dict(cfg)
    ref = GridEncoderRef(seed=seed, **ref_cfg)
    enc = GridEncoder(**cfg).cuda()
    assert np.array_equal(enc.offsets.cpu().numpy(), ref.offsets)
    # use O(1) embeddings so errors are visible (the 1e-4 init would hide them)
    rng = np.random.default_rng(seed + 1)
    ref.embeddings = rng.standard_normal(ref.embeddings.shape).astype(np.float32)
    with torch.no_grad():
        enc.embeddings.copy_(torch.from_numpy(ref.embeddings))
    return enc, ref


@pytest.mark.parametrize("cfg", [FACE, MOUTH, NGP3D, TILED], ids=["face", "mouth", "ngp3d-hash", "tiled-smooth"])
def test_grid_forward_backward(cfg):
    from oracle import grid_ref
    enc, ref = _pair(cfg)
    D = cfg["input_dim"]
    g = torch.Generator().manual_seed(3)
    x = torch.rand(5000, D, generator=g) * 2.2 - 1.1          # some points out of range -> zeros
    x[0] = 1.0
    x[1] = -1.0
    xh = x.cuda().requires_grad_(True)
    out = enc(xh, bound=1)
    out_ref, dy_dx_ref = ref.forward(x.numpy(), bound=1, calc_grad_inputs=True)
    assert out.shape == (5000, enc.output_dim)
    # O(1) random embeddings: one fp32 ulp of pos = x*scale+0.5 (FMA vs two roundings, exp2f ulp) is
    # ~6e-8*resolution and neighbouring vertices differ by O(1..4) -> tolerance grows with the finest level
    finest = cfg["desired_resolution"] if cfg["desired_resolution"] > cfg["base_resolution"] else cfg["base_resolution"]
    assert float((out.detach().cpu() - torch.from_numpy(out_ref)).abs().max()) <= 1.5e-6 * finest
    w = torch.randn(out.shape, generator=g)
    (out * w.cuda()).sum().backward()
    L, C = cfg["num_levels"], cfg["level_dim"]
    grad_lbc = w.view(5000, L, C).permute(1, 0, 2).contiguous().numpy()
    x01 = ((x.numpy().astype(np.float32) + np.float32(1)) / np.float32(2))
    ge, gi = grid_ref.grid_encode_backward(grad_lbc, x01, ref.embeddings, ref.offsets, np.log2(ref.per_level_scale),
                                           ref.base_resolution, dy_dx_ref, ref.gridtype_id, ref.align_corners,
                                           ref.interp_id)
    ge_h = enc.embeddings.grad.cpu().numpy()
    assert np.abs(ge_h - ge).max() <= 2e-4 * max(1.0, np.abs(ge).max())
    gi_h = xh.grad.cpu().numpy() * 2.0        # d/dx of (x+1)/2
    # dy_dx is piecewise constant per cell (and unrelated across cells of a hashed level): a point whose
    # pos = x*scale+0.5 lands within an ulp of a cell border may pick the other cell -> allow rare outliers
    rel = np.abs(gi_h - gi) / max(1.0, np.abs(gi).max())
    assert np.quantile(rel, 0.995) <= 2e-4 and (rel > 2e-4).mean() < 5e-3


def test_grid_known_answers():
    """Vertex value = embedding, bilinear midpoint = mean of 4 corners, out-of-range -> 0 (gridencoder.cu:111-191)."""
    from instag_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=2, num_levels=2, level_dim=1, base_resolution=16, log2_hashmap_size=17,
                      desired_resolution=32, align_corners=True).cuda()
    with torch.no_grad():
        enc.embeddings.copy_(torch.arange(enc.embeddings.numel(), dtype=torch.float32).view(-1, 1))
    # align_corners: scale = 15, resolution 16, stride 16: vertex (i, j) -> index i + 16 j
    v = torch.tensor([[3 / 15, 7 / 15], [3.5 / 15, 7.5 / 15], [1.5, 0.2], [-0.1, 0.5]]) * 2 - 1
    out = enc(v.cuda(), bound=1).cpu()[:, 0]          # level 0
    assert abs(out[0].item() - (3 + 16 * 7)) < 1e-3
    assert abs(out[1].item() - np.mean([3 + 16 * 7, 4 + 16 * 7, 3 + 16 * 8, 4 + 16 * 8])) < 1e-3
    assert out[2].item() == 0.0 and out[3].item() == 0.0


@pytest.mark.parametrize("cfg", [FACE, NGP3D, TILED], ids=["face-dense", "ngp3d-hash", "tiled-wrap"])
def test_grid_total_variation_matches_oracle(cfg):
    """grad_total_variation (gridencoder/grid.py:165-185): values against the numpy restatement of
    gridencoder.cu:506-610 -- levels walked per table entry (dense) and per sample in fixed point (hashed / wrapped);
    repeated calls are bitwise reproducible (no float atomics); error behaviour as the reference."""
    from oracle import grid_ref
    enc, ref = _pair(cfg, seed=2)
    D = cfg["input_dim"]
    g = torch.Generator().manual_seed(11)
    x = torch.rand(20000, D, generator=g) * 2.1 - 1.05            # a few samples out of range: ignored
    with pytest.raises(ValueError):
        enc.grad_total_variation(1e-3, x.cuda())
    base = torch.randn(enc.embeddings.shape, generator=g)
    results = []
    for _ in range(2):
        enc.embeddings.grad = base.clone().cuda()
        enc.grad_total_variation(weight=3e-3, inputs=x.cuda(), bound=1)
        results.append(enc.embeddings.grad.clone())
    assert torch.equal(results[0], results[1])
    x01 = ((x.numpy().astype(np.float32) + np.float32(1)) / np.float32(2))
    want = grid_ref.grad_total_variation(x01, ref.embeddings, base.numpy(), ref.offsets, 3e-3,
                                         np.log2(ref.per_level_scale), ref.base_resolution, ref.gridtype_id,
                                         ref.align_corners)
    got = results[0].cpu().numpy()
    added = np.abs(want - base.numpy())
    assert added.max() > 1e-3                                       # the term is not negligible against the bar
    # a sample within an ulp of a cell border may land in the neighbouring vertex: a rare +-1 in that entry's count
    err = np.abs(got - want)
    assert np.quantile(err, 0.999) <= 2e-6 * max(1.0, added.max())
    assert (err > 2e-6 * max(1.0, added.max())).mean() < 2e-3
    untouched = added == 0                                          # entries no sample hit keep their gradient
    assert (got[untouched] != base.numpy()[untouched]).mean() < 2e-3   # (bar the border flips counted above)


def test_backend_modules_serve_the_reference_call_sequence():
    """`_gridencoder` / `_shencoder`: the pybind names and tensor signatures of gridencoder/src/bindings.cpp:5-7 and
    shencoder/src/bindings.cpp, driven exactly as gridencoder/grid.py:27-89 and shencoder/sphere_harmonics.py:14-54
    drive them (caller-allocated outputs, [L,B,C] layout, zero-filled gradient buffers)."""
    import _gridencoder
    import _shencoder
    from oracle import grid_ref, sh_ref
    enc, ref = _pair(NGP3D, seed=5)
    B, D, C, L = 3000, 3, NGP3D["level_dim"], NGP3D["num_levels"]
    S, H = float(np.log2(ref.per_level_scale)), ref.base_resolution
    g = torch.Generator().manual_seed(4)
    inputs = torch.rand(B, D, generator=g).cuda()
    emb, offsets = enc.embeddings.detach(), enc.offsets
    outputs = torch.empty(L, B, C, device="cuda")
    dy_dx = torch.empty(B, L * D * C, device="cuda")
    _gridencoder.grid_encode_forward(inputs, emb, offsets, outputs, B, D, C, L, S, H, dy_dx, 0, False, 0)
    want, want_dy = grid_ref.grid_encode_forward(inputs.cpu().numpy(), ref.embeddings, ref.offsets, S, H, True, 0, False, 0)
    assert np.abs(outputs.cpu().numpy() - want).max() <= 1.5e-6 * 512
    grad = torch.randn(L, B, C, generator=g).cuda()
    grad_emb, grad_in = torch.zeros_like(emb), torch.zeros(B, D, device="cuda")
    _gridencoder.grid_encode_backward(grad, inputs, emb, offsets, grad_emb, B, D, C, L, S, H, dy_dx, grad_in, 0, False, 0)
    ge, gi = grid_ref.grid_encode_backward(grad.cpu().numpy(), inputs.cpu().numpy(), ref.embeddings, ref.offsets, S, H,
                                           want_dy, 0, False, 0)
    assert np.abs(grad_emb.cpu().numpy() - ge).max() <= 2e-4 * max(1.0, np.abs(ge).max())
    rel = np.abs(grad_in.cpu().numpy() - gi) / max(1.0, np.abs(gi).max())
    assert np.quantile(rel, 0.995) <= 2e-4
    tv = torch.zeros_like(emb)
    _gridencoder.grad_total_variation(inputs, emb, tv, offsets, 1e-3, B, D, C, L, S, H, 0, False)
    assert bool(torch.isfinite(tv).all()) and float(tv.abs().max()) > 0
    with pytest.raises(RuntimeError):
        _gridencoder.grid_encode_forward(inputs.cpu(), emb, offsets, outputs, B, D, C, L, S, H, None, 0, False, 0)
    with pytest.raises(RuntimeError):
        _gridencoder.grid_encode_forward(inputs, emb, offsets.long(), outputs, B, D, C, L, S, H, None, 0, False, 0)
    # SH encoder
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).cuda()
    out = torch.empty(B, 16, device="cuda")
    dsh = torch.empty(B, 48, device="cuda")
    _shencoder.sh_encode_forward(dirs, out, B, 3, 4, dsh)
    want_o, want_d = sh_ref.sh_encode_forward(dirs.cpu().numpy(), 4, True)
    assert np.abs(out.cpu().numpy() - want_o).max() <= 2e-6 and np.abs(dsh.cpu().numpy() - want_d).max() <= 2e-5
    gsh = torch.randn(B, 16, generator=g).cuda()
    gin = torch.zeros(B, 3, device="cuda")
    _shencoder.sh_encode_backward(gsh, dirs, B, 3, 4, dsh, gin)
    want_gi = sh_ref.sh_encode_backward(gsh.cpu().numpy(), want_d, 4)
    assert np.abs(gin.cpu().numpy() - want_gi).max() <= 2e-5 * max(1.0, np.abs(want_gi).max())


def test_grid_state_dict_names():
    from instag_amd.gridencoder import GridEncoder
    enc = GridEncoder(**FACE)
    sd = enc.state_dict()
    assert set(sd) == {"embeddings", "offsets"}
    assert sd["embeddings"].shape == (9464, 1) and sd["offsets"].dtype == torch.int32
    with pytest.raises(RuntimeError):
        enc(torch.rand(4, 2))            # CPU tensor: no CPU path


@pytest.mark.parametrize("degree", [1, 2, 4, 8])
def test_sh_encoder_vs_golden(degree, golden_dir):
    from instag_amd.shencoder import SHEncoder
    g = np.load(f"{golden_dir}/g6_sh_encoder.npz")
    x = torch.from_numpy(g["inputs"]).cuda().requires_grad_(True)
    enc = SHEncoder(3, degree)
    out = enc(x)
    C2 = degree * degree
    ref = g["outputs"][:, :C2]
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 5e-6 * scale
    gen = torch.Generator().manual_seed(1)
    w = torch.randn(out.shape, generator=gen)
    (out * w.cuda()).sum().backward()
    gi = np.stack([(g[k][:, :C2] * w.numpy()).sum(1) for k in ("dx", "dy", "dz")], axis=1)
    assert np.abs(x.grad.cpu().numpy() - gi).max() <= 2e-5 * max(1.0, np.abs(gi).max())


def test_sh_encoder_shapes_and_errors():
    from instag_amd.shencoder import SHEncoder
    enc = SHEncoder(3, 4)
    y = enc(torch.rand(2, 5, 3, device="cuda"), size=2)
    assert y.shape == (2, 5, 16)
    assert abs(float(y[0, 0, 0]) - 0.28209479) < 1e-6
    with pytest.raises(AssertionError):
        SHEncoder(3, 9)
    with pytest.raises(AssertionError):
        SHEncoder(2, 4)


def test_fused_l1_ssim_matches_torch_and_golden(golden_dir):
    from instag_amd import losses
    g = np.load(f"{golden_dir}/g3_losses.npz")
    a = torch.from_numpy(g["a"]).cuda().requires_grad_(True)
    b = torch.from_numpy(g["b"]).cuda()
    l1, s = losses.l1_and_ssim(a, b)
    assert abs(float(l1) - float(g["l1"])) < 1e-6 and abs(float(s) - float(g["ssim"])) < 2e-6
    (l1 + 0.2 * (1.0 - s)).backward()
    a2 = torch.from_numpy(g["a"]).double().requires_grad_(True)
    b2 = torch.from_numpy(g["b"]).double()
    (losses.l1_loss(a2, b2) + 0.2 * (1.0 - losses.ssim(a2, b2))).backward()
    assert float((a.grad.cpu().double() - a2.grad).abs().max()) <= 2e-5 * float(a2.grad.abs().max())
    # ragged size (not a multiple of the 16x16 tile) and full 512x512
    for shape in ((3, 75, 131), (3, 512, 512)):
        gen = torch.Generator().manual_seed(shape[1])
        x = torch.rand(shape, generator=gen)
        y = (x + 0.2 * torch.randn(shape, generator=gen)).clamp(0, 1)
        xh = x.cuda().requires_grad_(True)
        l1h, sh = losses.l1_and_ssim(xh, y.cuda())
        (l1h - 3.0 * sh).backward()
        xd = x.double().requires_grad_(True)
        l1d, sd = losses.l1_loss(xd, y.double()), losses.ssim(xd, y.double())
        (l1d - 3.0 * sd).backward()
        assert abs(float(l1h) - float(l1d)) < 1e-6 and abs(float(sh) - float(sd)) < 5e-6
        assert float((xh.grad.cpu().double() - xd.grad).abs().max()) <= 5e-5 * float(xd.grad.abs().max())


def test_tri_plane_encode_matches_three_encoders():
    """Fused tri-plane kernel == cat of the three per-plane GridEncoder calls (forward, d/dxyz, table gradients)."""
    from instag_amd.gridencoder import GridEncoder, tri_plane_encode, tri_plane_supported
    torch.manual_seed(0)
    encs = [GridEncoder(**FACE).cuda() for _ in range(3)]
    with torch.no_grad():
        for e in encs:
            e.embeddings.copy_(torch.randn_like(e.embeddings))
    assert tri_plane_supported(*encs)
    x = (torch.rand(7001, 3) * 0.34 - 0.17)          # a few points outside bound 0.15 -> zeros
    w = torch.randn(7001, 36)
    xa = x.cuda().requires_grad_(True)
    ya = tri_plane_encode(xa, *encs, 0.15)
    (ya * w.cuda()).sum().backward()
    ga = [e.embeddings.grad.clone() for e in encs]
    for e in encs:
        e.embeddings.grad = None
    xb = x.cuda().requires_grad_(True)
    xy, yz, xz = xb[:, :-1], xb[:, 1:], torch.cat([xb[:, :1], xb[:, -1:]], dim=-1)
    yb = torch.cat([encs[0](xy, bound=0.15), encs[1](yz, bound=0.15), encs[2](xz, bound=0.15)], dim=-1)
    (yb * w.cuda()).sum().backward()
    assert ya.shape == (7001, 36)
    assert float((ya - yb).abs().max()) <= 1e-5
    assert float((xa.grad - xb.grad).abs().max()) <= 2e-4 * float(xb.grad.abs().max())
    for g_, e in zip(ga, encs):
        assert float((g_ - e.embeddings.grad).abs().max()) <= 2e-4 * float(e.embeddings.grad.abs().max())


def test_tri_plane_encode_with_shift():
    """tri_plane_encode(xyz, shift=(p, s)) == tri_plane_encode(xyz + s * p[:, :3]); gradients reach xyz and p[:, :3]."""
    from instag_amd.gridencoder import GridEncoder, tri_plane_encode
    torch.manual_seed(4)
    cfg = dict(input_dim=2, num_levels=12, level_dim=1, base_resolution=16, log2_hashmap_size=17,
               desired_resolution=256 * 0.15)
    es = [GridEncoder(**cfg).cuda() for _ in range(3)]
    with torch.no_grad():
        for e in es:
            e.embeddings.copy_(torch.randn_like(e.embeddings) * 0.1)
    N = 4099
    xyz0, p0 = (torch.rand(N, 3) * 0.2 - 0.1).cuda(), torch.randn(N, 6).cuda()
    w = torch.randn(N, 36).cuda()

    def run(fused):
        xyz, p = xyz0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        for e in es:
            e.embeddings.grad = None
        if fused:
            out = tri_plane_encode(xyz, es[0], es[1], es[2], 0.15, shift=p, shift_scale=1e-2)
        else:
            out = tri_plane_encode(torch.add(xyz, p[:, :3], alpha=1e-2), es[0], es[1], es[2], 0.15)
        (out * w).sum().backward()
        return out.detach(), xyz.grad, p.grad, [e.embeddings.grad.clone() for e in es]

    o_r, gx_r, gp_r, ge_r = run(False)
    o_f, gx_f, gp_f, ge_f = run(True)
    assert torch.equal(o_f, o_r) or float((o_f - o_r).abs().max()) <= 1e-7
    assert float((gx_f - gx_r).abs().max()) <= 1e-6 * max(1.0, float(gx_r.abs().max()))
    assert float((gp_f - gp_r).abs().max()) <= 1e-6 * max(1.0, float(gp_r.abs().max()))
    assert float(gp_f[:, 3:].abs().max()) == 0.0
    for a, b in zip(ge_f, ge_r):
        assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))


def test_grid_renderer_constructs_runs_and_stays_out_of_adam():
    """GridRenderer (scene/neural_renderer.py:49-222): constructed with every Gaussian model, never evaluated by
    InsTaG.  forward == its parts composed by hand (3-D hashed encoder checked against the oracle above); its
    parameters sit in the optimizer's groups but, never receiving a gradient, get no Adam state and no work."""
    from instag_amd.gaussian_model import GaussianModel, OptimizationParams
    from instag_amd.neural_renderer import GridRenderer
    from instag_amd.optim import MultiTensorAdam
    nr = GridRenderer(bound=0.3, coord_center=[0.01, -0.02, 0.0]).cuda()
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(4000, 3, generator=g) * 0.5 - 0.25).cuda()
    d = torch.nn.functional.normalize(torch.randn(4000, 3, generator=g), dim=-1).cuda()
    sigma, color = nr(x, d)
    assert sigma.shape == (4000,) and color.shape == (4000, 3)
    assert bool(torch.isfinite(sigma).all()) and float(color.min()) >= -0.001 and float(color.max()) <= 1.001
    enc = nr.encoder_x(x - nr.coord_center, bound=0.3)
    h = nr.sigma_net(enc)
    assert torch.equal(sigma, h[..., 0])
    want = torch.sigmoid(nr.color_net(torch.cat([nr.encoder_dir(d), h[..., 1:]], dim=-1))) * 1.002 - 0.001
    assert torch.equal(color, want)
    (sigma.sum() + color.sum()).backward()
    assert nr.encoder_x.embeddings.grad is not None and float(nr.encoder_x.embeddings.grad.abs().max()) > 0
    gm = GaussianModel(1).create_random(500, "cuda", seed=3)
    gm.training_setup(OptimizationParams)
    assert isinstance(gm.optimizer, MultiTensorAdam)
    for p in gm.per_gaussian_parameters():
        p.grad = torch.randn_like(p) * 1e-3
    gm.optimizer.step()
    torch.cuda.synchronize()
    with_state = [grp["name"] for grp in gm.optimizer.param_groups
                  if any("exp_avg" in gm.optimizer.state.get(p, {}) for p in grp["params"])]
    assert with_state == ["xyz", "f_dc", "f_rest", "identity", "opacity", "scaling", "rotation"]


@pytest.mark.parametrize("cfg", ["face", "mouth"])
def test_tri_plane_kernels_against_the_numpy_oracle(cfg):
    """The fused tri-plane forward / backward kernels (the ones on the train step's path) directly against
    oracle/grid_ref.py, plane by plane: features, table gradients and d/dxyz -- not via the per-plane HIP encoder.
    "face": tables staged in LDS; "mouth": 46,600 entries per plane, read and accumulated in place."""
    FACE = {"face": globals()["FACE"], "mouth": MOUTH}[cfg]
    from instag_amd.gridencoder import GridEncoder, tri_plane_encode
    from oracle import grid_ref
    from oracle.grid_ref import GridEncoderRef
    bound, n = 0.15, 6000
    refs = [GridEncoderRef(seed=s, **FACE) for s in (1, 2, 3)]
    encs = [GridEncoder(**FACE).cuda() for _ in range(3)]
    rng = np.random.default_rng(7)
    for r, e in zip(refs, encs):
        r.embeddings = rng.standard_normal(r.embeddings.shape).astype(np.float32)
        with torch.no_grad():
            e.embeddings.copy_(torch.from_numpy(r.embeddings))
    g = torch.Generator().manual_seed(5)
    x = torch.rand(n, 3, generator=g) * 0.34 - 0.17               # some points outside the bound -> zeros
    w = torch.randn(n, 36, generator=g)
    xh = x.cuda().requires_grad_(True)
    out = tri_plane_encode(xh, *encs, bound)
    (out * w.cuda()).sum().backward()
    cols = [(0, 1), (1, 2), (0, 2)]                               # xy, yz, xz (scene/motion_net.py:244-258)
    L = FACE["num_levels"]
    gi_total = np.zeros((n, 3), dtype=np.float64)
    finest = FACE["desired_resolution"]
    for p, (r, e, c) in enumerate(zip(refs, encs, cols)):
        xp = x[:, list(c)].numpy()
        out_ref, dy_dx = r.forward(xp, bound=bound, calc_grad_inputs=True)
        got = out[:, p * L:(p + 1) * L].detach().cpu().numpy()
        assert np.abs(got - out_ref).max() <= 1.5e-6 * finest, p
        grad_lbc = w[:, p * L:(p + 1) * L].reshape(n, L, 1).permute(1, 0, 2).contiguous().numpy()
        x01 = (xp.astype(np.float32) + np.float32(bound)) / np.float32(2 * bound)
        ge, gi = grid_ref.grid_encode_backward(grad_lbc, x01, r.embeddings, r.offsets, np.log2(r.per_level_scale),
                                               r.base_resolution, dy_dx, r.gridtype_id, r.align_corners, r.interp_id)
        ge_h = e.embeddings.grad.cpu().numpy()
        assert np.abs(ge_h - ge).max() <= 2e-4 * max(1.0, np.abs(ge).max()), p
        gi_total[:, list(c)] += gi / (2 * bound)                  # d/dx of (x + bound) / (2 bound)
    rel = np.abs(xh.grad.cpu().numpy() - gi_total) / max(1.0, np.abs(gi_total).max())
    assert np.quantile(rel, 0.995) <= 2e-4 and (rel > 2e-4).mean() < 5e-3


@pytest.mark.parametrize("cfg", ["face", "mouth"])
def test_tri_plane_passthrough_sums_the_other_consumers_gradients(cfg):
    """gridencoder.passthrough: the position (and the shift) handed on by the encode carry their other consumers'
    gradients back into the encoder's backward kernel (dxyz_add / dshift_add) -- same sums as autograd's own adds."""
    FACE = {"face": globals()["FACE"], "mouth": MOUTH}[cfg]
    from instag_amd import gridencoder as ge
    from instag_amd.gridencoder import GridEncoder, tri_plane_encode
    torch.manual_seed(1)
    encs = [GridEncoder(**FACE).cuda() for _ in range(3)]
    with torch.no_grad():
        for e in encs:
            e.embeddings.copy_(torch.randn_like(e.embeddings))
    n = 4001
    x0 = (torch.rand(n, 3) * 0.3 - 0.15).cuda()
    p0 = torch.randn(n, 6).cuda()
    w, wx, wp = torch.randn(n, 36).cuda(), torch.randn(n, 3).cuda(), torch.randn(n, 6).cuda()

    def run(route):
        x, p = x0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        for e in encs:
            e.embeddings.grad = None
        if route:
            carrier = {}
            with ge.passthrough(carrier):
                out = tri_plane_encode(x, *encs, 0.15, shift=p, shift_scale=1e-2)
            xr, pr = carrier["xyz"], carrier["shift"]
            assert xr.data_ptr() == x.data_ptr() and pr.data_ptr() == p.data_ptr()
        else:
            out = tri_plane_encode(x, *encs, 0.15, shift=p, shift_scale=1e-2)
            xr, pr = x, p
        ((out * w).sum() + (xr * wx).sum() + (pr * wp).sum()).backward()
        return x.grad, p.grad, [e.embeddings.grad.clone() for e in encs]

    (gx_a, gp_a, gt_a), (gx_b, gp_b, gt_b) = run(False), run(True)
    assert float((gx_a - gx_b).abs().max()) <= 1e-6 * float(gx_a.abs().max())
    assert float((gp_a - gp_b).abs().max()) <= 1e-6 * float(gp_a.abs().max())
    for a, b in zip(gt_a, gt_b):
        assert torch.equal(a, b)                          # fixed-point LDS sums, fixed-order reduce: reproducible
