"""GPU parity: fused f32-MFMA MLP operator vs a plain PyTorch fp64 reference, and the full motion
networks (HIP grid encoders + HIP MLPs) vs the golden outputs of the reference's own modules (G5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (dim_in, hidden, dim_out, layers)  -- the MLPs of scene/motion_net.py:234-238,600-604
    pytest.param(74, 64, 11, 3, id="umf-sigma"),
    pytest.param(74, 32, 11, 3, id="pmf-sigma"),
    pytest.param(36, 32, 32, 2, id="aud_ch_att"),
    pytest.param(36, 16, 6, 2, id="eye_att"),
    pytest.param(36, 32, 6, 2, id="align"),
    pytest.param(96, 64, 32, 3, id="max-shape"),
    pytest.param(5, 7, 3, 2, id="odd-shape"),
]


def _ref(x, ws):
    h = x
    for i, w in enumerate(ws):
        h = h @ w.t()
        if i != len(ws) - 1:
            h = torch.relu(h)
    return h


@pytest.mark.parametrize("K0,H,O,NL", SHAPES)
@pytest.mark.parametrize("N", [1, 4999, 65536])
def test_fused_mlp_forward_backward(K0, H, O, NL, N):
    from instag_amd.mlp import fused_mlp
    g = torch.Generator().manual_seed(N + K0)
    dims = [K0] + [H] * (NL - 1) + [O]
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i]) for i in range(NL)]
    x = torch.randn(N, K0, generator=g)
    gy = torch.randn(N, O, generator=g)
    xd = x.double().requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]
    yd = _ref(xd, wd)
    (yd * gy.double()).sum().backward()
    xh = x.cuda().requires_grad_(True)
    wh = [w.cuda().requires_grad_(True) for w in ws]
    yh = fused_mlp(xh, wh)
    assert yh.shape == (N, O)
    (yh * gy.cuda()).sum().backward()

    def close(a, b, name, tol=2e-5, outliers=0.0):
        a, b = a.detach().cpu().double(), b.detach()
        err = (a - b).abs()
        bad = err > tol * max(1.0, float(b.abs().max()))
        assert float(bad.double().mean()) <= outliers, \
            f"{name}: max err {float(err.max())} scale {float(b.abs().max())} bad {int(bad.sum())}"
    close(yh, yd, "y")
    # a hidden unit whose pre-activation is within fp32 rounding of 0 may take the other ReLU branch than the
    # fp64 reference; that changes dx of its row by O(|w|): tolerate a few rows in a million
    close(xh.grad, xd.grad, "dx", outliers=2e-5)
    for i in range(NL):
        close(wh[i].grad, wd[i].grad, f"dW{i + 1}", tol=2e-5 * max(1.0, np.sqrt(N) / 16))


def test_fused_mlp_deterministic_and_no_input_grad():
    from instag_amd.mlp import fused_mlp
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(64, 74, generator=g).cuda().requires_grad_(True),
          torch.randn(64, 64, generator=g).cuda().requires_grad_(True),
          torch.randn(11, 64, generator=g).cuda().requires_grad_(True)]
    x = torch.randn(30000, 74, generator=g).cuda()          # no grad for x -> dx is skipped
    outs = []
    for _ in range(2):
        for w in ws:
            w.grad = None
        fused_mlp(x, ws).square().sum().backward()
        outs.append([w.grad.clone() for w in ws])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_motion_networks_on_gpu_match_reference_golden(golden_dir):
    """UMF / PMF with HIP grid encoders + HIP MLPs == outputs of the reference's modules (CPU, fixture G5)."""
    from argparse import Namespace
    from instag_amd.motion_net import MotionNetwork, PersonalizedMotionNetwork
    g = np.load(f"{golden_dir}/g5_motion_nets.npz")
    x, a, e = (torch.from_numpy(g[k]).cuda() for k in ("x", "a", "e"))
    args = Namespace(audio_extractor="deepspeech", type="face")
    for tag, cls in (("umf", MotionNetwork), ("pmf", PersonalizedMotionNetwork)):
        net = cls(args=args).cuda()
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}.sd.")}
        net.load_state_dict(sd, strict=True)
        out = net(x.clone().requires_grad_(True), a, e)
        for k in [k[len(tag) + 5:] for k in g.files if k.startswith(f"{tag}.out.")]:
            ref = torch.from_numpy(g[f"{tag}.out.{k}"])
            err = float((out[k].detach().cpu() - ref).abs().max())
            assert err <= 2e-6 + 2e-5 * float(ref.abs().max()), (tag, k, err)
        loss = sum(v.square().sum() for v in out.values() if v is not None)
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
        assert net.encoder_xy.embeddings.grad is not None and net.sigma_net.net[0].weight.grad is not None


def test_deferred_batched_weight_grads_match_autograd():
    """deferred_grads(): the MLPs' weight gradients are queued and computed in one batched launch, same values."""
    import torch
    from instag_amd.deferred import deferred_grads
    from instag_amd.mlp import fused_mlp
    torch.manual_seed(0)
    N = 5003
    specs = [(74, 64, 11, 3), (36, 32, 32, 2), (36, 16, 6, 2)]
    xs = [torch.randn(N, k0, device="cuda") for k0, _, _, _ in specs]

    def make():
        torch.manual_seed(1)
        ws = []
        for k0, h, o, nl in specs:
            dims = [k0] + [h] * (nl - 1) + [o]
            ws.append([torch.nn.Parameter(0.2 * torch.randn(dims[i + 1], dims[i], device="cuda")) for i in range(nl)])
        return ws

    def loss(ws):
        return sum((fused_mlp(x, w) ** 2).sum() for x, w in zip(xs, ws))

    ref = make()
    loss(ref).backward()
    got = make()
    with deferred_grads("cuda"):
        loss(got).backward()
    for wr, wg in zip(ref, got):
        for a, b in zip(wr, wg):
            assert b.grad is not None and torch.equal(a.grad, b.grad)
    # a second backward inside a block accumulates like autograd does
    with deferred_grads("cuda"):
        loss(got).backward()
    for wr, wg in zip(ref, got):
        for a, b in zip(wr, wg):
            assert torch.allclose(2 * a.grad, b.grad, rtol=1e-6, atol=0)


def test_shared_input_mlps_match_separate_mlps():
    """(MLP_a(x), MLP_b(x), x) with the three gradients of x summed inside the backward kernels == two fused_mlp calls
    and autograd's own sum (scene/motion_net.py:281-306: both attention MLPs and sigma_net's input read enc_x)."""
    from instag_amd.mlp import fused_mlp, shared_input_mlps
    torch.manual_seed(3)
    N = 5003
    x0 = torch.randn(N, 36, device="cuda")
    wa = [torch.randn(32, 36, device="cuda") * 0.3, torch.randn(32, 32, device="cuda") * 0.3]
    wb = [torch.randn(16, 36, device="cuda") * 0.3, torch.randn(6, 16, device="cuda") * 0.3]
    ga, gb, gx = torch.randn(N, 32, device="cuda"), torch.randn(N, 6, device="cuda"), torch.randn(N, 36, device="cuda")

    def run(shared):
        x = x0.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in wa + wb]
        if shared:
            ya, yb, xo = shared_input_mlps(x, ws[:2], ws[2:])
        else:
            ya, yb, xo = fused_mlp(x, ws[:2]), fused_mlp(x, ws[2:]), x
        ((ya * ga).sum() + (yb * gb).sum() + (xo * gx).sum()).backward()
        return [ya.detach(), yb.detach(), x.grad] + [w.grad for w in ws]

    ref, got = run(False), run(True)
    names = ["ya", "yb", "dx", "dwa1", "dwa2", "dwb1", "dwb2"]
    for n_, r, g in zip(names, ref, got):
        scale = float(r.abs().max())
        assert float((r - g).abs().max()) <= 2e-5 * scale + 1e-6, (n_, float((r - g).abs().max()), scale)
    # only some outputs used: the unused heads' gradients are zeros, the pass-through still arrives
    x = x0.clone().requires_grad_(True)
    ya, yb, xo = shared_input_mlps(x, wa, wb)
    (xo * gx).sum().backward()
    assert torch.equal(x.grad, gx)
