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
    assert names == ["xyz", "f_dc", "f_rest", "identity", "opacity", "scaling", "rotation"]
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
        p = grp["params"][0]
        assert p.shape[0] == n1 and p.requires_grad
        st = g.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
    assert g.xyz_gradient_accum.shape == (n1, 1) and g.max_radii2D.shape == (n1,)
    # the same seed gives the same result (what keeps DP replicas identical)
    g2 = _model(300)
    for p, q in zip(g2.per_gaussian_parameters(), _model(300).per_gaussian_parameters()):
        assert torch.equal(p, q)


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
        results[rank] = dict(g0=params[0].grad.clone(), g1=params[1].grad.clone(), g2=params[2].grad.clone(),
                             stat=stat.clone(), cnt=cnt.clone(), p1=params[1].detach().clone())
    finally:
        dist.destroy_process_group()


def test_dp_fused_bucket_allreduce_gloo_world2():
    """Gradients become the mean over ranks (== accumulation over the ranks' frames), statistics the sum,
    parameters without gradient are zero-filled consistently; both ranks end up identical."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(world, port, results), nprocs=world, join=True)
    r0, r1 = results[0], results[1]
    for k in ("g0", "g1", "g2", "stat", "cnt"):
        assert torch.equal(r0[k], r1[k]), k
    assert torch.allclose(r0["g0"], torch.full((50, 3), 1.5))            # mean of 1 and 2
    assert torch.allclose(r0["g1"], 2 * r0["p1"] * 1.5)
    assert float(r0["g2"].abs().max()) == 0.0
    assert torch.allclose(r0["stat"], torch.full((50, 1), 3.0)) and torch.allclose(r0["cnt"], torch.full((50, 1), 2.0))


def test_grad_bucket_roundtrip():
    from instag_amd.train import flat_grad_bucket, scatter_grad_bucket
    ps = [torch.nn.Parameter(torch.randn(5, 2)), torch.nn.Parameter(torch.randn(3))]
    ps[0].grad = torch.arange(10.0).view(5, 2)
    b = flat_grad_bucket(ps)
    assert b.shape == (13,) and float(b[10:].abs().max()) == 0
    scatter_grad_bucket(ps, b * 2)
    assert torch.equal(ps[0].grad, torch.arange(10.0).view(5, 2) * 2) and ps[1].grad is not None
