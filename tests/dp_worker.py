"""Worker of test_dp_two_ranks_share_one_gpu: two ranks (gloo) on one card run the data-parallel train step in its
graph-split form (graph A | gradient all-reduce | graph B) and in eager form; replicas must stay identical and the two
forms must agree.  Launched with torch.distributed.run; exits non-zero on failure."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from instag_amd import diff_gauss
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import build_trainer, make_frame
    cams = toy_cameras(128)
    frames = [make_frame(cams[(rank + k * world) % len(cams)].to(dev), synthetic_frame(128, rank + k * world, dev))
              for k in range(3)]
    results = {}
    for mode in ("eager", "graph", "graph-early"):
        tr = build_trainer(3000, dev, seed=1)
        try:
            if mode == "graph":
                tr.enable_graph(frames[0], warmup_steps=2, split_for_allreduce=True)      # the two-graph form
                assert tr._graph.split and tr._graph.graph_a2 is None
            elif mode == "graph-early":
                # the default with several ranks: three graphs, the per-Gaussian bucket is exchanged (asynchronously)
                # beside the motion fields' backward
                tr.enable_graph(frames[0], warmup_steps=2)
                assert tr._graph.split and tr._graph.graph_a2 is not None, "several ranks must use the split forms"
            else:
                for _ in range(4):
                    tr.step(frames[0])
            for i in range(4):
                tr.step(frames[i % 3])
            if mode != "eager":
                assert not tr._graph.check_overflow()
        finally:
            diff_gauss.set_capacity_plan(None)
        vec = torch.cat([tr.g.get_xyz.detach().reshape(-1), tr.g._p["f_dc"].detach().reshape(-1),
                         next(tr.motion_net.sigma_net.parameters()).detach().reshape(-1)]).cpu()
        gathered = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(gathered, vec)
        for other in gathered[1:]:
            assert torch.equal(gathered[0], other), f"{mode}: replicas diverged"
        results[mode] = vec
    diff = float((results["eager"] - results["graph"]).abs().max())
    assert diff <= 1e-5, f"graph-split DP step differs from the eager DP step by {diff}"
    diff_early = float((results["eager"] - results["graph-early"]).abs().max())
    assert diff_early <= 1e-5, f"three-segment DP step differs from the eager DP step by {diff_early}"

    # Only ONE rank overflows its instance capacity: the decision to capture again is collective (all-reduce MAX of the
    # sticky flags at a replay count every rank reaches in the same step), the capture consumes no iteration and runs no
    # collective, so the ranks keep stepping in lock step and the replicas stay identical.
    from instag_amd.train import GraphedStep
    GraphedStep.CHECK_EVERY = 3
    tr = build_trainer(3000, dev, seed=2)
    try:
        tr.enable_graph(frames[0], warmup_steps=1)
        phase = tr._graph.phase
        need = max(tr._graph.plan.needed())
        assert need > 2048, need
        # rank 1 captures again with a capacity far below what its frames need
        tr._drop_graph(keep_mode=True)
        if rank == 1:
            tr._graph_mode["capacity"] = 1024
        tr._recapture(frames[0], phase)
        caps = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(caps, torch.tensor([tr._graph_cache[phase].capacity]))
        assert int(caps[1]) == 1024 < need <= int(caps[0]), caps
        base = tr.recaptures
        it0 = tr.iteration
        for i in range(8):
            tr.step(frames[i % 3])
        assert tr.iteration == it0 + 8
        assert tr.recaptures > base, "the overflow of rank 1 must make EVERY rank capture again"
        assert tr._graph is not None and tr._graph.capacity >= need and not tr._graph.check_overflow()
        its = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(its, torch.tensor([tr.iteration, tr.recaptures]))
        assert torch.equal(its[0], its[1]), its
    finally:
        diff_gauss.set_capacity_plan(None)
    vec = torch.cat([tr.g.get_xyz.detach().reshape(-1), next(tr.motion_net.sigma_net.parameters()).detach().reshape(-1)]).cpu()
    gathered = [torch.zeros_like(vec) for _ in range(world)]
    dist.all_gather(gathered, vec)
    assert torch.equal(gathered[0], gathered[1]), "replicas diverged across the one-sided overflow"
    dist.barrier()
    if rank == 0:
        print("DP-OK", diff)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
