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
    for mode in ("eager", "graph"):
        tr = build_trainer(3000, dev, seed=1)
        try:
            if mode == "graph":
                tr.enable_graph(frames[0], warmup_steps=2)
                assert tr._graph.split, "several ranks must use the two-graph form"
            else:
                for _ in range(4):
                    tr.step(frames[0])
            for i in range(4):
                tr.step(frames[i % 3])
            if mode == "graph":
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
    dist.barrier()
    if rank == 0:
        print("DP-OK", diff)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
