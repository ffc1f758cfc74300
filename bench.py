#!/usr/bin/env python
"""Headline benchmark: InsTaG face-branch train step at 512x512 with 100k Gaussians (config C3).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full train step on one synthetic frame per rank: PMF + UMF (6 grid encodes + MLPs)
-> activations -> raster pass (image + attention map) -> L1 + 0.2 DSSIM + regularisers -> backward
-> fused-bucket gradient all-reduce (N>1) -> Adam/AdamW.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes itself (before this process
touches a GPU); under torch.distributed.run it must agree with WORLD_SIZE.

Timing: after W warm-up steps the trainer state is snapshotted; `--windows` (default 5) windows of exactly K steps each
start from that same state, each bracketed by barrier + synchronize on both sides and reduced with MAX over ranks; the
reported `ms_per_step` / `value` are those of the MEDIAN window (all windows are listed).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KERNEL_IDS = {"preprocess": 0, "duplicate": 1, "sort": 2, "ranges": 3, "blend_fwd": 4, "blend_bwd": 5,
              "preprocess_bwd": 6, "grid_fwd": 7, "grid_bwd": 8, "mlp_fwd": 11, "mlp_bwd": 12, "mlp_wgrad": 13,
              "blend_bwd_mean": 16, "empty_bracket": 17}
NON_RASTER = ("grid_fwd", "grid_bwd", "mlp_fwd", "mlp_bwd", "mlp_wgrad", "empty_bracket")
# the blend launches of the C3 step, each a kernel of its own (template variant) with its own byte count:
#   blend_fwd       image + attention map, all channels                     60 R + 44 P   
#   blend_bwd       colour pass, aux colours' gradient in idle GEMM columns  124 R + 44 P  
#   blend_bwd_mean  the attention map's d/dmean pass: reads id 4 + xy 8 + conic, opacity 16 + aux colour 12, writes the
#                   8-byte mean gradient per entry; per pixel dL/daux 12 + n_contrib 4 + final_T 4   48 R + 20 P
BLEND_VARIANTS = {"blend_fwd": ("blend_forward_claim_kernel<true>", 60, 44),
                  "blend_bwd": ("blend_backward_kernel<false, 2, false>", 124, 44),
                  "blend_bwd_mean": ("blend_backward_kernel<false, 0, true>", 48, 20)}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA (32x32x2 / 16x16x4), 64 FLOP/clk/SIMD
SIMDS, CLOCK_GHZ = 1024, 2.4
# roofline_valu: instructions per (Gaussian, 64-pixel wave) pair of the blend kernels' inner loops, counted in the ISA
# by scripts/isa_loop_count.py -> profiles/<tag>_blend_isa_counts.json.  MI355X_MICROARCH.md constants table: a wave64
# VALU instruction occupies its SIMD-32 for 2 cycles (transcendental: 4), v_mfma_f32_16x16x4_f32 for 32; ONE wave
# alone issues one instruction of any class per ~4 cycles (8 for a transcendental).
VALU_CYC, TRANS_CYC, MFMA_16x16x4_CYC, LONE_WAVE_CYC = 2, 4, 32, 4
PROFILE_TAG = "r03"


def algorithmic_bytes(kernel, N, M, R, P, grid_points=0):
    """Per-launch algorithmic bytes, SURVEY.md section 8(d)."""
    return {
        "preprocess": N * (128 + 12 * M),
        "duplicate": 12 * R,
        "sort": 24 * R,
        "ranges": 8 * R,
        "blend_fwd": 60 * R + 44 * P,
        "blend_bwd": 124 * R + 44 * P,
        "blend_bwd_mean": 48 * R + 20 * P,
        "preprocess_bwd": 64 * R + N * (128 + 24 * M),
        "grid_fwd": grid_points * 156,      # tri-plane launch: xyz 12 B + 3 planes x 12 levels x 4 B out
        "grid_bwd": grid_points * 168,      # + 12 B gradient to xyz
    }.get(kernel, 0)


PMC_KERNEL = {"preprocess": "preprocess_kernel", "duplicate": "duplicate_kernel", "ranges": "ranges_kernel",
              "blend_fwd": BLEND_VARIANTS["blend_fwd"][0], "blend_bwd": BLEND_VARIANTS["blend_bwd"][0],
              "blend_bwd_mean": BLEND_VARIANTS["blend_bwd_mean"][0], "preprocess_bwd": "preprocess_backward_kernel"}


def pmc_traffic(kernel, n_gaussians, size):
    """HBM bytes per launch of `kernel` (one template variant, never an average over variants) from the rocprofv3 PMC
    passes of THIS workload committed under profiles/ (written by scripts/pmc_summary.py: 2*FETCH_SIZE + WRITE_SIZE; the
    counters cannot be read from inside the run), newest round first; None for any other workload."""
    if (n_gaussians, size) != (100000, 512) or kernel not in PMC_KERNEL:
        return None, None
    for tag in (PROFILE_TAG, "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic_c3.json")
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        row = table.get(PMC_KERNEL[kernel])
        if row is None:
            continue
        return int(row["hbm_bytes_per_launch"]), os.path.basename(path)
    return None, None


def traversed_entries(state):
    """List entries each tile's waves walk in the blend kernels, from the integer state of one forward pass:
    backward walks exactly min(list length, max n_contrib of the tile) entries; the forward walks at least that many
    (it stops at the next 256-entry batch boundary after every pixel has finished, or at the end of the list)."""
    import torch
    from instag_amd import diff_gauss
    d = diff_gauss.debug_export(state)
    nc = d["n_contrib"]
    H, W = nc.shape
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = torch.zeros(gy * 16, gx * 16, dtype=nc.dtype, device=nc.device)
    pad[:H, :W] = nc
    tile_max = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values.to(torch.int64)
    length = (d["ranges"][:, 1] - d["ranges"][:, 0]).to(torch.int64)
    walked = torch.minimum(length, tile_max)
    return {"bwd": int(walked.sum()), "fwd_min": int(walked.sum()), "fwd_max": int(length.sum()),
            "tiles_populated": int((length > 0).sum()), "longest_list": int(length.max()),
            "longest_walk": int(walked.max())}


def roofline_valu(kernel, variants, entries, avg_us, launches_per_step):
    """VALU / MFMA issue floor of a blend kernel: `entries` list entries x 4 waves per tile = (Gaussian, wave) pairs."""
    table = None
    for tag in (PROFILE_TAG, "r02", "r01"):
        try:
            table = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_blend_isa_counts.json")))
            src = f"{tag}_blend_isa_counts.json"
            break
        except (OSError, ValueError):
            continue
    if table is None or not entries:
        return None
    rows = [table[v]["per_gaussian"] for v, _ in variants if v in table]
    if len(rows) != len(variants):
        return None
    pairs = 4 * entries
    valu = sum(r.get("valu", 0.0) for r in rows) / len(rows)
    trans = sum(r.get("valu_trans", 0.0) for r in rows) / len(rows)
    every = sum(r["all"] for r in rows) / len(rows)
    nmat = sum(n for _, n in variants) / len(variants)
    simd_hz = SIMDS * CLOCK_GHZ * 1e9
    valu_us = pairs * (valu * VALU_CYC + trans * TRANS_CYC) / simd_hz * 1e6
    mfma_us = pairs * nmat * MFMA_16x16x4_CYC / simd_hz * 1e6      # phase B: one 16x16x4 per (matrix, Gaussian, wave)
    lone_us = pairs * ((every - trans) * LONE_WAVE_CYC + trans * 2 * LONE_WAVE_CYC + nmat * MFMA_16x16x4_CYC) / simd_hz * 1e6
    floor = max(valu_us, mfma_us)
    return {"kernel": kernel, "variants": [v for v, _ in variants], "isa_counts": src,
            "pairs_per_launch": pairs, "valu_per_pair": round(valu, 2), "transcendental_per_pair": round(trans, 2),
            "instructions_per_pair": round(every, 2), "mfma_16x16x4_per_pair": nmat,
            "valu_floor_us": round(valu_us, 2), "mfma_floor_us": round(mfma_us, 2),
            "floor_us": round(floor, 2), "achieved_us": round(avg_us, 2), "frac": round(floor / avg_us, 4),
            "one_wave_per_simd_us": round(lone_us, 2),
            "note": "floor = max(VALU, MFMA) issue time with every SIMD busy; one_wave_per_simd_us = the same pairs "
                    "issued by lone waves (any instruction 4 cycles) spread evenly over all 1024 SIMDs -- the step's "
                    "~410 populated tiles x 4 waves cover at most 1640 wave slots and finish unevenly"}


def cpu_baseline(n_gaussians, size, sh_degree, budget_s=30.0):
    """Oracle (pure-PyTorch CPU rasterizer) fwd+bwd of the same scene on the host cores."""
    import torch
    from instag_amd.scene_synth import activated, synthetic_gaussians, toy_cameras
    from oracle import rasterize_ref as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))            # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(cores)
    cam = toy_cameras(size)[0]
    a = activated(synthetic_gaussians(n_gaussians, sh_degree=sh_degree, seed=0))
    s = R.RasterSettings(size, size, cam.tanfovx, cam.tanfovy, torch.tensor([0.0, 1.0, 0.0]), 1.0,
                         cam.world_view_transform, cam.full_proj_transform, sh_degree, cam.camera_center)
    frames, t0 = 0, time.perf_counter()
    while True:
        inp = {k: v.clone().requires_grad_(True) for k, v in a.items()}
        m2 = torch.zeros(n_gaussians, 3, requires_grad=True)
        outs = R.rasterize(inp["means3D"], m2, inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"],
                           None, torch.ones(n_gaussians, 1), s)
        (outs[0].sum() + outs[3].sum()).backward()
        frames += 1
        el = time.perf_counter() - t0
        if el * (frames + 1) / frames > budget_s or frames >= 2:
            break
    return {"value": frames / el, "unit": "frames/s of the rasterizer's forward+backward ONLY (not the train step)",
            "cores": cores, "kind": "port",
            "sample": f"{frames} frame(s) of the main raster pass fwd+bwd of the same {n_gaussians}-Gaussian scene "
                      f"@{size}x{size} (oracle/rasterize_ref.py, torch {torch.get_num_threads()} threads); the GPU "
                      f"figure to set beside it is raster_fwd_bwd_ms_per_frame, not value"}


def spawn_ranks(n):
    """`--gpus n` without a launcher: start the n rank processes.  Runs BEFORE this process touches a GPU (counting
    devices does not initialise one on this image) and never re-executes a process that has."""
    import torch
    forced = os.environ.get("INSTAG_BENCH_FORCE_DEVICE") is not None
    have = torch.cuda.device_count()
    if have < n and not forced:
        raise SystemExit(f"bench.py --gpus {n}: needs {n} devices, this node shows {have}")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "2"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:          # a rank died: the others would wait for it for ever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for q in procs:
            q.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps, each from the same state")
    ap.add_argument("--gaussians", type=int, default=100000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--sh-degree", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stable-targets", action="store_true", help="skip the second workload (rendered targets)")
    ap.add_argument("--no-graph", action="store_true", help="launch every operator eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--no-host-frames", action="store_true",
                    help="skip the PCIe-inclusive workload (frames in pinned host memory, uploaded one step ahead)")
    ap.add_argument("--no-schedule", action="store_true",
                    help="skip the reference_schedule workload (700 iterations across density-control events)")
    ap.add_argument("--schedule-iterations", type=int, default=700)
    ap.add_argument("--all-workloads", action="store_true",
                    help="several ranks: also run the secondary workloads (by default a multi-rank run measures the "
                         "headline workload only: its line must not depend on the extras)")
    ap.add_argument("--allow-eager-fallback", action="store_true",
                    help="several ranks: if the step cannot be captured next to the collective library, time eager "
                         "launches instead of failing (the line then says so in config.execution)")
    args = ap.parse_args()

    t_start = time.perf_counter()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)               # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: they must agree")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal knobs (several ranks on ONE card over gloo); the driver's runs use neither
    if os.environ.get("INSTAG_BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["INSTAG_BENCH_FORCE_DEVICE"])
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: needs device {local_rank}, this node shows {torch.cuda.device_count()}")
    backend = os.environ.get("INSTAG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)               # the communicator exists before anything is captured
        assert int(probe.item()) == world == dist.get_world_size()

    from instag_amd import _lib, diff_gauss
    from instag_amd.scene_synth import synthetic_frame, toy_cameras
    from instag_amd.train import build_trainer, make_frame

    N, size = args.gaussians, args.size
    cams = toy_cameras(size)

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    def make_frames(trainer, stable):
        """rank r renders frames r, r+world, ... (all resident in HBM before timing).  stable=False: SURVEY 8(d)'s
        targets (uniform noise); stable=True: the targets are renders of a perturbed copy of the scene itself."""
        frames = []
        for k in range(8):
            idx = (rank + k * world) % len(cams)
            fd = synthetic_frame(size, seed=rank + k * world, device=dev)
            cam = cams[idx].to(dev)
            if stable:
                fd["gt_image"] = rendered_target(trainer, cam, fd, seed=1000 + rank + k * world)
            frames.append(make_frame(cam, fd))
        return frames

    def rendered_target(trainer, cam, fd, seed):
        from instag_amd.renderer import render
        g = trainer.g
        gen = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            keep = {k: g._p[k].data.clone() for k in ("xyz", "f_dc")}
            g._p["xyz"].data.add_(torch.randn(g._p["xyz"].shape, generator=gen, device=dev) * 2e-4)
            g._p["f_dc"].data.add_(torch.randn(g._p["f_dc"].shape, generator=gen, device=dev) * 0.05)
            img = render(cam, g, None, trainer.bg)["render"].clamp(0.0, 1.0).clone()
            for k, v in keep.items():
                g._p[k].data.copy_(v)
        return img

    L = _lib.lib()
    import ctypes as C

    def measure(stable, windows):
        """-> dict(ms windows, median, kernels, R, graph info) of one workload, from a freshly built trainer."""
        trainer = build_trainer(N, dev, sh_degree=args.sh_degree, seed=0, densify=False)
        frames = make_frames(trainer, stable)

        def run(n):
            for i in range(n):
                trainer.step(frames[i % len(frames)])

        use_graph = not args.no_graph
        graph, why = None, None
        if use_graph:
            # whole step captured into a hipGraph (rasterizer in sync-free capacity mode); see instag_amd/train.py
            try:
                graph = trainer.enable_graph(frames[0])
                log(f"step captured into a hipGraph (instance capacity {graph.capacity})")
            except Exception as exc:       # e.g. a collective library that cannot coexist with stream capture
                why = f"{type(exc).__name__}: {exc}"
                if world > 1 and not args.allow_eager_fallback:
                    # a scaling line must never silently be an eager number
                    raise SystemExit(f"rank {rank}: graph capture failed with {world} ranks ({why}); "
                                     "pass --allow-eager-fallback to time eager launches instead")
                log(f"graph capture failed ({why}); running the same HIP operators eagerly")
                trainer._drop_graph()
                diff_gauss.set_capacity_plan(None)
                use_graph = False
        run(args.warmup)
        torch.cuda.synchronize()
        snap = trainer.snapshot()
        if graph is not None:
            graph.plan.clear()
        times = []
        for w in range(windows):
            trainer.restore(snap)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            times.append(el)
        log(f"{'rendered' if stable else 'noise'} targets: windows ms/step " +
            " ".join(f"{1e3 * t / args.steps:.3f}" for t in times))
        if use_graph:
            overflow = graph.check_overflow()
            if overflow:
                raise SystemExit(f"instance capacity exceeded during the timed region: {overflow}")
        # Per-kernel durations IN REPLAY MODE: the step is captured once more with every launch of the C ABI bracketed by
        # external event-record nodes (instag_prof_graph_*: HIP events on the stream the kernel is launched on, inside the
        # graph) and replayed the same number of steps from the SAME restored state; the kernels run next to the same
        # concurrent branches as in the timed windows.  (The step is captured again WITHOUT running a train step:
        # FaceTrainer._recapture.)  If the runtime refuses external event nodes the instrumented steps run eagerly.
        trainer.restore(snap)
        from instag_amd import mlp as mlp_ops
        n_rendered = 0
        if use_graph:
            need = graph.plan.needed()
            n_rendered = int(max(need)) if need else 0
        kern, entries, mlp_stats, source = {}, None, None, None
        if use_graph and not os.environ.get("INSTAG_BENCH_EAGER_EVENTS"):
            try:
                kern, entries, mlp_stats = replay_mode_durations(trainer, frames, snap)
                source = "external event-record nodes inside the replayed hipGraph"
            except Exception as exc:
                log(f"graph-mode kernel timing unavailable ({type(exc).__name__}: {exc}); timing eager launches instead")
                L.instag_prof_enable(0)
                trainer.restore(snap)
        if not kern:
            trainer._drop_graph()
            diff_gauss.set_capacity_plan(None)
            L.instag_prof_enable(-1)
            L.instag_prof_reset()
            mlp_ops.STATS.update(fwd_flops=0, bwd_flops=0)
            diff_gauss.KEEP_LAST_STATE = True
            run(args.steps)
            torch.cuda.synchronize()
            diff_gauss.KEEP_LAST_STATE = False
            entries = traversed_entries(diff_gauss.LAST_STATS.pop("state")) if rank == 0 else None
            kern = read_kernels()
            L.instag_prof_enable(0)
            mlp_stats = dict(mlp_ops.STATS)
            n_rendered = int(diff_gauss.LAST_STATS.get("num_rendered", 0))
            source = "HIP events around eager launches (not the replayed graph)"
        trainer._drop_graph()
        diff_gauss.set_capacity_plan(None)
        med = statistics.median(times)
        return dict(times=times, median=med, kern=kern, R=n_rendered, graph=use_graph, why=why, mlp=mlp_stats,
                    entries=entries, durations_from=source)

    def read_kernels():
        kern = {}
        for name, kid in KERNEL_IDS.items():
            ms, cnt = C.c_double(0), C.c_int64(0)
            L.instag_prof_read(kid, C.byref(ms), C.byref(cnt))
            if cnt.value:
                kern[name] = {"launches": int(cnt.value), "avg_us": 1e3 * ms.value / cnt.value, "total_ms": ms.value}
        return kern

    def replay_mode_durations(trainer, frames, snap):
        import gc
        from instag_amd import mlp as mlp_ops
        _lib.check(L.instag_prof_graph_begin(512), "prof_graph_begin")
        try:
            L.instag_prof_enable(int(os.environ.get("INSTAG_BENCH_PROF_MASK", "-1")))
            L.instag_prof_reset()
            trainer._drop_graph(keep_mode=True)          # the next step() captures the step again, now instrumented
            mlp_ops.STATS.update(fwd_flops=0, bwd_flops=0)
            diff_gauss.KEEP_LAST_STATE = True
            entries = None
            for i in range(args.steps):
                trainer.step(frames[i % len(frames)])
                torch.cuda.synchronize()
                if i == 0:
                    diff_gauss.KEEP_LAST_STATE = False
                    mlp_stats = {k: v * args.steps for k, v in mlp_ops.STATS.items()}    # counted at capture: one step's
                    if L.instag_prof_graph_pairs_used() == 0:
                        raise RuntimeError("no external event pair was captured")
                _lib.check(L.instag_prof_graph_collect(), "prof_graph_collect")
            if rank == 0:
                entries = traversed_entries(diff_gauss.LAST_STATS.pop("state"))
            else:
                diff_gauss.LAST_STATS.pop("state", None)
            kern = read_kernels()
            return kern, entries, mlp_stats
        finally:
            diff_gauss.KEEP_LAST_STATE = False
            L.instag_prof_enable(0)
            trainer._drop_graph(keep_mode=True)          # the graphs that hold the pool's events go first
            gc.collect()
            torch.cuda.synchronize()
            L.instag_prof_graph_end()

    def measure_host_frames(windows):
        """PCIe-inclusive variant of the headline workload: the frames live in pinned HOST memory and are uploaded one
        step ahead on a copy stream (instag_amd/train.py HostFrameFeeder; train_face.py:324-327,
        gaussian_renderer/__init__.py:188-189 upload them inside the step)."""
        from instag_amd.train import HostFrameFeeder
        trainer = build_trainer(N, dev, sh_degree=args.sh_degree, seed=0, densify=False)
        frames = make_frames(trainer, False)
        host = [HostFrameFeeder.to_host(f) for f in frames]
        feeder = HostFrameFeeder(frames[0], dev)
        graph = trainer.enable_graph(frames[0])

        def run(n):
            feeder.run(trainer.step, host, n)

        run(args.warmup)
        torch.cuda.synchronize()
        snap = trainer.snapshot()
        times = []
        for _ in range(windows):
            trainer.restore(snap)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            times.append(el)
        overflow = graph.check_overflow()
        if overflow:
            raise SystemExit(f"instance capacity exceeded during the host-frames windows: {overflow}")
        trainer._drop_graph()
        diff_gauss.set_capacity_plan(None)
        log("host frames (PCIe-inclusive): windows ms/step " + " ".join(f"{1e3 * t / args.steps:.3f}" for t in times))
        return dict(times=times, median=statistics.median(times), bytes_per_frame=int(host[0]._buf.numel()))

    def measure_schedule(iterations, densify, schedule="reference", start=550):
        """End-to-end throughput of a run across density-control events (train_face.py:667-746: densify / prune every 100
        iterations after iteration 500, arguments/__init__.py:92-97) from iteration `start`.  Every event drops the
        captured steps; step() captures them again by itself (FaceTrainer._recapture).
        schedule="reference": the reference's iteration-dependent phases (the alignment switches on after iteration 1000:
        one phase change inside the run) and its extra prunes; schedule=None: the C3 phase (the headline's step) in every
        iteration.  densify=False is the control: same iterations and phases, no density control.
        (The run stays below iteration 3000: from there on the reference prunes every Gaussian whose screen radius exceeds
        20 px, which in SURVEY 8(d)'s synthetic scene -- sigma = 6.4 px -- is nearly all of them.)"""
        trainer = build_trainer(N, dev, sh_degree=args.sh_degree, seed=0, densify=densify, schedule=schedule)
        frames = make_frames(trainer, True)
        trainer.iteration = start
        trainer.enable_graph(frames[0], keep_state=True)
        n0 = trainer.g.num_points
        counts, events = [n0], 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(iterations):
            due = trainer._densify_due(trainer.iteration + 1)
            trainer.step(frames[i % len(frames)])
            if due:
                events += 1
                counts.append(trainer.g.num_points)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        rec = trainer.recaptures
        trainer._drop_graph()
        diff_gauss.set_capacity_plan(None)
        log(f"schedule={schedule} densify={densify}: {iterations} iterations in {el:.3f} s, {events} density-control "
            f"events, {rec} re-captures ({1e3 * getattr(trainer, 'recapture_seconds', 0.0):.1f} ms of host time), density "
            f"control {1e3 * getattr(trainer, 'density_seconds', 0.0):.1f} ms, Gaussians {counts[0]} -> {counts[-1]}")
        return dict(seconds=el, iterations=iterations, events=events, recaptures=rec, gaussians=counts,
                    recapture_ms=1e3 * getattr(trainer, "recapture_seconds", 0.0),
                    density_ms=1e3 * getattr(trainer, "density_seconds", 0.0), start=start)

    if world > 1 and not args.all_workloads:
        args.no_stable_targets = args.no_host_frames = args.no_schedule = True
    log(f"config: {N} Gaussians, {size}x{size}, world {world}")
    main_run = measure(False, max(1, args.windows))
    stable_run = None
    if not args.no_stable_targets:
        stable_run = measure(True, max(1, min(3, args.windows)))

    host_run = None if args.no_host_frames else measure_host_frames(max(1, min(3, args.windows)))
    sched_run = sched_ctrl = c3_run = None
    if not args.no_schedule:
        sched_run = measure_schedule(args.schedule_iterations, True)
        sched_ctrl = measure_schedule(args.schedule_iterations, False)
        c3_run = measure_schedule(args.schedule_iterations, True, schedule=None)

    if rank == 0:
        kern, R, med = main_run["kern"], main_run["R"], main_run["median"]
        P = size * size
        M = (args.sh_degree + 1) ** 2
        raster_kernels = [k for k in kern if k not in NON_RASTER]
        ent = main_run["entries"]
        # One roofline object per blend launch VARIANT (each is a kernel of its own): its own algorithmic byte count
        # (SURVEY 8d's per-instance / per-pixel figures x the measured R, P) / its own average launch duration.  Next to
        # it: the bytes of the list entries the launch really walks (entries in front of each tile's last contributor)
        # and the HBM bytes the PMC passes counted for that variant.  `roofline` is the variant furthest below the bound.
        # A bracket (event record -> kernel -> event record) also times the dispatch latency behind the first record and
        # the completion signal in front of the second: measured by an EMPTY bracket in the same graph and taken off.
        bracket_us = kern["empty_bracket"]["avg_us"] if "empty_bracket" in kern else 0.0
        blend = {}
        for k, (variant, per_r, per_p) in BLEND_VARIANTS.items():
            if k not in kern:
                continue
            dur = max(kern[k]["avg_us"] - bracket_us, 1e-3) * 1e-6
            ab = per_r * R + per_p * P
            walked = ent["bwd" if k != "blend_fwd" else "fwd_min"] if ent else None
            traffic, traffic_src = pmc_traffic(k, N, size)
            blend[k] = {"bound": "hbm", "kernel": k, "variant": variant, "achieved": round(ab / dur / 1e9, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ab / dur / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "achieved_counter": None if traffic is None else round(traffic / dur / 1e9, 2),
                        "algorithmic_bytes_per_launch": ab, "bytes_per_instance": per_r, "bytes_per_pixel": per_p,
                        "walked_entries": walked,
                        "walked_bytes": None if walked is None else per_r * walked + per_p * P,
                        "achieved_walked": None if walked is None else round((per_r * walked + per_p * P) / dur / 1e9, 2),
                        "avg_launch_us": round(dur * 1e6, 2), "avg_bracket_us": round(kern[k]["avg_us"], 2),
                        "empty_bracket_us": round(bracket_us, 2), "launches": kern[k]["launches"],
                        "num_rendered": R, "durations_from": main_run["durations_from"]}
        roofline = min(blend.values(), key=lambda r: r["frac"]) if blend else None
        dom = roofline["kernel"] if roofline else None
        if roofline:
            roofline = dict(roofline, blend_launches={k: v for k, v in blend.items() if k != dom})
        # the other two figures SURVEY.md 8(d) asks for: hash-grid GB/s (HBM) and the MLP kernels' MFMA rate
        secondary = {}
        for k in ("grid_fwd", "grid_bwd"):
            if k in kern:
                ab = algorithmic_bytes(k, N, M, R, P, grid_points=N)
                gbs = ab / (kern[k]["avg_us"] * 1e-6) / 1e9
                secondary[k] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": ab,
                                "avg_launch_us": round(kern[k]["avg_us"], 2)}
        for k, key in (("mlp_fwd", "fwd_flops"), ("mlp_bwd", "bwd_flops")):
            if k in kern and kern[k]["total_ms"] > 0:
                tf = main_run["mlp"][key] / (kern[k]["total_ms"] * 1e-3) / 1e12
                secondary[k] = {"bound": "mfma", "dtype": "f32", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF,
                                "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                                "flops_per_step": main_run["mlp"][key] // max(1, args.steps),
                                "launches_per_step": kern[k]["launches"] // max(1, args.steps)}
        # the blend kernels are instruction-issue bound, not HBM bound: their VALU / MFMA floors
        valu = {}
        if ent and "blend_bwd" in kern:
            valu["blend_bwd"] = roofline_valu("blend_bwd", [(BLEND_VARIANTS["blend_bwd"][0], 2)], ent["bwd"],
                                              blend["blend_bwd"]["avg_launch_us"], 1)
        if ent and "blend_bwd_mean" in kern:
            valu["blend_bwd_mean"] = roofline_valu("blend_bwd_mean", [(BLEND_VARIANTS["blend_bwd_mean"][0], 1)],
                                                   ent["bwd"], blend["blend_bwd_mean"]["avg_launch_us"], 1)
        if ent and "blend_fwd" in kern:
            valu["blend_fwd"] = roofline_valu("blend_fwd", [(BLEND_VARIANTS["blend_fwd"][0], 0)], ent["fwd_min"],
                                              blend["blend_fwd"]["avg_launch_us"], 1)
            if valu["blend_fwd"]:
                valu["blend_fwd"]["pairs_are"] = "a lower bound (entries up to the tile's last contributor)"
        value = world * args.steps / med
        out = {
            "metric": f"train-step frames/sec @{size}x{size}, {N // 1000}k Gaussians", "value": round(value, 3),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * med / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: train_face step, {N // 1000}k Gaussians + 6 grid encodes + UMF/PMF motion nets "
                                   f"(DeepSpeech feats), {size}x{size}, SH degree {args.sh_degree}, image + attention "
                                   "map in one raster pass, L1+DSSIM, Adam; targets = uniform noise (SURVEY 8d)",
                       "gaussians": N, "image": [size, size], "sh_degree": args.sh_degree,
                       "frames_per_step_per_gpu": 1, "parallelism": f"dp{world}",
                       "ranks_reported_by_backend": dist.get_world_size() if world > 1 else 1,
                       "execution": "hipGraph replay" if main_run["graph"] else
                                    ("eager" + (f" (graph capture failed: {main_run['why']})" if main_run["why"] else "")),
                       "timing": f"median of {len(main_run['times'])} windows of {args.steps} steps, each from the "
                                 "same snapshotted state"},
            "windows_ms_per_step": [round(1e3 * t / args.steps, 4) for t in main_run["times"]],
            "roofline": roofline,
            "roofline_valu": valu.get(dom) and dict(valu[dom], other={k: v for k, v in valu.items() if v and k != dom},
                                                    walked=ent),
            "kernel_durations_from": main_run["durations_from"],
            "secondary_rooflines": secondary,
            "kernels_us": {k: round(v["avg_us"], 2) for k, v in kern.items()},
            "raster_fwd_bwd_ms_per_frame": round(sum(v["total_ms"] for k, v in kern.items()
                                                     if k not in NON_RASTER) / args.steps, 4),
        }
        if stable_run is not None:
            sm = stable_run["median"]
            out["stable_targets"] = {
                "workload": "same step; every frame's target image is a render of a perturbed copy of the scene "
                            "(positions +-0.2 mm, colours +-0.05), so the loss is small and opacities do not drift",
                "value": round(world * args.steps / sm, 3), "unit": "frames/s",
                "ms_per_step": round(1e3 * sm / args.steps, 4),
                "windows_ms_per_step": [round(1e3 * t / args.steps, 4) for t in stable_run["times"]],
                "num_rendered": stable_run["R"],
                "kernels_us": {k: round(v["avg_us"], 2) for k, v in stable_run["kern"].items()
                               if k in ("blend_fwd", "blend_bwd", "sort", "preprocess")}}
        if host_run is not None:
            hm = host_run["median"]
            out["host_frames"] = {
                "workload": "same step, PCIe-inclusive: every frame (image, masks, audio window, expression vector = one "
                            "packed buffer) lives in pinned host memory and is uploaded one step ahead on a copy stream "
                            "into a device staging buffer (the reference uploads inside the step, train_face.py:324-327)",
                "value": round(world * args.steps / hm, 3), "unit": "frames/s",
                "ms_per_step": round(1e3 * hm / args.steps, 4),
                "windows_ms_per_step": [round(1e3 * t / args.steps, 4) for t in host_run["times"]],
                "host_bytes_per_frame": host_run["bytes_per_frame"],
                "vs_resident": round((args.steps / hm) / (args.steps / med), 4)}
        def schedule_entry(run, what):
            fps = world * run["iterations"] / run["seconds"]
            return {"workload": what, "value": round(fps, 3), "unit": "frames/s",
                    "iterations": [run["start"] + 1, run["start"] + run["iterations"]],
                    "seconds": round(run["seconds"], 4), "density_control_events": run["events"],
                    "recaptures": run["recaptures"], "recapture_host_ms": round(run["recapture_ms"], 2),
                    "density_control_host_ms": round(run["density_ms"], 2),
                    "gaussians_after_each_event": run["gaussians"], "vs_headline": round(fps / value, 4)}
        if sched_run is not None:
            out["reference_schedule"] = schedule_entry(
                sched_run, "FaceTrainer(schedule='reference', densify=True): the reference's phases (alignment on after "
                           "iteration 1000: one phase change) and density control (densify / prune every 100 iterations + "
                           "its colour / depth prunes), rendered targets, end to end INCLUDING density control and the "
                           "re-captures of the step it forces")
            if sched_ctrl is not None:
                cfps = world * sched_ctrl["iterations"] / sched_ctrl["seconds"]
                out["reference_schedule"]["same_schedule_without_density_control"] = {
                    "value": round(cfps, 3), "unit": "frames/s", "recaptures": sched_ctrl["recaptures"]}
                out["reference_schedule"]["vs_no_density_control"] = round(out["reference_schedule"]["value"] / cfps, 4)
        if c3_run is not None:
            out["c3_phase_with_density_control"] = schedule_entry(
                c3_run, "the headline's step (C3 phase in every iteration) with density control on (densify / prune every "
                        "100 iterations), rendered targets, end to end including the re-captures")
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU oracle baseline (bounded sample)")
            out["cpu_baseline"] = cpu_baseline(N, size, args.sh_degree, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
