"""Forward-only fused-head render (face 100k + mouth 20k Gaussians, 512x512): frames/s eager vs hipGraph."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from instag_amd import diff_gauss
from instag_amd.gaussian_model import GaussianModel
from instag_amd.infer import FuseRenderer
from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
from instag_amd.scene_synth import synthetic_frame, synthetic_gaussians, toy_cameras
from instag_amd.train import make_frame
size = 512
fa = SimpleNamespace(audio_extractor="deepspeech", type="face")
ma = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
pc = GaussianModel(1, PersonalizedMotionNetwork(args=fa).cuda()).load_raw(synthetic_gaussians(100000, sh_degree=1, seed=0), "cuda")
pcm = GaussianModel(1, PersonalizedMotionNetwork(args=ma).cuda()).load_raw(synthetic_gaussians(20000, sh_degree=1, seed=1), "cuda")
net, netm = MotionNetwork(args=fa).cuda(), MouthMotionNetwork(args=ma).cuda()
cams = toy_cameras(size)
frames = [make_frame(cams[i % len(cams)].to("cuda"), synthetic_frame(size, i, "cuda")) for i in range(8)]
r = FuseRenderer(pc, net, pcm, netm, torch.zeros(3, device="cuda"))
def run(n):
    for i in range(n): r.render(frames[i % 8])
for mode in ("eager", "graph"):
    if mode == "graph": r.enable_graph(frames[0])
    run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 50
    run(K); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"inference {mode}: {1/dt:8.1f} frames/s  {dt*1e3:.3f} ms/frame  overflow={r.check_overflow()}", flush=True)
r.close()
for lanes in (2, 4, 8):
    r.enable_graph(frames[0], frames_per_replay=lanes)
    for _ in range(3): r.render_batch(frames[:lanes])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 24
    for i in range(K): r.render_batch([frames[(i + k) % 8] for k in range(lanes)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (K * lanes)
    print(f"inference graph, {lanes} frames per replay: {1/dt:8.1f} frames/s  {dt*1e3:.3f} ms/frame  "
          f"overflow={r.check_overflow()}", flush=True)
    r.close()
