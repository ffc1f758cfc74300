"""Train-step time of the mouth branch and of the fuse stage (face 100k + mouth 20k Gaussians, 512x512), and of
the face branch's late phase (monocular normal / depth priors: the all-channel blend backward).  Eager launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from instag_amd.gaussian_model import GaussianModel
from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
from instag_amd.scene_synth import synthetic_frame, synthetic_gaussians, toy_cameras
from instag_amd.train import FacePhase, build_trainer, make_frame
from instag_amd.train_stages import FuseTrainer, MouthTrainer

size, dev = 512, torch.device("cuda")
fa = SimpleNamespace(audio_extractor="deepspeech", type="face")
ma = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
cams = toy_cameras(size)
frames = [make_frame(cams[i % len(cams)].to(dev), synthetic_frame(size, i, dev, priors=True, background=True))
          for i in range(8)]
bg = torch.tensor([0.0, 1.0, 0.0], device=dev)


def models():
    torch.manual_seed(0)
    pc = GaussianModel(1, PersonalizedMotionNetwork(args=fa).to(dev)).load_raw(
        synthetic_gaussians(100000, sh_degree=1, seed=0), dev)
    pcm = GaussianModel(1, PersonalizedMotionNetwork(args=ma).to(dev)).load_raw(
        synthetic_gaussians(20000, sh_degree=1, seed=1), dev)
    return pc, MotionNetwork(args=fa).to(dev), pcm, MouthMotionNetwork(args=ma).to(dev)


def timed(name, step, warm=5, K=30):
    for i in range(warm):
        step(frames[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(frames[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{name:34s} {dt * 1e3:7.3f} ms/step  {1 / dt:7.1f} steps/s", flush=True)


pc, net, pcm, netm = models()
mt = MouthTrainer(pcm, netm, pc, net, bg, densify=False)
mt.iteration = 3000                    # warm phase: alignment + alpha terms
timed("mouth step (20k | face 100k), eager", mt.step)
mt.enable_graph(frames[0])
timed("mouth step, hipGraph replay", mt.step)
print("   overflow:", mt._graph.check_overflow(), "capacities", mt._graph.capacities, flush=True)
from instag_amd import diff_gauss
diff_gauss.set_capacity_plan(None)

pc, net, pcm, netm = models()
ft = FuseTrainer(pc, net, pcm, netm, bg)
timed("fuse step (100k + 20k), eager", ft.step)
ft.enable_graph(frames[0])
timed("fuse step, hipGraph replay", ft.step)
print("   overflow:", ft._graph.check_overflow(), "capacities", ft._graph.capacities, flush=True)
diff_gauss.set_capacity_plan(None)

tr = build_trainer(100000, dev, seed=0)
late = FacePhase(priors=True, prior_depth=True)
def face_late(frame):
    tr.iteration += 1
    tr._set_learning_rates(tr.iteration)
    pkg, _, _ = tr._forward_backward(frame, late)
    tr._stats_and_optimizers(pkg, False)
    tr._zero_grad()
timed("face step, normal+depth priors, eager", face_late)
def face_c3(frame):
    tr.step(frame)
timed("face step, C3 phase, eager", face_c3)
tr.phase_of = lambda it: late                 # every iteration in the late phase (the schedule's iterations > 5000)
tr.enable_graph(frames[0], phase=late)
timed("face step, normal+depth priors, hipGraph", tr.step)
print("   overflow:", tr._graph.check_overflow(), flush=True)
tr._drop_graph()
