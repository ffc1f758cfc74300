"""The fuse train step as a hipGraph, 40 replays (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from instag_amd.gaussian_model import GaussianModel
from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
from instag_amd.scene_synth import synthetic_frame, synthetic_gaussians, toy_cameras
from instag_amd.train import make_frame
from instag_amd.train_stages import FuseTrainer
size, dev = 512, torch.device("cuda")
fa = SimpleNamespace(audio_extractor="deepspeech", type="face")
ma = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
cams = toy_cameras(size)
frames = [make_frame(cams[i % len(cams)].to(dev), synthetic_frame(size, i, dev, priors=True, background=True)) for i in range(8)]
bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
torch.manual_seed(0)
pc = GaussianModel(1, PersonalizedMotionNetwork(args=fa).to(dev)).load_raw(synthetic_gaussians(100000, sh_degree=1, seed=0), dev)
pcm = GaussianModel(1, PersonalizedMotionNetwork(args=ma).to(dev)).load_raw(synthetic_gaussians(20000, sh_degree=1, seed=1), dev)
ft = FuseTrainer(pc, MotionNetwork(args=fa).to(dev), pcm, MouthMotionNetwork(args=ma).to(dev), bg)
for i in range(3):
    ft.step(frames[i])
ft.enable_graph(frames[0])
for i in range(40):
    ft.step(frames[i % 8])
torch.cuda.synchronize()
