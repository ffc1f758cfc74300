"""The mouth train step as a hipGraph, 40 replays (for rocprofv3 --kernel-trace + scripts/median_timeline.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from instag_amd.gaussian_model import GaussianModel
from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
from instag_amd.scene_synth import synthetic_frame, synthetic_gaussians, toy_cameras
from instag_amd.train import make_frame
from instag_amd.train_stages import MouthTrainer
size, dev = 512, torch.device("cuda")
fa = SimpleNamespace(audio_extractor="deepspeech", type="face")
ma = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
cams = toy_cameras(size)
frames = [make_frame(cams[i % len(cams)].to(dev), synthetic_frame(size, i, dev, priors=True, background=True)) for i in range(8)]
bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
torch.manual_seed(0)
pc = GaussianModel(1, PersonalizedMotionNetwork(args=fa).to(dev)).load_raw(synthetic_gaussians(100000, sh_degree=1, seed=0), dev)
pcm = GaussianModel(1, PersonalizedMotionNetwork(args=ma).to(dev)).load_raw(synthetic_gaussians(20000, sh_degree=1, seed=1), dev)
mt = MouthTrainer(pcm, MouthMotionNetwork(args=ma).to(dev), pc, MotionNetwork(args=fa).to(dev), bg, densify=False)
mt.iteration = 3000
for i in range(3):
    mt.step(frames[i])
mt.enable_graph(frames[0])
for i in range(40):
    mt.step(frames[i % 8])
torch.cuda.synchronize()
