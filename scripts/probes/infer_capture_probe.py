import sys, os, torch
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
from instag_amd import diff_gauss
from instag_amd.gaussian_model import GaussianModel
from instag_amd.infer import FuseRenderer
from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import make_frame
torch.manual_seed(9)
size = 96
fa = SimpleNamespace(audio_extractor="deepspeech", type="face")
ma = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
pc = GaussianModel(1, PersonalizedMotionNetwork(args=fa).cuda()).create_random(3000, "cuda", seed=1)
pcm = GaussianModel(1, PersonalizedMotionNetwork(args=ma).cuda()).create_random(800, "cuda", seed=2)
net, netm = MotionNetwork(args=fa).cuda(), MouthMotionNetwork(args=ma).cuda()
cams = toy_cameras(size)
frames = [make_frame(cams[i].to("cuda"), synthetic_frame(size, i, "cuda")) for i in range(3)]
r = FuseRenderer(pc, net, pcm, netm, torch.zeros(3, device="cuda"))
print("eager ok", float(r.render(frames[0]).sum()), flush=True)
r.enable_graph(frames[0], frames_per_replay=int(sys.argv[1]))
print("captured", flush=True)
print(float(r.render_batch(frames[:int(sys.argv[1])]).sum()), flush=True)
