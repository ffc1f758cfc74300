"""Which operator breaks hipStreamEndCapture when it runs on a stream FORKED from the capture's origin stream?
   python scripts/probes/infer_capture_probe2.py <piece>"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
from instag_amd import diff_gauss, motion_net as MN
from instag_amd.gaussian_model import GaussianModel
from instag_amd.motion_net import MotionNetwork, MouthMotionNetwork, PersonalizedMotionNetwork
from instag_amd.renderer import render, render_motion, render_motion_mouth_con
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import make_frame
piece = sys.argv[1]
torch.manual_seed(9)
size = 96
dev = torch.device("cuda")
fa = SimpleNamespace(audio_extractor="deepspeech", type="face")
ma = SimpleNamespace(audio_extractor="deepspeech", type="mouth")
pc = GaussianModel(1, PersonalizedMotionNetwork(args=fa).cuda()).create_random(3000, "cuda", seed=1)
pcm = GaussianModel(1, PersonalizedMotionNetwork(args=ma).cuda()).create_random(800, "cuda", seed=2)
net, netm = MotionNetwork(args=fa).cuda(), MouthMotionNetwork(args=ma).cuda()
frame = make_frame(toy_cameras(size)[0].to("cuda"), synthetic_frame(size, 0, "cuda")).clone_static()
bg = torch.zeros(3, device=dev)
aud, exp = frame.talking_dict["auds"], frame.talking_dict["au_exp"]
if piece.endswith("-noaudiofork"):
    MN.CONCURRENT_AUDIO = False
    piece = piece[:-len("-noaudiofork")]

@torch.no_grad()
def body():
    if piece == "raster":
        return render(frame, pc, None, bg)["render"]
    if piece == "umf":
        net.start_audio(aud, 1, exp)
        return net(pc.get_xyz, aud, exp)["_h"]
    if piece == "umf-nostart":
        return net(pc.get_xyz, aud, exp)["_h"]
    if piece == "mouthnet":
        return netm(pcm.get_xyz, aud, torch.ones(1, 3, device=dev))["d_xyz"]
    if piece == "face":
        return render_motion(frame, pc, net, None, bg, personalized=False, align=True)["render"]
    if piece == "mouth":
        net(pc.get_xyz, aud, exp)
        return render_motion_mouth_con(frame, pcm, netm, pc, net, None, bg, personalized=False, align=True,
                                       inference=True)["render"]
    if piece == "elementwise":
        return torch.sigmoid(pc.get_xyz) * 2
    raise SystemExit("unknown piece")

plan = diff_gauss.CapacityPlan([200000, 200000], dev)
diff_gauss.set_capacity_plan(plan)
lane = torch.cuda.Stream(device=dev)
def on_lane():
    main = torch.cuda.current_stream(dev)
    lane.wait_stream(main)
    with torch.cuda.stream(lane):
        plan.begin_step()
        out = body()
    main.wait_stream(lane)
    return out
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    for _ in range(2):
        on_lane()
torch.cuda.current_stream(dev).wait_stream(s)
torch.cuda.synchronize()
print(piece, "warm-up ok", flush=True)
g = torch.cuda.CUDAGraph()
import gc; gc.collect(); gc.disable()
with torch.cuda.graph(g):
    out = on_lane()
print(piece, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
print(piece, "replayed", float(out.sum()), flush=True)
