import sys, torch
sys.path.insert(0, "/root/repo")
from instag_amd import _lib, diff_gauss
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
dev = torch.device("cuda", 0)
orig = _lib.may_fork
log = []
def mf(device=None):
    r = orig(device)
    log.append((r, torch.cuda.is_current_stream_capturing(), str(torch.cuda.current_stream(device)), str(_lib._CAPTURE_ORIGIN)))
    return r
_lib.may_fork = mf
tr = build_trainer(20000, dev, sh_degree=1, seed=0, densify=False)
cams = toy_cameras(256)
fr = make_frame(cams[0].to(dev), synthetic_frame(256, seed=0, device=dev))
tr.step(fr)
log.clear()
g = tr.enable_graph(fr)
for x in log: print(x)
