import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instag_amd import diff_gauss
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
dev = torch.device("cuda")
cams = toy_cameras(128)
frame = make_frame(cams[0].to(dev), synthetic_frame(128, 0, dev))
tr = build_trainer(3000, dev, seed=1)
names = {}
for n, p in tr.motion_net.named_parameters(): names["umf." + n] = p
for n, p in tr.g.neural_motion_grid.named_parameters(): names["pmf." + n] = p
for i, p in enumerate(tr.g.per_gaussian_parameters()): names[f"gauss.{i}"] = p
runs = []
for r in range(3):
    pkg, loss, l1 = tr._forward_backward(frame)
    torch.cuda.synchronize()
    g = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in names.items()}
    g["vs"] = pkg["viewspace_points"].grad.detach().clone()
    g["loss"] = loss.detach().clone()
    runs.append(g)
    tr._zero_grad()
    del pkg
for k in runs[0]:
    a, b, c = runs[0][k], runs[1][k], runs[2][k]
    if a is None: continue
    d = max(float((a - b).abs().max()), float((a - c).abs().max()))
    if d > 0: print(f"{k:50s} maxdiff {d:.3e}  (max |g| {float(a.abs().max()):.3e})")
print("done")
