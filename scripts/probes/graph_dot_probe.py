"""Capture the C3 train step with INSTAG_GRAPH_DOT set and print each kernel node with its predecessors."""
import os, re, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
out = os.path.abspath(os.path.join("gpurun_out", "step_graph.dot"))
os.makedirs("gpurun_out", exist_ok=True)
os.environ["INSTAG_GRAPH_DOT"] = out
from instag_amd.scene_synth import synthetic_frame, toy_cameras
from instag_amd.train import build_trainer, make_frame
dev = torch.device("cuda", 0)
tr = build_trainer(100000, dev, sh_degree=1, seed=0, densify=False)
fr = make_frame(toy_cameras(512)[0].to(dev), synthetic_frame(512, seed=0, device=dev))
tr.enable_graph(fr)
text = open(out).read()
print(len(text), "bytes of DOT")
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', text))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', text)
def short(lbl):
    m = re.search(r"(\w+_kernel|memcpy|memset|\w+Functor\w*|reduce_kernel|EMPTY|EVENT\w*|\w+)", lbl.replace("\\n", " "))
    return (lbl.replace("\\n", " ")[:70])
pred = {}
for a, b in edges: pred.setdefault(b, []).append(a)
order = list(nodes)
idx = {n: i for i, n in enumerate(order)}
for n in order:
    print(idx[n], short(nodes[n]), "<-", sorted(idx.get(p, -1) for p in pred.get(n, [])))
