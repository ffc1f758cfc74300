"""Generate the committed golden fixtures from the reference checkout.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

Writes small .npz / .json DATA files (inputs + expected outputs) next to this
script and the camera extract ``instag_amd/data/toy_cameras.json``.  The
reference's importable Python helpers are imported from /root/reference; its
CUDA extensions are never imported (SURVEY.md §0 hazard).  The SH-encoder
vectors are obtained by reading shencoder/src/shencoder.cu as text and
evaluating its polynomial table numerically (fp32) on seeded inputs.
"""
import sys
sys.dont_write_bytecode = True   # never write __pycache__ into the read-only reference tree
import importlib.util
import json
import os
import re
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cameras():
    src = json.load(open(f"{REF}/camera_extrinsic_toy_test/transforms_val.json"))
    idx = list(range(0, 301, 20))
    out = dict(focal_len=src["focal_len"], cx=src["cx"], cy=src["cy"], source_indices=idx,
               frames=[dict(transform_matrix=src["frames"][i]["transform_matrix"]) for i in idx])
    os.makedirs(f"{ROOT}/instag_amd/data", exist_ok=True)
    json.dump(out, open(f"{ROOT}/instag_amd/data/toy_cameras.json", "w"))
    # G2: matrices through the reference's own helper functions
    gu = _load(f"{REF}/utils/graphics_utils.py", "ref_graphics_utils")
    res = {}
    for k, i in enumerate(idx[:4]):
        c2w = np.array(src["frames"][i]["transform_matrix"])
        c2w[:3, 1:3] *= -1
        w2c = np.linalg.inv(c2w)
        R = np.transpose(w2c[:3, :3])
        T = w2c[:3, 3]
        fov = gu.focal2fov(src["focal_len"], 512)
        V = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        P = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fov, fovY=fov).transpose(0, 1)
        full = (V.unsqueeze(0).bmm(P.unsqueeze(0))).squeeze(0)
        center = V.inverse()[3, :3]
        res[f"view_{k}"] = V.numpy()
        res[f"full_{k}"] = full.numpy()
        res[f"center_{k}"] = center.numpy()
        res[f"fov_{k}"] = np.float64(fov)
    np.savez(f"{HERE}/g2_cameras.npz", **res)


def eval_sh():
    sh = _load(f"{REF}/utils/sh_utils.py", "ref_sh_utils")
    g = torch.Generator().manual_seed(11)
    coef = torch.randn(64, 3, 16, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    res = dict(coef=coef.numpy(), dirs=dirs.numpy())
    for deg in range(4):
        res[f"deg{deg}"] = sh.eval_sh(deg, coef, dirs).numpy()
    res["rgb2sh"] = sh.RGB2SH(torch.linspace(0, 1, 9)).numpy()
    np.savez(f"{HERE}/g1_eval_sh.npz", **res)


def losses():
    sys.path.insert(0, REF)
    lu = _load(f"{REF}/utils/loss_utils.py", "ref_loss_utils")
    iu = _load(f"{REF}/utils/image_utils.py", "ref_image_utils")
    sys.path.pop(0)
    g = torch.Generator().manual_seed(12)
    a = torch.rand(3, 64, 64, generator=g)
    b = (a + 0.1 * torch.randn(3, 64, 64, generator=g)).clamp(0, 1)
    res = dict(a=a.numpy(), b=b.numpy(), l1=lu.l1_loss(a, b).numpy(), ssim=lu.ssim(a, b).numpy(),
               psnr=iu.psnr(a[None], b[None]).numpy())
    if hasattr(lu, "normalize"):
        res["normalize"] = lu.normalize(a[0]).numpy()
    np.savez(f"{HERE}/g3_losses.npz", **res)


def lr_schedule():
    sys.path.insert(0, REF)
    gen = _load(f"{REF}/utils/general_utils.py", "ref_general_utils")
    sys.path.pop(0)
    f = gen.get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=45000)
    steps = np.array([0, 1, 100, 1000, 10000, 45000])
    vals = np.array([f(int(s)) for s in steps], dtype=np.float64)
    inv = gen.inverse_sigmoid(torch.tensor([0.1, 0.5, 0.9])).numpy()
    np.savez(f"{HERE}/g4_lr.npz", steps=steps, vals=vals, inverse_sigmoid=inv)


def sh_encoder_table():
    """Evaluate the reference kernel's polynomial table (read as text) on seeded inputs."""
    text = open(f"{REF}/shencoder/src/shencoder.cu").read()
    pat = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.*?)\s*;", re.M)
    table = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    for name, i, expr in pat.findall(text):
        table[name][int(i)] = re.sub(r"(\d+\.?\d*(?:[eE][-+]?\d+)?)f\b", r"\1", expr)
    assert all(len(table[k]) == 64 for k in table), {k: len(v) for k, v in table.items()}
    g = torch.Generator().manual_seed(13)
    pts = torch.nn.functional.normalize(torch.randn(96, 3, generator=g), dim=-1)
    pts[64:] = torch.rand(32, 3, generator=g) * 2 - 1      # not unit length: table is polynomial
    P = pts.numpy().astype(np.float32)
    f32 = np.float32
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    env = dict(x=x, y=y, z=z, xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z)
    env["xyz"] = env["xy"] * z
    env["x4"], env["y4"], env["z4"] = env["x2"] * env["x2"], env["y2"] * env["y2"], env["z2"] * env["z2"]
    env["x6"], env["y6"], env["z6"] = env["x4"] * env["x2"], env["y4"] * env["y2"], env["z4"] * env["z2"]

    class F(float):
        pass

    def ev(expr):
        # numeric literals -> fp32 so every operation stays in fp32 like the kernel
        e = re.sub(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\d+\.?)(?![\w.])", r"f32(\1)", expr)
        v = eval(e, {"f32": f32}, env)
        return np.broadcast_to(np.asarray(v, dtype=np.float32), x.shape).copy()

    res = dict(inputs=P)
    for k in table:
        res[k] = np.stack([ev(table[k][i]) for i in range(64)], axis=1)
    np.savez(f"{HERE}/g6_sh_encoder.npz", **res)


def motion_nets():
    """G5: the reference's UMF / PMF forward on CPU with the oracle grid encoder injected as `gridencoder`."""
    import types
    from argparse import Namespace
    sys.path.insert(0, ROOT)
    from oracle import grid_torch
    fake = types.ModuleType("gridencoder")
    fake.GridEncoder = grid_torch.GridEncoder
    sys.modules["gridencoder"] = fake
    sys.path.insert(0, REF)
    mn = _load(f"{REF}/scene/motion_net.py", "ref_motion_net")
    sys.path.remove(REF)
    torch.manual_seed(21)
    res = {}
    g = torch.Generator().manual_seed(22)
    x = torch.rand(256, 3, generator=g) * 0.2 - 0.1
    a = torch.randn(8, 29, 16, generator=g)
    e = torch.rand(6, generator=g)
    res.update(x=x.numpy(), a=a.numpy(), e=e.numpy())
    for tag, cls in (("umf", mn.MotionNetwork), ("pmf", mn.PersonalizedMotionNetwork)):
        net = cls(args=Namespace(audio_extractor="deepspeech", type="face"))
        with torch.no_grad():
            for n_, p_ in net.named_parameters():
                if n_.endswith("embeddings"):
                    p_.copy_(torch.randn(p_.shape, generator=g) * 0.1)
        out = net(x, a, e)
        for k_, v_ in net.state_dict().items():
            res[f"{tag}.sd.{k_}"] = v_.numpy()
        for k_, v_ in out.items():
            if v_ is not None:
                res[f"{tag}.out.{k_}"] = v_.detach().numpy()
    np.savez_compressed(f"{HERE}/g5_motion_nets.npz", **res)


def mouth_nets():
    """G7: the reference's MouthMotionNetwork and the mouth-type PMF forward on CPU (oracle grid encoder injected)."""
    import types
    from argparse import Namespace
    sys.path.insert(0, ROOT)
    from oracle import grid_torch
    fake = types.ModuleType("gridencoder")
    fake.GridEncoder = grid_torch.GridEncoder
    sys.modules["gridencoder"] = fake
    sys.path.insert(0, REF)
    mn = _load(f"{REF}/scene/motion_net.py", "ref_motion_net_mouth")
    sys.path.remove(REF)
    torch.manual_seed(31)
    g = torch.Generator().manual_seed(32)
    x = torch.rand(192, 3, generator=g) * 0.2 - 0.1
    a = torch.randn(8, 29, 16, generator=g)
    move = torch.randn(1, 3, generator=g)
    res = dict(x=x.numpy(), a=a.numpy(), move=move.numpy())
    args = Namespace(audio_extractor="deepspeech", type="mouth")
    for tag, net, call in (("mouth", mn.MouthMotionNetwork(args=args), lambda n: n(x, a, move)),
                           ("pmf_mouth", mn.PersonalizedMotionNetwork(args=args), lambda n: n(x, a))):
        with torch.no_grad():
            for n_, p_ in net.named_parameters():
                if n_.endswith("embeddings"):        # fp16-representable values: stored as fp16 without loss
                    p_.copy_((torch.randn(p_.shape, generator=g) * 0.1).half().float())
        out = call(net)
        for k_, v_ in net.state_dict().items():
            res[f"{tag}.sd.{k_}"] = v_.numpy().astype(np.float16 if k_.endswith("embeddings") else v_.numpy().dtype)
        for k_, v_ in out.items():
            if v_ is not None:
                res[f"{tag}.out.{k_}"] = v_.detach().numpy()
    np.savez_compressed(f"{HERE}/g7_mouth_nets.npz", **res)


if __name__ == "__main__":
    cameras()
    eval_sh()
    losses()
    lr_schedule()
    sh_encoder_table()
    motion_nets()
    mouth_nets()
    print("golden fixtures written to", HERE)
