#!/usr/bin/env python
"""Instructions per (Gaussian, 64-pixel wave) pair in the blend kernels' inner loops, counted in the gfx950 ISA.

    python scripts/isa_loop_count.py [profiles/r02_blend_isa_counts.json]

Compiles instag_amd/csrc/raster_blend.hip to assembly with the build's own flags, finds in every blend kernel the
innermost loop that walks the staged records (forward: groups of four Gaussians; backward phase A: unrolled by four)
and counts the instructions between the loop header and its back edge by class.  bench.py's `roofline_valu` uses the
VALU count (a wave64 VALU instruction occupies its SIMD-32 for 2 cycles, a transcendental for 4); the other classes say
how far a lone wave (one instruction of ANY class per ~4 cycles) is from that.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instag_amd import build as B   # noqa: E402

TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32")
PER_ITER = 4      # Gaussians per trip of the counted loops (forward: a group; backward: unroll factor)


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def loops(lines):
    """-> [(header label, depth, first line, back-edge line)] of loops whose back edge is a branch to the header."""
    out = []
    label = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label = m.group(1)
        m = re.search(r"This Inner Loop Header: Depth=(\d+)", ln)
        if m and label:
            depth = int(m.group(1))
            for j in range(i, len(lines)):
                if re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\b", lines[j]):
                    out.append((label, depth, i, j))
                    break
    return out


def count(lines, lo, hi):
    c = {}
    for ln in lines[lo:hi + 1]:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        k = classify(t.split()[0])
        c[k] = c.get(k, 0) + 1
    return c


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    src = os.path.join(B.CSRC, "raster_blend.hip")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "blend.s")
        flags = [f for f in B.COMMON if f != "-fPIC"] + B.SOURCES["raster_blend.hip"]
        subprocess.run([B.hipcc()] + flags + ["--cuda-device-only", "-S", "-o", asm, src], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(asm).read().splitlines()
    # split into functions
    funcs, cur, start = {}, None, 0
    for i, ln in enumerate(text):
        m = re.match(r"^(_ZN6instag\S+):", ln)
        if m:
            if cur:
                funcs[cur] = (start, i)
            cur, start = m.group(1), i
    if cur:
        funcs[cur] = (start, len(text))
    result = {}
    for name, (lo, hi) in funcs.items():
        m = re.search(r"\d+(blend_\w+?_kernel)I((?:L[bi]\d+E)+)E", name)
        if not m:
            continue
        targs = [("true" if v == "1" else "false") if t == "b" else v for t, v in re.findall(r"L([bi])(\d+)E", m.group(2))]
        short = f"{m.group(1)}<{', '.join(targs)}>"
        body = text[lo:hi]
        ls = loops(body)
        if not ls:
            continue
        deepest = max(d for _, d, _, _ in ls)
        inner = [x for x in ls if x[1] == deepest]
        # the walk: of the innermost loops that evaluate an exponential, the longest (the segment-wise forward kernel has
        # a second, shorter one: its transmittance-only pass)
        with_exp = [x for x in inner if any("v_exp_f32" in ln for ln in body[x[2]:x[3] + 1])] or inner[:1]
        label, depth, a, b = max(with_exp, key=lambda x: sum(count(body, x[2], x[3]).values()))
        c = count(body, a, b)
        per = {k: round(v / PER_ITER, 2) for k, v in sorted(c.items())}
        per["all"] = round(sum(c.values()) / PER_ITER, 2)
        result[short] = {"loop": label, "gaussians_per_trip": PER_ITER, "per_gaussian": per}
    js = json.dumps(result, indent=1)
    print(js)
    if out_path:
        with open(out_path, "w") as f:
            f.write(js + "\n")


if __name__ == "__main__":
    main()
