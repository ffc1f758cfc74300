"""Build libinstag_hip.so (gfx950) in-tree with hipcc.

    python -m instag_amd.build [--force]

The shared library is written to ``instag_amd/lib/libinstag_hip.so``; objects under
``instag_amd/lib/obj``.  Both are git-ignored but travel to the GPU box with the snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libinstag_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
          "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function",
          "-DNDEBUG"]

# per-file extra flags.  raster_preprocess must not fuse mul+add: its integer outputs (radii, tile
# rectangles, sort keys) are checked bit-for-bit against the CPU oracle.
SOURCES = {
    "raster_preprocess.hip": ["-ffp-contract=off"],
    # raster_blend: SLP pairing of scalars ACROSS Gaussians costs a dozen v_mov per group of four in the forward loop
    "raster_blend.hip": ["-fno-slp-vectorize"],
    "raster_backward.hip": [],
    "raster_api.hip": [],
    "raster_sort.hip": [],
    "grid.hip": [],
    "sh.hip": [],
    "mlp.hip": [],
    "ssim.hip": [],
    "glue.hip": [],
    "adam.hip": [],
    "audio.hip": [],
    "knn.hip": [],
    "select.hip": [],
    "prior.hip": [],
}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build libinstag_hip.so)")


def _digest(paths, flags):
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(ROOT, "include", "instag_hip.h"))
    return hs


def _compile(cc, name, flags, force):
    # (experiments: INSTAG_EXTRA_FLAGS_<stem>="-DFOO=1 ..." adds flags to one source, e.g. INSTAG_EXTRA_FLAGS_mlp)
    flags = flags + os.environ.get("INSTAG_EXTRA_FLAGS_" + name.split(".")[0], "").split()
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJDIR, name.replace(".hip", ".o"))
    stamp = obj + ".sha"
    dig = _digest([src] + _headers(), COMMON + flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [cc, "-c", src, "-o", obj] + COMMON + flags
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> str:
    cc = hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    names = [n for n in SOURCES if os.path.exists(os.path.join(CSRC, n))]
    with ThreadPoolExecutor(max_workers=min(6, len(names))) as ex:
        results = list(ex.map(lambda n: _compile(cc, n, SOURCES[n], force), names))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[instag_amd.build] linked {LIB}")
    elif verbose:
        print(f"[instag_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
