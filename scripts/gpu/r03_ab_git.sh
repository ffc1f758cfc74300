#!/bin/bash
# usage: r03_ab_git.sh <source file relative to repo> -> headline bench with the committed version of that source (from
# .ab_prev/<basename>, put there before the call: git show HEAD:<file> > .ab_prev/<basename>) and with the working-tree version, twice each, same box
mkdir -p gpurun_out
f=$1; b=$(basename $f)
B="python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
cp $f .ab_prev/new_$b
run() {
  python -m instag_amd.build > /dev/null 2>&1 || { echo "build failed ($1)"; return; }
  out=$(timeout -k 10 200 $B 2>/dev/null | tail -1)
  python - "$1" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); k = d["kernels_us"]
print(f"{sys.argv[1]:10s} ms/step {d['ms_per_step']:.4f} {d['windows_ms_per_step']}", flush=True)
PY
}
for rep in 1 2; do
  cp .ab_prev/$b $f; touch $f; run prev
  cp .ab_prev/new_$b $f; touch $f; run new
done
