#!/bin/bash
# usage: r03_knobs.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one short bench per setting
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
for setting in "$@"; do
  out=$(env $setting $B 2>/dev/null | tail -1)
  python - "$setting" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d["kernels_us"]
print(f"{sys.argv[1]:45s} ms/step {d['ms_per_step']:.4f}  windows {d['windows_ms_per_step']}  mlp_bwd {k.get('mlp_bwd')} wgrad {k.get('mlp_wgrad')} grid_bwd {k.get('grid_bwd')} blend_fwd {k.get('blend_fwd')} replay: {d['roofline'].get('kernel')} {d['roofline'].get('avg_launch_us')}")
PY
done
