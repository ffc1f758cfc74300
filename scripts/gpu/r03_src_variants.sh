#!/bin/bash
# usage: r03_src_variants.sh <source stem> "<flags A>" "<flags B>" ...  -> rebuild that source with each flag set, short bench
mkdir -p gpurun_out
stem=$1; shift
B="python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
for cfg in "$@"; do
  export INSTAG_EXTRA_FLAGS_${stem}="$cfg"
  python -m instag_amd.build > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  out=$(timeout -k 10 200 $B 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); k = d["kernels_us"]
print(f"{sys.argv[1]:44s} ms/step {d['ms_per_step']:.4f} {d['windows_ms_per_step']} wgrad {k.get('mlp_wgrad')} mlp_bwd {k.get('mlp_bwd')}", flush=True)
PY
done
