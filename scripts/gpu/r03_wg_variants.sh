#!/bin/bash
# A/B of weight-gradient kernel variants: rebuild mlp.hip with the given -D flags on the box, test, bench
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
for cfg in "$@"; do
  export INSTAG_EXTRA_FLAGS_mlp="$cfg"
  python -m instag_amd.build > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  python -m pytest tests/test_mlp_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1
  out=$($B 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); k = d["kernels_us"]
print(f"{sys.argv[1]:42s} ms/step {d['ms_per_step']:.4f} {d['windows_ms_per_step']} wgrad {k.get('mlp_wgrad')} mlp_bwd {k.get('mlp_bwd')} grid_bwd {k.get('grid_bwd')}")
PY
done
