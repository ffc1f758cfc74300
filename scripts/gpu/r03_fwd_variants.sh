#!/bin/bash
# usage: r03_fwd_variants.sh "<flags A>" "<flags B>" ...  -> rebuild raster_blend with each flag set, short bench (no tests)
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 5 --windows 3 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
for cfg in "$@"; do
  export INSTAG_EXTRA_FLAGS_raster_blend="$cfg"
  python -m instag_amd.build > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  out=$($B 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); k = d["kernels_us"]
print(f"{sys.argv[1]:44s} ms/step {d['ms_per_step']:.4f} {d['windows_ms_per_step']} blend_fwd {k.get('blend_fwd')} blend_bwd {k.get('blend_bwd')}")
PY
done
