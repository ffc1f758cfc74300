#!/bin/bash
# usage: r03_flag_variants.sh <source stem> "<flags A>" "<flags B>" ...  -> rebuild that source with each flag set, raster tests, bench
mkdir -p gpurun_out
stem=$1; shift
B="python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
for cfg in "$@"; do
  export INSTAG_EXTRA_FLAGS_${stem}="$cfg"
  python -m instag_amd.build > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  python -m pytest tests/test_raster_gpu.py -m gpu -x -q -p no:cacheprovider -k "forward_images or backward_gradients or binning or long_walks" 2>&1 | tail -1
  out=$($B 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); k = d["kernels_us"]
print(f"{sys.argv[1]:36s} ms/step {d['ms_per_step']:.4f} {d['windows_ms_per_step']} blend_fwd {k.get('blend_fwd')} blend_bwd {k.get('blend_bwd')}")
PY
done
