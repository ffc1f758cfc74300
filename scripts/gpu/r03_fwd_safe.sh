#!/bin/bash
# usage: r03_fwd_safe.sh "<flags>" ...  -> like r03_fwd_variants.sh, every bench under its own short timeout, progress kept
mkdir -p gpurun_out
B="python bench.py --steps 30 --warmup 5 --windows 3 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule"
i=0
for cfg in "$@"; do
  i=$((i+1))
  export INSTAG_EXTRA_FLAGS_raster_blend="$cfg"
  timeout -k 10 300 python -m instag_amd.build > gpurun_out/fwd_safe_build_$i.log 2>&1 || { echo "build failed: $cfg"; continue; }
  echo "built: $cfg" | tee -a gpurun_out/fwd_safe_progress.log
  timeout -k 10 100 $B > gpurun_out/fwd_safe_$i.json 2> gpurun_out/fwd_safe_$i.err
  rc=$?
  echo "bench rc=$rc: $cfg" | tee -a gpurun_out/fwd_safe_progress.log
  if [ $rc -ne 0 ]; then tail -3 gpurun_out/fwd_safe_$i.err; echo "STOP: no further GPU step after a failed one"; exit 1; fi
  python - "$cfg" gpurun_out/fwd_safe_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); k = d["kernels_us"]
print(f"{sys.argv[1]:44s} ms/step {d['ms_per_step']:.4f} {d['windows_ms_per_step']} blend_fwd {k.get('blend_fwd')}", flush=True)
PY
done
