#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r3_bench_full.log 2> gpurun_out/r3_bench_full.err
rc=$?
grep "bench +" gpurun_out/r3_bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_full.log').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["windows_ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("kernel","achieved","frac","traffic","achieved_counter","walked_bytes","achieved_walked","avg_launch_us","empty_bracket_us")})
for k,v in d["roofline"]["blend_launches"].items(): print(" ", k, v["frac"], v["avg_launch_us"], v["achieved_counter"])
print("secondary", d["secondary_rooflines"])
print("kernels", d["kernels_us"])
for k in ("stable_targets","host_frames","reference_schedule","c3_phase_with_density_control","cpu_baseline"):
    v=d.get(k); 
    if v: print(k, {kk: vv for kk, vv in v.items() if kk not in ("workload","sample")})
PY
bash scripts/gpu/r03_stages.sh > gpurun_out/r3_stages_out.log 2>&1; head -12 gpurun_out/r3_stages_out.log
exit $rc
