#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_t5.log 2>&1
rc=$?
tail -6 gpurun_out/r3_t5.log
[ $rc -eq 0 ] || exit $rc
python scripts/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_stages.log
python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"], d[\"windows_ms_per_step\"], d[\"kernels_us\"])"
