#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_t6.log 2>&1
rc=$?
tail -6 gpurun_out/r3_t6.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu/r03_timeline.sh > gpurun_out/r3_tl_out.log 2>&1
grep -n "bench\|replayed" gpurun_out/r3_tl.log gpurun_out/r3_timeline.txt | head; sed -n 28,50p gpurun_out/r3_timeline.txt | cut -c1-100
python bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"], d[\"windows_ms_per_step\"], d[\"kernels_us\"])"
python scripts/check_determinism.py 2>&1 | tail -3
