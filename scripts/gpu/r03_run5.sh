#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_t5.log 2>&1
rc=$?
tail -6 gpurun_out/r3_t5.log
[ $rc -eq 0 ] || exit $rc
python scripts/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_stages.log
python scripts/bench_infer.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_infer.log
