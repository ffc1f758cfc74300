#!/bin/bash
# round-3 first GPU pass: full GPU suite, then bench (all workloads)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_t1.log 2>&1
rc=$?
tail -15 gpurun_out/r3_t1.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 30 --warmup 5 > gpurun_out/r3_b1.log 2> gpurun_out/r3_b1.err
rc=$?
tail -5 gpurun_out/r3_b1.err; cat gpurun_out/r3_b1.log
exit $rc
