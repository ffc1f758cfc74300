#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python bench.py --steps 4 --warmup 2 --windows 1 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule > gpurun_out/r3_b3.log 2> gpurun_out/r3_b3.err
rc=$?
grep -n "instag prof\|bench" gpurun_out/r3_b3.err | head -80
exit $rc
