#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_tl -- python3 $R/bench.py --steps 30 --warmup 5 --windows 5 --no-stable-targets --no-cpu-baseline --no-host-frames --no-schedule > $R/gpurun_out/r3_tl.log 2>&1
cd $R
python scripts/median_timeline.py gpurun_out/r3_tl > gpurun_out/r3_timeline.txt 2>&1
cat gpurun_out/r3_timeline.txt | cut -c1-120
rm -rf gpurun_out/r3_tl
