#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python scripts/bench_stages.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_stages.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_mouth_prof -- python3 $R/scripts/probes/mouth_profile.py > $R/gpurun_out/r3_mouth_prof.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_fuse_prof -- python3 $R/scripts/probes/fuse_profile.py > $R/gpurun_out/r3_fuse_prof.log 2>&1
cd $R
python scripts/stage_timeline.py gpurun_out/r3_mouth_prof 1 > gpurun_out/r3_mouth_timeline.txt 2>&1; head -80 gpurun_out/r3_mouth_timeline.txt
python scripts/stage_timeline.py gpurun_out/r3_fuse_prof 1 > gpurun_out/r3_fuse_timeline.txt 2>&1; head -3 gpurun_out/r3_fuse_timeline.txt
rm -rf gpurun_out/r3_mouth_prof gpurun_out/r3_fuse_prof
