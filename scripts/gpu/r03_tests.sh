#!/bin/bash
# usage: r03_tests.sh <log tag> <pytest args...>
set -o pipefail
mkdir -p gpurun_out
tag=$1; shift
python -m pytest "$@" -m gpu -x -q -p no:cacheprovider > gpurun_out/${tag}.log 2>&1
rc=$?
tail -60 gpurun_out/${tag}.log
exit $rc
