#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
INSTAG_CONCURRENT_FUSE=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_fuse_on.log 2>&1
rc=$?
tail -8 gpurun_out/r3_fuse_on.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 30 --warmup 5 --windows 2 --no-stable-targets --no-cpu-baseline > gpurun_out/r3_b2.log 2> gpurun_out/r3_b2.err
rc=$?
tail -20 gpurun_out/r3_b2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_b2.log').read().strip().splitlines()[-1])
for k in ("value","ms_per_step","kernel_durations_from","kernels_us","host_frames","reference_schedule","c3_phase_with_density_control"):
    print(k, json.dumps(d.get(k)))
PY
exit $rc
