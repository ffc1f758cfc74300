#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
echo "== system runtime" > gpurun_out/r3_event_probe.log
./scripts/probes/event_node_probe >> gpurun_out/r3_event_probe.log 2>&1
echo "== torch's runtime ($TL)" >> gpurun_out/r3_event_probe.log
LD_LIBRARY_PATH=$TL ./scripts/probes/event_node_probe >> gpurun_out/r3_event_probe.log 2>&1
cat gpurun_out/r3_event_probe.log
INSTAG_CONCURRENT_FUSE=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r3_fuse_on.log 2>&1
rc=$?
tail -12 gpurun_out/r3_fuse_on.log
exit $rc
