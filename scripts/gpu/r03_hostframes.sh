#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python scripts/probes/host_frames_probe.py 60 2>&1 | tee gpurun_out/r3_hf_plain.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_hf_prof -- python3 $GRAFT_REPO_ROOT/scripts/probes/host_frames_probe.py 30 > $GRAFT_REPO_ROOT/gpurun_out/r3_hf_prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r3_hf_prof.log
find gpurun_out/r3_hf_prof -name "*.csv" | head
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r3_hf_prof/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    by = collections.defaultdict(list)
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        by[(r.get("Direction"), )].append(d)
    for k, v in by.items():
        v.sort()
        print(k, "n", len(v), "median us", v[len(v)//2]/1e3, "max us", v[-1]/1e3, "sum ms", sum(v)/1e6)
    big = [r for r in rows if int(r.get("Bytes", r.get("bytes", 0)) or 0) > 1000000] if rows and ("Bytes" in rows[0] or "bytes" in rows[0]) else []
    print("copies > 1 MB:", len(big))
    for r in big[-12:]:
        print(r)
PY
