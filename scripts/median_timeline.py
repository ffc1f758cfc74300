#!/usr/bin/env python
"""Timeline of the replayed step whose duration is the median of all replayed steps of a rocprofv3 kernel trace.
    python scripts/median_timeline.py <trace dir>"""
import csv, glob, statistics, subprocess, sys, os
d = sys.argv[1]
f = (glob.glob(f'{d}/*/*_kernel_trace.csv') + glob.glob(f'{d}/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with its LAST optimizer launch: the one that no kernel of the same step follows (the next step begins with
# the frame's copy; a step in two optimizer launches has the first one in the middle of the backward pass)
ends = [i for i, r in enumerate(rows) if 'adam_step' in r['Kernel_Name']
        and (i + 1 == len(rows) or 'copyBuffer' in rows[i + 1]['Kernel_Name'])]
prev, info = -1, []
for k, e in enumerate(ends):
    sub = rows[prev + 1:e + 1]; prev = e
    info.append((k, len(sub), (int(sub[-1]['End_Timestamp']) - int(sub[0]['Start_Timestamp'])) / 1e3))
nk = statistics.mode(x[1] for x in info)
rep = [x for x in info if x[1] == nk]
med = statistics.median(x[2] for x in rep)
best = min(rep, key=lambda x: abs(x[2] - med))
print(f"# {len(rep)} replayed steps of {nk} kernels, median {med:.1f} us; showing step {best[0]}")
sys.stdout.flush()
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "timeline.py"), d, str(best[0])])
