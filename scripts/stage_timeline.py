#!/usr/bin/env python
"""Timeline of the median replayed step of a stage trainer's trace (two optimizer launches per step).
    python scripts/stage_timeline.py <trace dir> [adam launches per step = 2]"""
import csv, glob, re, statistics, sys
d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = (glob.glob(f'{d}/*/*_kernel_trace.csv') + glob.glob(f'{d}/*_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adam_step' in r['Kernel_Name']]
segs, prev = [], -1
for k in range(per - 1, len(ends), per):
    segs.append(rows[prev + 1:ends[k] + 1]); prev = ends[k]
cnt = statistics.mode(len(s) for s in segs)
rep = [s for s in segs if len(s) == cnt]
dur = [(int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp'])) / 1e3 for s in rep]
med = statistics.median(dur)
best = rep[min(range(len(rep)), key=lambda i: abs(dur[i] - med))]
t0 = int(best[0]['Start_Timestamp'])
print(f"# {len(rep)} replayed steps of {cnt} kernels, median {med:.1f} us")
for r in best:
    n = r['Kernel_Name']; m = re.search(r'(\w+_kernel)', n)
    n2 = ('instag:' + m.group(1)) if 'instag' in n and m else n[:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r.get('Queue_Id', '?')} {n2}")
