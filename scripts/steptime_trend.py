#!/usr/bin/env python
"""Step-time trend of a long run from a rocprofv3 kernel trace: per chunk of steps (counted by adam_step launches) the
mean step window, the mean sum of kernel durations, and the mean duration of a few named kernels.
    python scripts/steptime_trend.py <dir> [chunk]"""
import csv, glob, sys
d = sys.argv[1]; chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = (glob.glob(f'{d}/*/*_kernel_trace.csv') + glob.glob(f'{d}/*_kernel_trace.csv'))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:200]))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'adam_step' in r[2]]
names = ['blend_backward_kernel', 'blend_forward_kernel', 'mlp_forward_kernel', 'preprocess_kernel', 'adam_step_kernel']
print('steps', len(ends))
print('chunk  step_window_us  kernel_sum_us  ' + '  '.join(n[:14] for n in names))
for c0 in range(1, len(ends), chunk):
    c1 = min(c0 + chunk, len(ends))
    if c1 - c0 < chunk // 2: break
    win = (rows[ends[c1 - 1]][1] - rows[ends[c0 - 1]][1]) / (c1 - c0) / 1e3
    sub = rows[ends[c0 - 1] + 1: ends[c1 - 1] + 1]
    ksum = sum(e - s for s, e, _ in sub) / (c1 - c0) / 1e3
    per = []
    for n in names:
        ds = [e - s for s, e, k in sub if n in k]
        per.append(sum(ds) / max(1, len(ds)) / 1e3)
    print(f'{c0:5d}  {win:10.1f}  {ksum:10.1f}  ' + '  '.join(f'{p:12.1f}' for p in per))
