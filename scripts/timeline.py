#!/usr/bin/env python
"""Print one step's kernel timeline from a rocprofv3 kernel trace: start offset, duration, queue, name.
    python scripts/timeline.py <dir> <step index (counted by adam_step launches)>"""
import csv, glob, re, sys
d = sys.argv[1]; step = int(sys.argv[2])
f = (glob.glob(f'{d}/*/*_kernel_trace.csv') + glob.glob(f'{d}/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with its LAST optimizer launch: the one that no kernel of the same step follows (the next step begins with
# the frame's copy; a step in two optimizer launches has the first one in the middle of the backward pass)
ends = [i for i, r in enumerate(rows) if 'adam_step' in r['Kernel_Name']
        and (i + 1 == len(rows) or 'copyBuffer' in rows[i + 1]['Kernel_Name'])]      # one launch per step
a = ends[step - 1] + 1 if step > 0 else 0
b = ends[step] + 1
sub = rows[a:b]
t0 = int(sub[0]['Start_Timestamp'])
def short(n):
    m = re.search(r'instag::\(anonymous namespace\)::(\w+)', n)
    if m: return 'instag:' + m.group(1)
    n = re.sub(r'void |at::native::|\(anonymous namespace\)::', '', n)
    return n[:70]
last_end = t0
for r in sub:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{r['Queue_Id']:>2}  {short(r['Kernel_Name'])}")
print('total us', (int(sub[-1]['End_Timestamp']) - t0) / 1e3, 'kernels', len(sub))
