#!/usr/bin/env python
"""Per-step kernel categories over the hipGraph-replayed steps of a rocprofv3 kernel trace of bench.py.
    python scripts/category_summary.py <trace dir>
Steps are delimited by adam_step launches; eagerly launched steps (warm-up, the instrumented pass) take about twice
the time of a replayed one, so the replayed steps are those within 1.25x of the 10th-percentile step window."""
import collections, csv, glob, re, sys
d = sys.argv[1]
f = (glob.glob(f'{d}/*/*_kernel_trace.csv') + glob.glob(f'{d}/*_kernel_trace.csv'))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
ends = [i for i, r in enumerate(rows) if 'adam_step' in r[2]]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    sub = rows[a + 1:b + 1]
    steps.append((sub[-1][1] - rows[a][1], sub))
w10 = sorted(w for w, _ in steps)[len(steps) // 10]
fast = [(w, sub) for w, sub in steps if w < 1.25 * w10]
n = len(fast)
def cat(name):
    m = re.search(r'instag::\(anonymous namespace\)::(\w+)', name)
    if m: return 'instag:' + m.group(1)
    if 'CountAtRank' in name or 'rocprim' in name: return 'rocprim (scan, radix sorts)'
    if 'rocclr' in name: return 'memset/copy'
    if 'at::native' in name: return 'aten elementwise/reduce/cat'
    return 'other:' + name[:40]
acc = collections.defaultdict(lambda: [0, 0.0])
for w, sub in fast:
    for s, e, k in sub:
        c = cat(k); acc[c][0] += 1; acc[c][1] += (e - s) / 1e3
win = sum(w for w, _ in fast) / n / 1e3
ksum = sum(v[1] for v in acc.values()) / n
print(f"hipGraph-replayed steps ({n} of {len(steps)} traced) of `rocprofv3 --kernel-trace -- python3 bench.py`: "
      f"window {win:.1f} us/step, kernels/step {sum(v[0] for v in acc.values()) / n:.1f}, "
      f"sum of kernel durations {ksum:.1f} us/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:45s} launches/step {v[0] / n:5.1f}   us/step {v[1] / n:8.1f}")
