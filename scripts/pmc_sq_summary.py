#!/usr/bin/env python
"""Fold a rocprofv3 --pmc pass of SQ counters into per-kernel means and the shares that say what bounds a kernel.

    python scripts/pmc_sq_summary.py <counter_collection.csv> <out.csv>

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md "rocprofv3
PMC slots"): ACTIVE_INST_ANY + WAIT_ANY + WAIT_INST_ANY ~= WAVE_CYCLES.  valu_share = ACTIVE_INST_VALU / WAVE_CYCLES
is the fraction of a resident wave's life spent issuing vector ALU work."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    if "instag" in name and m:
        return m.group(1) + (m.group(2) or "")
    return None


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in acc.values() for c in v})
    with open(sys.argv[2], "w") as fh:
        fh.write("kernel,launches," + ",".join(names) + ",valu_share,wait_share,issue_stall_share\n")
        for k, v in sorted(acc.items()):
            mean = {c: sum(x) / len(x) for c, x in v.items()}
            wc = mean.get("SQ_WAVE_CYCLES", 0.0) or float("nan")
            fh.write(f"\"{k}\",{len(next(iter(v.values())))}," + ",".join(f"{mean.get(c, float('nan')):.0f}" for c in names)
                     + f",{mean.get('SQ_ACTIVE_INST_VALU', float('nan')) / wc:.3f}"
                       f",{mean.get('SQ_WAIT_ANY', float('nan')) / wc:.3f}"
                       f",{mean.get('SQ_WAIT_INST_ANY', float('nan')) / wc:.3f}\n")


if __name__ == "__main__":
    main()
