#!/usr/bin/env python3
"""idle time between consecutive kernels of the replayed sampling step, from a rocprofv3 --kernel-trace CSV:
python tools/gap_report.py <dir with *_kernel_trace.csv> [min_kernels_per_step]
Prints, for the last replayed step found (the run of kernels between two step_update kernels), the sum of kernel times, the sum
of the gaps and the gaps by (previous kernel -> next kernel) pair."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
print(f"{len(rows)} kernels in the trace")
# steps = segments ending with the step-update kernel
idx = [i for i, r in enumerate(rows) if "step_update_kernel" in r[2]]
if len(idx) < 3:
    sys.exit("no replayed steps found")
a, b = idx[-2] + 1, idx[-1] + 1
seg = rows[a:b]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = collections.Counter()
cnt = collections.Counter()
import re
short = lambda n: (re.findall(r"(\w+_kernel\w*|\w+)\s*(?:<|\()", n) or [n])[0][:40]
for (s0, e0, n0), (s1, e1, n1) in zip(seg, seg[1:]):
    g = s1 - e0
    gaps[(short(n0), short(n1))] += g
    cnt[(short(n0), short(n1))] += 1
tot = sum(gaps.values())
print(f"step of {len(seg)} kernels: span {span/1e6:.3f} ms, kernels {busy/1e6:.3f} ms, gaps {tot/1e6:.3f} ms ({tot/span*100:.2f} %), "
      f"mean gap {tot/max(1,len(seg)-1)/1e3:.1f} us")
for (k, g) in gaps.most_common(18):
    print(f"  {g/1e3:8.1f} us over {cnt[k]:3d}  {k[0]} -> {k[1]}")
