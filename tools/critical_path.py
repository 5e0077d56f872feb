#!/usr/bin/env python3
"""Walk the critical chain of one graph-replayed step backwards from its last kernel: the predecessor of a kernel is
the kernel that ended last before it started (what it waited for, whatever the queue).  Gaps are printed.
usage: critical_path.py <kernel_trace.csv> [step index from the front, default 8] [d_steps, default 1]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_steps import step_ends  # noqa: E402

path = sys.argv[1]
kth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:44], r['Queue_Id']))
rows.sort()
ends = step_ends([r[2] for r in rows], d_steps)
lo, hi = ends[kth - 1] + 1, ends[kth] + 1
step = rows[lo:hi]
t0 = rows[lo - 1][1]
cur = step[-1]
chain = []
while True:
    chain.append(cur)
    prev = [k for k in step if k[1] <= cur[0] + 300 and k is not cur and k[0] < cur[0]]
    if not prev:
        break
    cur = max(prev, key=lambda k: k[1])
tot = 0.0
for k in reversed(chain):
    print('%8.1f %7.1f  q%s %s' % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
    tot += (k[1] - k[0]) / 1e3
print('chain of %d kernels, %.1f us of kernel time in a %.1f us step' % (len(chain), tot, (step[-1][1] - t0) / 1e3))
