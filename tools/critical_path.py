#!/usr/bin/env python3
"""Walk the critical chain of one graph-replayed step backwards from its last kernel: the predecessor of a kernel is
the kernel that ended last before it started (what it waited for, whatever the queue).  Gaps are printed.
usage: critical_path.py <kernel_trace.csv> [step index from the end, default 3]"""
import csv
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:44], r['Queue_Id']))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith('rng_advance_kernel')]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
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
