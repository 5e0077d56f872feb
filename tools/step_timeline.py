#!/usr/bin/env python3
"""Timeline of ONE graph-replayed training step from a rocprofv3 --kernel-trace CSV.

usage: step_timeline.py <..._kernel_trace.csv> [step index from the front, default 8] [d_steps, default 1]
A step ends with its last adam_kernel launch (tools/trace_steps.py).  Prints every kernel of the step with its start offset, duration,
queue and grid (blocks), then the time during which 0 / 1 / 2+ kernels were running."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_steps import step_ends  # noqa: E402


def main():
    path = sys.argv[1]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    d_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'),
                         int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y'])))
    rows.sort()
    ends = step_ends([r[2] for r in rows], d_steps)
    lo, hi = ends[k - 1] + 1, ends[k] + 1
    step = rows[lo:hi]
    t0 = rows[lo - 1][1]                     # end of the previous step's last kernel
    print('step of %d kernels, %.1f us from the end of the previous step to its own end' % (len(step), (step[-1][1] - t0) / 1e3))
    for s, e, name, q, gx, gy in step:
        short = name.split('(')[0].replace('void ', '')[:48]
        print('%8.1f %7.1f q%-2s %-48s %5d %3d' % ((s - t0) / 1e3, (e - s) / 1e3, q, short, gx, gy))
    ev = sorted([(s, 1) for s, e, *_ in step] + [(e, -1) for s, e, *_ in step])
    busy = {0: 0, 1: 0, 2: 0}
    depth, last = 0, t0
    for t, d in ev:
        busy[min(depth, 2)] += t - last
        depth += d
        last = t
    print('idle %.1f us, one kernel %.1f us, two or more %.1f us' % (busy[0] / 1e3, busy[1] / 1e3, busy[2] / 1e3))


if __name__ == '__main__':
    main()
