#!/usr/bin/env python3
"""Sum of a rocprofv3 --pmc counter per kernel name and grid (KiB counters -> MB; FETCH_SIZE x2 on gfx950)."""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[(r['Kernel_Name'].split('(')[0][-50:], r.get('Grid_Size', '?'))].append(float(r['Counter_Value']))
mul = 2.0 if 'FETCH' in sys.argv[1].upper() or (len(sys.argv) > 2 and sys.argv[2] == 'x2') else 1.0
tot = 0.0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    if 'gemm' in k[0]:
        mb = mul * sum(v) / len(v) / 1024
        tot += mb * len(v)
        print('%8.1f MB x%-3d %s grid %s' % (mb, len(v), k[0], k[1]))
print('total %.0f MB' % tot)
