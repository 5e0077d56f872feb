#!/usr/bin/env python3
"""Shrinks a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter, ~450 bytes each: tens of MB for
a wide workload) to one row per (kernel, counter): Kernel_Name, Counter_Name, Dispatches, Counter_Value (the SUM over
the dispatches).  tools/kernel_roofline.py reads either form.
usage: pmc_reduce.py <pmc_counter_collection.csv> [--keep]     -> pmc_counter_summary.csv beside it; the raw file is
removed unless --keep."""
import collections
import csv
import os
import sys


def rows_of(path):
    """(kernel name, counter name, dispatches, summed value) from either form."""
    for r in csv.DictReader(open(path)):
        yield r['Kernel_Name'], r['Counter_Name'], int(r.get('Dispatches', 1)), float(r['Counter_Value'])


def main():
    path = sys.argv[1]
    acc = collections.OrderedDict()
    for k, c, n, v in rows_of(path):
        e = acc.setdefault((k, c), [0, 0.0])
        e[0] += n
        e[1] += v
    out = os.path.join(os.path.dirname(path), 'pmc_counter_summary.csv')
    with open(out, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(['Kernel_Name', 'Counter_Name', 'Dispatches', 'Counter_Value'])
        for (k, c), (n, v) in acc.items():
            w.writerow([k, c, n, v])
    if '--keep' not in sys.argv:
        os.remove(path)
    print('%s: %d rows' % (out, len(acc)))


if __name__ == '__main__':
    main()
