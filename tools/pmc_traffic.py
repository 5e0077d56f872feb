#!/usr/bin/env python3
"""HBM-side traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
of `python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph`.

Units / corrections as MI355X_MICROARCH.md (HBM) prescribes: both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of 16-byte-per-lane coalesced reads (calibrated here on
activate_fwd_kernel, which reads the generator's raw output once: 4 x B x Dd bytes; earlier rounds used
interpolate_kernel, which the Gram-form penalty no longer launches), so it is doubled; WRITE_SIZE is
exact.  Writes profiles/<tag>_gemm_traffic.json, which bench.py reports as roofline.traffic.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']].append(float(r['Counter_Value']))
    return d


def main():
    F, W = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    is_gemm = lambda k: 'gemm_f32_kernel' in k or 'gemm_ksplit_kernel' in k
    gf = [v for k, vs in F.items() if is_gemm(k) for v in vs]
    gw = [v for k, vs in W.items() if is_gemm(k) for v in vs]
    cal = [v for k, vs in F.items() if 'activate_fwd_kernel' in k for v in vs]
    out = {
        'kernel': 'gemm_f32_kernel', 'launches': len(gf),
        'fetch_size_kib_avg': sum(gf) / len(gf), 'write_size_kib_avg': sum(gw) / len(gw),
        'fetch_correction': 2.0,
        'traffic_bytes_per_launch': (2.0 * sum(gf) / len(gf) + sum(gw) / len(gw)) * 1024.0,
        'calibration': {'kernel': 'activate_fwd_kernel', 'fetch_size_kib_avg': sum(cal) / len(cal) if cal else None,
                        'expected_read_bytes': 4 * 8192 * 335},
        'command': 'rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 3 --warmup 2 '
                   '--no-cpu-baseline --no-roofline --no-graph',
    }
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
