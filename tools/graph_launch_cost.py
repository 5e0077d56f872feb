#!/usr/bin/env python3
"""Host-side cost of replaying the step graph against the GPU time of a step (T1M defaults of bench.py):
how long the host needs to ENQUEUE one replay (loop without synchronisation, timed until the last call returns)
and how long the device needs per step (same loop timed to the final synchronize).  If the first is not well
below the second, the step is bound by the host's graph launch, not by the kernels.
usage: python tools/graph_launch_cost.py [--steps 300]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--rows', type=int, default=1000000)
    ap.add_argument('--batch', type=int, default=8192)
    ap.add_argument('--dims', default='256,256')
    a = ap.parse_args()
    import torch
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    from dp_cgans_amd.synthetic import build_encoded_table
    dev = torch.device('cuda', 0)
    dims = tuple(int(x) for x in a.dims.split(','))
    table, span = build_encoded_table('T1M', dev, n_rows=a.rows)
    eng = StepEngine(span, 128, dims, dims, a.batch, 8, dev, private=True, seed=0, dp_mode='dpsgd', dp_sigma=1.0, dp_clip=1.0)
    eng.sampler = DataSampler(table, span, device=dev, seed=17)
    g = GraphStep(eng, 1)
    for _ in range(20):
        g.run()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.run()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('host enqueue %.1f us per replay; device %.1f us per step (%d graph nodes per step)'
              % (1e6 * (t1 - t0) / a.steps, 1e6 * (t2 - t0) / a.steps, 0), flush=True)
    # one replay at a time (host waits for each): launch latency + device time of a single step
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.run()
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    print('one replay at a time, synchronised: %.1f us per step' % (1e6 * (t1 - t0) / a.steps))


if __name__ == '__main__':
    main()
