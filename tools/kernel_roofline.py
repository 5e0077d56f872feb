#!/usr/bin/env python3
"""Per-kernel roofline table of the graph-replayed training step, from the rocprofv3 records of ONE command:

  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o bench -- python3 bench.py --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph
  rocprofv3 --pmc WRITE_SIZE -- (same)

usage: kernel_roofline.py --trace <bench_kernel_trace.csv> --bench <stdout of that bench run (its JSON line)>
                          --stats-name <file name of the kernel_stats.csv as committed under profiles/>
                          [--fetch <counter_collection.csv> --write <counter_collection.csv>] --out profiles/r02_kernel_roofline.json

The trace is cut into training steps at each step's last adam_kernel launch (tools/trace_steps.py); the `steps` timed
replays of the bench run (after its warm-up) are averaged: launches per step and average duration of every kernel
as it runs INSIDE the step graph, beside the graph's side branches.  Algorithmic work per kernel follows DESIGN.md
section 4 (bytes each kernel must move once, flops of the products); achieved = work / time, frac = achieved / peak
(HBM 8 TB/s, fp32 MFMA 157.3 TF/s: MI355X_MICROARCH.md).  HBM-side traffic of the products: FETCH_SIZE x 2 +
WRITE_SIZE in KiB per dispatch (the guide's gfx950 correction: FETCH_SIZE counts 64 B per 128-B request).
bench.py quotes these figures (``in_graph_trace``, ``traffic``) only while source_sha matches the current sources.
"""
import argparse
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, source_sha  # noqa: E402
from trace_steps import step_ends  # noqa: E402
from pmc_reduce import rows_of  # noqa: E402


def family(name):
    n = name.replace('void ', '')
    n = n.split('(')[0]
    if n.startswith('gemm_f32_kernel') or n.startswith('gemm_ksplit_kernel') or n.startswith('gemm_group_kernel'):
        return 'gemm'
    if n.startswith('at::native') or n.startswith('__amd_rocclr'):
        return 'torch fill / copy (zero pools)'
    return n.strip()


def base_name(fam):
    """Kernel family without its template arguments."""
    return re.sub(r'<.*$', '', fam)


def algorithmic_bytes_per_step(c):
    """bytes every kernel family must move once per training step (d_steps discriminator updates +
    1 generator update).  Keys: the family name as the trace prints it, or - where the template arguments do not change
    the bytes - the name without them (``lookup`` tries both)."""
    B, Dd, C, pac = c['batch'], c['data_dim'], c['cond_dim'], c['pac']
    n = B // pac
    g, d = c['gen_dims'], c['disc_dims']
    K0x = c['k0_compact']
    ds = int(c.get('d_steps', 1))
    nd = c['n_discrete']
    E = c['embedding_dim']
    d0, dl = d[0], d[-1]
    clip = 4.0 * sum(3 * n * w for w in d + [1]) + 4.0 * (2 * n * K0x + sum(3 * n * w for w in d[:-1] + [d[-1]])) + 16.0 * n * d0
    rows = c.get('table_rows', 0)
    words = (rows + 63) // 64
    S = 64
    act_f, act_t, act_b = 8.0 * B * Dd, 12.0 * B * Dd, 16.0 * B * Dd
    m = {
        # raw output in, activated out (the discriminator updates' fake rows); + the cond-loss gradient written; backward
        'activate_fwd_small_kernel<false>': ds * act_f, 'activate_fwd_chunk_kernel<false>': ds * act_f,
        'activate_fwd_tiled_kernel<false>': ds * act_f, 'activate_fwd_wide_kernel<false>': ds * act_f,      # (kernels of rounds 1-2)
        'activate_fwd_small_kernel<true>': act_t, 'activate_fwd_chunk_kernel<true>': act_t,
        'activate_fwd_tiled_kernel<true>': act_t, 'activate_fwd_wide_kernel<true>': act_t,
        'activate_bwd_tiled_kernel': act_b, 'activate_bwd_wide_kernel': act_b,
        'persample_gram_all_kernel': ds * clip,
        # (+ the cond slices' transposed gradient addend in and transposed copy out, moved by the same launch)
        'adam_kernel_disc': ds * (28.0 * c['disc_params'] + 8.0 * pac * C * d0),
        'adam_kernel_gen': 28.0 * c['gen_params'] + 8.0 * C * (sum(g) + Dd),
        'row_gather_kernel': ds * 8.0 * B * Dd,
        # k-th set bit of bitmap_a & bitmap_b: the S superblock counts, then ONE superblock of both bitmaps
        'row_select_kernel': ds * B * (4.0 * S + 2 * 8.0 * ((words + S - 1) // S)),
        'perm_feistel_kernel': ds * 8.0 * B,
        'bn_relu_fwd_fused_kernel': (ds + 1) * sum(8.0 * B * w for w in g),      # every generator forward of the step
        'bn_relu_bwd_fused_kernel': sum(16.0 * B * w for w in g),
        'colreduce_kernel<2>': sum(12.0 * B * w for w in g),
        'disc_out_kernel': 12.0 * (ds * 3 * n + n) * dl,
        'gp_gram_rows_kernel': ds * 16.0 * n * d0,
        'disc_first_act_kernel': 4.0 * d0 * (ds * ((2 * n + 2 * 3 * n + 2 * n) + 2 * pac * 2 * n) + (n + 2 * n) + 2 * pac * n),
        'condvec_pair_kernel': (ds + 1) * (4.0 * B * 2 + 4.0 * B * nd),
        # the updates' inputs: cond vector twice (own copy + the generator-input columns), mask, latent batch, and the
        # zero fill of the accumulator pools (flat gradients, transposed cond-column gradients, Gram products)
        'step_inputs_kernel': (ds + 1) * (8.0 * B * C + 4.0 * B * nd + 4.0 * B * E)
                              + 4.0 * (ds * (c['disc_params'] + 2 * d0 * d0 + pac * C * d0) + c['gen_params'] + C * (sum(g) + Dd)),
        # cond columns of the weight gradients: one delta row read per (row, hot id) pair - 4B pairs of d0 floats in a
        # discriminator update, 2B pairs per generator Linear - plus one atomic row per id change (~pairs / 64 + C rows)
        'hot_rows_sum_kernel': 4.0 * (ds * (4 * B * d0 + (4 * B // 64 + pac * C) * d0)
                                      + sum(2 * B * w + (2 * B // 64 + C) * w for w in g + [Dd])),
        # counting sort of the (row, id) pairs: ids read twice (count, place), (row, id) written, the per-id counters
        'hot_sort_kernel': 4.0 * (ds * (3 * 4 * B + 2 * pac * C) + 3 * 2 * B + 2 * C),
        # bias gradients + the head's weight gradient: every delta matrix of the update read once
        'colsum_multi_partial_kernel': 4.0 * (ds * (sum(2 * n * w for w in d) + 3 * n * dl + 2 * n) + B * (sum(g) + Dd)),
        'colsum_multi_finish_kernel': 8.0 * 64 * (ds * (sum(d) + dl + 1) + sum(g) + Dd),
        'colreduce_finish_kernel': 8.0 * 4 * 2 * sum(g) * 2,
        'transpose_slices_kernel': 8.0 * (ds * pac * C * d0 + C * (sum(g) + Dd)),
        'rng_advance_kernel': 64.0,
    }
    return m


def lookup(alg, fam):
    if fam in alg:
        return alg[fam]
    return alg.get(base_name(fam))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--trace', required=True)
    ap.add_argument('--bench', required=True)
    ap.add_argument('--stats-name', required=True)
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--mfma', help='counter_collection.csv of the SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass')
    ap.add_argument('--out', required=True)
    ap.add_argument('--command', default='rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline')
    a = ap.parse_args()

    line = [l for l in open(a.bench) if l.startswith('{')][-1]
    bench = json.loads(line)
    cfg = bench['config']
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    ends = step_ends([r[2] for r in rows], cfg.get('d_steps', 1))
    warm, steps = max(bench['warmup'], 2), bench['steps']
    # step k of the run = rows (ends[k-1], ends[k]]; the timed replays are steps warm .. warm+steps-1
    per = collections.defaultdict(lambda: [0, 0.0])
    span_us = 0.0
    for k in range(warm, warm + steps):
        lo, hi = ends[k - 1] + 1, ends[k] + 1
        span_us += (rows[hi - 1][1] - rows[lo - 1][1]) / 1e3
        for j, (s, e, name) in enumerate(rows[lo:hi]):
            fam = family(name)
            if fam.startswith('adam_kernel'):
                fam = 'adam_kernel_gen' if lo + j == hi - 1 else 'adam_kernel_disc'    # the step's last kernel
            per[fam][0] += 1
            per[fam][1] += (e - s) / 1e3
    alg = algorithmic_bytes_per_step(cfg)
    kernels = {}
    for fam, (cnt, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        ent = {'launches_per_step': cnt / steps, 'avg_us': us / cnt, 'us_per_step': us / steps}
        if fam == 'gemm':
            gf = bench['roofline']['gflop_per_step']
            ach = gf * 1e9 / (us / steps * 1e-6) / 1e12
            ent.update(bound='mfma', gflop_per_step=gf, achieved=ach, unit='TFLOP/s', peak=FP32_MFMA_PEAK_TFLOPS,
                       frac=ach / FP32_MFMA_PEAK_TFLOPS,
                       algorithmic_bytes_per_launch=bench['roofline']['algorithmic_bytes_per_launch'])
        elif lookup(alg, fam) is not None:
            by = lookup(alg, fam)
            ach = by / (us / steps * 1e-6) / 1e9
            ent.update(bound='hbm', algorithmic_bytes_per_step=by, algorithmic_bytes_per_launch=by / (cnt / steps), achieved=ach,
                       unit='GB/s', peak=HBM_PEAK_GBS, frac=ach / HBM_PEAK_GBS)
            if by / (cnt / steps) < 1e6:      # (said only where the modelled bytes really are small)
                ent['note'] = 'under 1 MB per launch: bounded by launch + dependency latency, not by bandwidth'
        else:
            ent.update(bound='unmodelled', note='no byte model for this kernel (tools/kernel_roofline.py)')
        kernels[fam] = ent
    if a.fetch and a.write:
        # every family's HBM-side traffic per launch (adam: both launches' average); the counter files may be the raw
        # per-dispatch CSV or its per-kernel sums (tools/pmc_reduce.py)
        def by_family(path, counts=None):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for name, _, n, v in rows_of(path):
                f = family(name)
                acc[f][0] += v
                acc[f][1] += n
            if counts is not None:
                counts.update({f: v[1] for f, v in acc.items()})
            return {f: v[0] / v[1] for f, v in acc.items()}
        launches = {}
        ff, fw = by_family(a.fetch, launches), by_family(a.write)
        for fam, ent in kernels.items():
            key = 'adam_kernel<true>' if fam.startswith('adam_kernel_') else fam
            if fam != 'gemm' and key in ff and key in fw:
                ent['traffic_bytes_per_launch'] = (2.0 * ff[key] + fw[key]) * 1024.0
                if 'algorithmic_bytes_per_launch' in ent and ent['algorithmic_bytes_per_launch'] > 0:
                    ent['traffic_over_algorithmic'] = ent['traffic_bytes_per_launch'] / ent['algorithmic_bytes_per_launch']
        kernels['gemm'].update(
            traffic_bytes_per_launch=(2.0 * ff['gemm'] + fw['gemm']) * 1024.0,
            fetch_size_kib_avg=ff['gemm'], write_size_kib_avg=fw['gemm'], fetch_correction=2.0,
            traffic_launches=launches['gemm'],
            traffic_command='rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py --steps 3 --warmup 2 '
                            '--no-cpu-baseline --no-roofline --no-graph')
        kernels['gemm']['traffic_over_algorithmic'] = (kernels['gemm']['traffic_bytes_per_launch']
                                                       / kernels['gemm']['algorithmic_bytes_per_launch'])
    if a.mfma:
        # matrix-core busy share of the products: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs
        # (64 per v_mfma_f32_32x32x2_f32: the sum equals flops / 64 for full tiles), GRBM_GUI_ACTIVE the busy cycles
        # summed over the 8 XCDs (MI355X_MICROARCH.md)
        acc = collections.defaultdict(float)
        for name, counter, _, v in rows_of(a.mfma):
            if family(name) == 'gemm':
                acc[counter] += v
        kernels['gemm'].update(
            mfma_busy_cycles=acc['SQ_VALU_MFMA_BUSY_CYCLES'], gui_active_cycles_sum_xcd=acc['GRBM_GUI_ACTIVE'],
            mfma_util=acc['SQ_VALU_MFMA_BUSY_CYCLES'] / (acc['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0),
            wave_cycles={k: acc[k] for k in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_BUSY_CU_CYCLES')},
            mfma_command='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY '
                         'SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 2 '
                         '--no-cpu-baseline --no-roofline --no-graph (launches serialised by the counter collection)')
    out = {
        'command': a.command,
        'workload': cfg['workload'], 'source_sha': source_sha(), 'stats_csv': a.stats_name,
        'steps_averaged': steps, 'step_us_in_trace': span_us / steps, 'bench_ms_per_step_under_rocprof': bench['ms_per_step'],
        'kernel_us_per_step': sum(v['us_per_step'] for v in kernels.values()),
        'kernels': kernels,
    }
    json.dump(out, open(a.out, 'w'), indent=1)
    for fam, v in kernels.items():
        print('%-40s %5.1f x %7.1f us  %s' % (fam, v['launches_per_step'], v['avg_us'],
                                             ('%8.1f %s  frac %.3f' % (v['achieved'], v['unit'], v['frac'])) if 'achieved' in v else ''))
    print('step %.1f us in trace; sum of kernel time %.1f us' % (out['step_us_in_trace'], out['kernel_us_per_step']))


if __name__ == '__main__':
    main()
