#!/usr/bin/env python3
"""Benchmark of the DP-CGAN training step on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[2] - synthetic 1M-row table (20 continuous + 30
categorical columns), per-GPU batch 8192, pac 8, G/D dims (256,256), private=True, one D update +
one G update per step.  A "step" is that pair of updates over one freshly sampled batch; value =
rows/s over all ranks (weak scaling: every rank draws its own 8192-row batch shard).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--table', default='T1M')
    ap.add_argument('--rows', type=int, default=None, help='override the number of table rows (debug)')
    ap.add_argument('--batch', type=int, default=8192)
    ap.add_argument('--pac', type=int, default=8)
    ap.add_argument('--dims', default='256,256')
    ap.add_argument('--d-steps', type=int, default=1)
    ap.add_argument('--dp-mode', default='dpsgd', choices=['dpsgd', 'reference'],
                    help="private=True semantics: north_star DP-SGD (per-example clip + Gaussian noise) or the "
                         "reference's grad-hook noise + weight clamp")
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying HIP graphs')
    ap.add_argument('--serial', action='store_true', help='no side streams: the step is one chain of kernels (tuning)')
    ap.add_argument('--bn-sync', type=int, default=0, choices=[0, 1],
                    help='N > 1 (weak scaling: every rank trains on a batch of the reference\'s own size): 0 = per-rank '
                         'BatchNorm statistics, what DistributedDataParallel computes - 2 all-reduces per step (default); '
                         '1 = statistics over the global batch, the step of DPCGANSynthesizer.fit\'s default, which splits '
                         'ONE batch over the ranks and promises the single-process result - 6 all-reduces per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--devices', default=None,
                    help='GPU index of each local rank, comma separated (default: local rank r computes on GPU r); '
                         '"0,0" rehearses two ranks on one GPU (then over gloo: RCCL needs a GPU per rank)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launch plumbing only: the ranks form their process group, meet at a barrier, agree on a '
                         'number and rank 0 prints a JSON line - no GPU is touched (runs on a CPU-only host over gloo)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Started directly (`python bench.py --gpus N`): this process becomes the launcher.  It starts one fresh
        # child per GPU BEFORE anything here has touched the GPU (torch is not even imported yet), relays rank 0's
        # JSON line and exits with the first non-zero child status.
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    from dp_cgans_amd.parallel import Dist

    if args.dry_run:
        os.environ.setdefault('DPCGAN_DIST_BACKEND', 'gloo')
    dev_index = int(os.environ.get('LOCAL_RANK', '0'))
    if args.devices:
        devs = [int(x) for x in args.devices.split(',')]
        if len(devs) != args.gpus:
            raise SystemExit('--devices names %d GPUs for %d ranks' % (len(devs), args.gpus))
        dev_index = devs[dev_index]
        if len(set(devs)) < len(devs):
            os.environ.setdefault('DPCGAN_DIST_BACKEND', 'gloo')      # ranks sharing a GPU cannot form an RCCL group
    dist = Dist(device='cpu' if args.dry_run else 'cuda:%d' % dev_index)
    if dist.world != args.gpus:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d: run it directly (it starts its own ranks) '
                         'or under torch.distributed.run --nproc-per-node %d' % (args.gpus, dist.world, args.gpus))
    if args.dry_run:
        dist.barrier()
        agreed = dist.max_float(float(dist.rank), 'cpu')
        ok = agreed == float(dist.world - 1)
        dist.barrier()
        if dist.rank == 0:
            print(json.dumps({'dry_run': True, 'n_gpus': dist.world, 'backend': dist.backend, 'max_rank_seen': agreed}))
        dist.shutdown()
        raise SystemExit(0 if ok else 1)

    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    from dp_cgans_amd.synthetic import build_encoded_table

    dev = torch.device('cuda', dev_index)
    torch.cuda.set_device(dev)
    dims = tuple(int(x) for x in args.dims.split(','))

    def progress(what):
        # one line per phase on stderr: a wide workload spends minutes between start and the JSON line
        if dist.rank == 0:
            sys.stderr.write('bench.py: %s (%.1f s)\n' % (what, time.time() - t_start))
            sys.stderr.flush()

    t_start = t0 = time.time()
    table, span = build_encoded_table(args.table, dev, n_rows=args.rows)
    progress('table encoded')
    sampler = DataSampler(table, span, device=dev, seed=17 + dist.rank)
    progress('sampler tables built')
    eng = StepEngine(span, 128, dims, dims, args.batch, args.pac, dev, private=True, seed=0, dist=dist,
                     dp_mode=args.dp_mode, dp_sigma=1.0, dp_clip=1.0, bn_sync=bool(args.bn_sync))
    eng.sampler = sampler
    if args.serial:
        eng.concurrent = False
    torch.cuda.synchronize()
    setup_s = time.time() - t0

    def eager_step():
        eng.train_step(args.d_steps)

    gstep = GraphStep(eng, args.d_steps)
    step = eager_step if args.no_graph else gstep.run
    for _ in range(max(args.warmup, 2)):      # (first call eager, second captures the graph)
        step()
    torch.cuda.synchronize()
    progress('warm-up done')
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = dist.max_float(time.perf_counter() - t0, dev)
    progress('timed steps done')
    losses = eng.losses.cpu().tolist()
    rows = args.batch * dist.world * args.steps
    args.workload = workload_string(args, table, dims)
    out = {
        'metric': 'DP-CGAN training rows/sec (D+G steps)', 'value': rows / elapsed, 'unit': 'rows/s',
        'n_gpus': dist.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': args.workload,
                   'global_batch': args.batch * dist.world, 'parallelism': 'dp%d' % dist.world,
                   'batchnorm': ('statistics over the global batch' if eng.bn_sync else 'per-rank statistics') if dist.active else 'one rank',
                   'launch': 'eager' if args.no_graph else 'hipGraph replay (%d graph segment(s)%s)' % (
                       sum(1 for k, _ in (gstep.items or []) if k == 'graph'),
                       '' if not dist.active else (', RCCL all-reduces captured in the graph' if gstep.collectives_in_graph
                                                   else ', %d eager all-reduces between them' % sum(1 for k, _ in gstep.items if k != 'graph'))),
                   'table_rows': int(table.shape[0]), 'd_steps': args.d_steps,
                   'data_dim': eng.Dd, 'cond_dim': eng.C, 'n_discrete': eng.nd, 'batch': args.batch, 'pac': args.pac,
                   'gen_dims': list(dims), 'disc_dims': list(dims), 'embedding_dim': eng.E,
                   'k0_compact': getattr(eng, 'K0x', eng.K0), 'disc_params': eng.disc.n, 'gen_params': eng.gen.n,
                   'setup_s': round(setup_s, 2),
                   'final_losses': [round(x, 5) for x in losses]},
    }

    if dist.rank == 0 and not args.no_roofline:
        if dist.world == 1:
            out.update(step_kernel_rooflines(eng, eager_step, args))
            progress('per-kernel roofline figures done')
        out.update(standalone_dp_kernels(eng, dev))
        progress('stand-alone DP kernels done')

    if dist.rank == 0 and dist.world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(eng, span, args)
        progress('CPU baseline done')

    dist.barrier()
    if dist.rank == 0:
        print(json.dumps(out))
    dist.shutdown()


def launch_ranks(n):
    """`python bench.py --gpus N` started directly: N child processes of this same command line, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would set; the children
    see WORLD_SIZE and run as ranks).  The launcher itself never initialises the GPU - no torch import, no HIP call -
    and nothing is exec'ed: children are fresh processes.  Rank 0's stdout (the ONE JSON line) is relayed; the other
    ranks' stdout is dropped, every rank's stderr passes through.  If a rank fails, the others are ended (they would
    wait for it in the next collective) and its exit status is returned."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL's intra-node transport needs here
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc, out0 = 0, None
    live = set(range(n))
    while live:
        for r in sorted(live):
            try:
                if r == 0 and out0 is None:
                    out0, _ = procs[0].communicate(timeout=0.2)
                else:
                    procs[r].wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            live.discard(r)
            if procs[r].returncode != 0 and rc == 0:
                rc = procs[r].returncode
                sys.stderr.write('bench.py: rank %d exited with status %d; ending the other ranks\n' % (r, rc))
                for q in sorted(live):
                    procs[q].terminate()         # exactly the processes started above
    # rank 0's stdout carries the JSON line - and whatever a library chose to print there (gloo announces its
    # connections on stdout): only the JSON goes to this command's stdout, the rest to stderr
    for line in (out0 or '').splitlines():
        (sys.stdout if line.lstrip().startswith('{') else sys.stderr).write(line + '\n')
    sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc


def source_sha():
    """Identity of the code a profile was taken with: kernels + the host code that decides the launches."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, 'dp_cgans_amd', 'csrc', '*.hip')))
    files += [os.path.join(ROOT, 'dp_cgans_amd', f) for f in ('engine.py', 'ops.py')]
    for f in files:
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:12]


def workload_string(args, table, dims):
    return ('%s synthetic table (%d rows, %d encoded columns), per-GPU batch %d, pac %d, G/D dims %s, private=True (%s), '
            'discriminator_steps %d; BASELINE.json configs[%d]'
            % (args.table, table.shape[0], table.shape[1], args.batch, args.pac, dims,
               'DP-SGD: per-example clip C=1 + Gaussian noise sigma=1' if args.dp_mode == 'dpsgd'
               else 'reference hook-noise + weight clamp', args.d_steps, {'T1M': 2, 'T10M': 3, 'TWIDE': 4}.get(args.table, 2)))


def trace_figures(workload):
    """Per-kernel figures of the rocprofv3 kernel trace + PMC passes of THIS command (the step replayed as one graph
    with its side branches: tools/profile_round.sh, tools/kernel_roofline.py -> profiles/r03_kernel_roofline*.json,
    one file per profiled workload).  They are records of earlier runs, so one is only quoted when it was taken on
    this workload with the current sources; otherwise None."""
    import glob
    sha = source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r03_kernel_roofline*.json'))):
        if path.endswith('_serial.json'):
            continue
        rec = json.load(open(path))
        if rec.get('workload') != workload or rec.get('source_sha') != sha:
            continue
        # the same step traced as ONE chain (bench.py --serial): its kernels never overlap, so their durations are
        # the kernels' own - the trace figure that the live chain replay below has to agree with
        spath = path[:-len('.json')] + '_serial.json'
        if os.path.exists(spath):
            srec = json.load(open(spath))
            if srec.get('source_sha') == sha and srec.get('workload') == workload:
                rec['serial'] = srec
        rec['file'] = os.path.basename(path)
        return rec
    return None


def step_kernel_rooflines(eng, eager_step, args):
    """Roofline figures of the kernels the timed step runs, measured live: the launches of one training step are
    logged (identical arguments and buffers), and each kernel family's launches are issued again inside a
    captured HIP graph of their own - one chain on one stream, so the graph's duration is the sum of the kernel
    durations, with no host launch gaps - between two HIP events.  In the real step the same kernels share the
    chip with the step's side branches; that figure comes from the rocprofv3 trace (``in_graph_trace``)."""
    import torch
    from dp_cgans_amd._lib import lib
    L = lib()
    dev = eng.dev
    eng.concurrent = False            # one stream: the log is in issue order
    L.log = []
    eager_step()
    torch.cuda.synchronize()
    log, L.log = L.log, None
    eng.concurrent = True

    def timed(entries, copies, reps=10):
        """average duration (us) of one pass over ``entries``"""
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            g.capture_begin()
            h = torch.cuda.current_stream(dev).cuda_stream
            for _ in range(copies):
                for e in entries:
                    L.replay(e, h)
            g.capture_end()
        torch.cuda.current_stream(dev).wait_stream(side)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / (reps * copies)

    res = {}
    trace = trace_figures(args.workload)
    tk = (trace or {}).get('kernels', {})
    # ---- the dominant kernel: the fp32 MFMA products (gemm_f32_kernel / gemm_ksplit_kernel, all variants)
    gemms = [e for e in log if e[0] in ('dpc_gemm_f32_ex', 'dpc_gemm_f32_group')]
    flops = sum(e[2]['flops'] for e in gemms)
    # launches: one per product, except that the products of a grouped call share a launch per kernel variant
    # (operand layouts + k-scale: dpc_gemm_f32_group)
    nprod = sum(len(e[2]['shapes']) for e in gemms)
    nlaunch = sum(len(e[2]['shapes']) if e[0] == 'dpc_gemm_f32_ex' else len(set(sh[3:] for sh in e[2]['shapes'])) for e in gemms)
    alg_bytes = sum(e[2]['bytes'] for e in gemms) / nlaunch
    us = timed(gemms, copies=4)
    achieved = flops / (us * 1e-6) / 1e12
    if os.environ.get('DPC_GEMM_BREAKDOWN'):      # per-shape launch times of one step, to stderr
        for i, e in enumerate(gemms):
            t = timed([e], copies=20, reps=3)
            m, n, k, ta, tb = e[2]['shapes'][0][:5]
            sys.stderr.write('%2d M=%5d N=%5d K=%6d ta=%d tb=%d x%d  %7.1f us %6.1f TF/s\n'
                             % (i, m, n, k, ta, tb, len(e[2]['shapes']), t, 1e-6 * e[2]['flops'] / t))
    tg = tk.get('gemm')
    res['roofline'] = {
        'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': achieved / FP32_MFMA_PEAK_TFLOPS,
        'traffic': tg['traffic_bytes_per_launch'] if tg else None,
        'traffic_unit': 'HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes of this workload: '
                        'tools/profile_round.sh, tools/kernel_roofline.py)',
        'algorithmic_bytes_per_launch': alg_bytes,
        'kernel': 'gemm_f32_kernel + gemm_group_kernel + gemm_ksplit_kernel (all tile variants)', 'launches_per_step': nlaunch,
        'products_per_step': nprod,
        'gflop_per_launch': flops / nlaunch / 1e9, 'avg_launch_us': us / nlaunch,
        'gflop_per_step': flops / 1e9, 'gemm_ms_per_step': us / 1e3,
        'method': "one step's GEMM launches issued again as one chain inside a captured graph (kernel time, no "
                  'host gaps), HIP events',
        'in_graph_trace': None if not tg else {
            'achieved': tg['achieved'], 'frac': tg['frac'], 'avg_launch_us': tg['avg_us'],
            'launches_per_step': tg['launches_per_step'],
            'source': 'profiles/%s + profiles/%s (rocprofv3 --kernel-trace --stats of this command: the launches '
                      'inside the replayed step graph, beside its side branches)' % (trace['stats_csv'], trace['file'])},
    }
    ts = (trace or {}).get('serial', {}).get('kernels', {}).get('gemm')
    if ts:
        res['roofline']['serial_graph_trace'] = {
            'achieved': ts['achieved'], 'frac': ts['frac'], 'avg_launch_us': ts['avg_us'],
            'source': 'profiles/%s (the same command with --serial: the step graph as one chain, no kernel beside '
                      'the products)' % trace['serial']['stats_csv']}
    # ---- the DP kernels as the step runs them (HBM-bound; at this size their buffers sit in the 256 MB
    # Infinity Cache / L2, and n = B / pac examples do not fill the chip: latency, not bandwidth, bounds them)
    n, M = eng.n, eng.M
    clip = [e for e in log if e[0] in ('dpc_persample_clip_all_gram', 'dpc_persample_clip_all')]
    if clip:
        gram = clip[0][0].endswith('_gram')
        d = eng.ddim
        k_in = [eng.K0x if gram else eng.K0] + d[:-1] + [d[-1]]
        rows_in = [2 * n if gram else 3 * n] + [3 * n] * M
        by = 4.0 * sum(3 * n * w for w in d + [1]) + 4.0 * sum(r * w for r, w in zip(rows_in, k_in))
        if gram:
            by += 4.0 * 2 * (2 * n * d[0])             # the first layer's plain products: read, then cleared
        us = timed(clip[:1], copies=20)
        gbs = by / (us * 1e-6) / 1e9
        t = tk.get('persample_gram_all_kernel')
        res['dp_clip_norm_in_step'] = {
            'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
            'kernel': 'persample_gram_all_kernel', 'examples': n, 'algorithmic_bytes': by, 'launch_us': us,
            'in_graph_trace': None if not t else {'achieved': t['achieved'], 'frac': t['frac'], 'avg_launch_us': t['avg_us']}}
    adam = [e for e in log if e[0] in ('dpc_adam_step', 'dpc_adam_step_slices')]
    if adam and eng.dp_mode == 'dpsgd':
        dsc = eng.disc
        saved = [x.clone() for x in (dsc.p, dsc.m, dsc.v, dsc.t)]
        by = 28.0 * dsc.n                               # p, g, m, v read; p, m, v written; the noise is drawn in registers
        if adam[0][0].endswith('_slices'):              # + the cond slices' transposed gradient addend in and copy out
            by += 8.0 * eng.pac * eng.C * eng.ddim[0]
        us = timed(adam[:1], copies=20)
        for x, y in zip((dsc.p, dsc.m, dsc.v, dsc.t), saved):
            x.copy_(y)
        dsc.loaded = True
        gbs = by / (us * 1e-6) / 1e9
        t = tk.get('adam_kernel_disc')
        res['dp_noise_in_step'] = {
            'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
            'kernel': 'adam_kernel<true> (Gaussian noise add fused into the discriminator Adam step; with the cond slices of W1 moved in the same launch)',
            'parameters': dsc.n, 'algorithmic_bytes': by, 'launch_us': us,
            'in_graph_trace': None if not t else {'achieved': t['achieved'], 'frac': t['frac'], 'avg_launch_us': t['avg_us']}}
    eng.Zp.zero_()               # the replayed first-layer products added into it
    torch.cuda.synchronize()
    return res


def standalone_dp_kernels(eng, dev):
    """The DP clip / noise kernels on buffer sets larger than the Infinity Cache (north_star's >= 60 % of the HBM
    roofline is stated for these): what the kernels reach when the example count fills the chip."""
    import torch
    from dp_cgans_amd import ops
    out = {}
    # DP noise/clamp kernel (HBM-bound): x = clamp(x + s*N(0,1)), 8 B/element algorithmic
    n_el = 1 << 28
    x = torch.zeros(n_el, device=dev)
    for _ in range(3):
        ops.noise_add_clamp(x, 1e-3, rng=eng.rng_shared, clamp=0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.noise_add_clamp(x, 1e-3, rng=eng.rng_shared, clamp=0.01)
    e1.record()
    torch.cuda.synchronize()
    gbs = 8.0 * n_el * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out['dp_noise_clamp'] = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': gbs / HBM_PEAK_GBS, 'elements': n_el, 'kernel': 'noise_add_clamp_kernel (stand-alone)'}
    del x
    # DP clip kernel (per-example Gram / ghost norm, HBM-bound): 3n rows of K0 + d floats read once
    K0, dd = eng.K0, eng.ddim[0]
    nn = 16384
    act = torch.randn(3 * nn, K0, device=dev)
    dl = torch.randn(3 * nn, dd, device=dev)
    n2 = torch.zeros(nn, dtype=torch.float64, device=dev)
    for _ in range(2):
        ops.persample_gram(dl, act, nn, n2)
    e0.record()
    for _ in range(5):
        ops.persample_gram(dl, act, nn, n2)
    e1.record()
    torch.cuda.synchronize()
    gbs = 4.0 * 3 * nn * (K0 + dd) * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out['dp_clip_norm'] = {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': gbs / HBM_PEAK_GBS, 'examples': nn, 'row_floats': K0 + dd,
                           'kernel': 'persample_gram_kernel (stand-alone)'}
    return out


def cpu_baseline(eng, span, args):
    """The CPU oracle (oracle/nets.py: the reference's op sequence on torch-CPU autograd, closed-form
    cond loss because the reference's per-pair loop cannot run 9453 pairs) timed on the host cores
    for a bounded sample: a few full D+G steps of the same workload, inputs taken from the device
    sampler.  Reported only; it is not the optimisation target."""
    import torch
    from oracle import nets as onets
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    B, n, Dd, pac = eng.B, eng.n, eng.Dd, eng.pac
    gp = {k: v.cpu() for k, v in eng.gen.state_dict().items()}
    dp = {k: v.cpu() for k, v in eng.disc.state_dict().items()}
    cond, mask = eng.cond.cpu(), eng.mask.cpu()
    real = eng.Xd[B:2 * B, :Dd].cpu()
    g = torch.Generator().manual_seed(0)
    t_total, steps = 0.0, 0
    sd, sg = {'t': 0, 'm': {}, 'v': {}}, {'t': 0, 'm': {}, 'v': {}}
    while t_total < 12.0 and steps < 60:
        z = torch.randn(B, eng.E, generator=g)
        e = torch.empty(B, Dd).exponential_(generator=g)
        perm = torch.randperm(B, generator=g)
        mk = lambda rows: [(torch.rand(rows, w, generator=g) > 0.5).to(torch.uint8) for w in eng.ddim]
        alpha = torch.rand(n, generator=g)
        t0 = time.perf_counter()
        for _ in range(args.d_steps):
            out = onets.d_step_grads(gp, dp, span, pac, z, cond, perm, real, e, mk(n), mk(n), mk(n), alpha)
            dp = onets.apply_update(dp, out['grads'], sd, clamp=0.01)
        out = onets.g_step_grads(gp, dp, span, pac, z, cond, mask, e, mk(n), loop_loss=False)
        gp = onets.apply_update(gp, out['grads'], sg)
        t_total += time.perf_counter() - t0
        steps += 1
    return {'value': B * steps / t_total, 'unit': 'rows/s', 'cores': cores, 'kind': 'port',
            'sample': '%d full D+G steps at batch %d (torch-CPU autograd restatement of the reference step: '
                      "its private=True is hook noise + weight clamp, not per-example clipping; closed-form cond "
                      'loss, host sampling excluded), %.1f s' % (steps, B, t_total)}


if __name__ == '__main__':
    main()
