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
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    import torch
    from dp_cgans_amd import ops
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    from dp_cgans_amd.parallel import Dist
    from dp_cgans_amd.synthetic import build_encoded_table

    dist = Dist()
    if dist.world != args.gpus:
        if args.gpus != 1 or dist.world != 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d seen)' % (args.gpus, dist.world))
    dev = torch.device('cuda', dist.local_rank)
    torch.cuda.set_device(dev)
    dims = tuple(int(x) for x in args.dims.split(','))

    t0 = time.time()
    table, span = build_encoded_table(args.table, dev, n_rows=args.rows)
    sampler = DataSampler(table, span, device=dev, seed=17 + dist.rank)
    eng = StepEngine(span, 128, dims, dims, args.batch, args.pac, dev, private=True, seed=0, dist=dist,
                     dp_mode=args.dp_mode, dp_sigma=1.0, dp_clip=1.0)
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
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = dist.max_float(time.perf_counter() - t0, dev)
    losses = eng.losses.cpu().tolist()
    rows = args.batch * dist.world * args.steps
    out = {
        'metric': 'DP-CGAN training rows/sec (D+G steps)', 'value': rows / elapsed, 'unit': 'rows/s',
        'n_gpus': dist.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s synthetic table (%d rows, %d encoded columns), per-GPU batch %d, '
                               'pac %d, G/D dims %s, private=True (%s), discriminator_steps %d; BASELINE.json configs[%d]'
                               % (args.table, table.shape[0], table.shape[1], args.batch, args.pac, dims,
                                  'DP-SGD: per-example clip C=1 + Gaussian noise sigma=1' if args.dp_mode == 'dpsgd'
                                  else 'reference hook-noise + weight clamp', args.d_steps,
                                  {'T1M': 2, 'T10M': 3, 'TWIDE': 4}.get(args.table, 2)),
                   'global_batch': args.batch * dist.world, 'parallelism': 'dp%d' % dist.world,
                   'launch': 'eager' if args.no_graph else 'hipGraph replay (%d segment(s))' % len(gstep.items or []),
                   'data_dim': eng.Dd, 'cond_dim': eng.C, 'n_discrete': eng.nd, 'setup_s': round(setup_s, 2),
                   'final_losses': [round(x, 5) for x in losses]},
    }

    if dist.rank == 0 and not args.no_roofline:
        # Live per-launch timing of the dominant kernel (gemm_f32_kernel, all tile variants): HIP
        # events on the launch stream around every GEMM of a further `steps` training steps.
        ops.GEMM_PROFILE = []
        nprof = min(args.steps, 20)
        if dist.world == 1:
            eng.concurrent = False       # one stream: every GEMM's events bracket that GEMM alone
            for _ in range(nprof):
                eager_step()
            torch.cuda.synchronize()
            eng.concurrent = True
            prof = ops.GEMM_PROFILE
            ops.GEMM_PROFILE = None
            if os.environ.get('DPC_GEMM_BREAKDOWN'):      # per-shape launch times of one step, to stderr
                per = {}
                for i, (e0, e1, f, shp, _) in enumerate(prof):
                    per.setdefault((i % (len(prof) // nprof), shp), []).append(e0.elapsed_time(e1) * 1e3)
                for (i, shp), v in sorted(per.items()):
                    us = sum(v) / len(v)
                    sys.stderr.write('%2d M=%5d N=%5d K=%5d ta=%d tb=%d  %7.1f us %6.1f TF/s\n'
                                     % (i, shp[0], shp[1], shp[2], shp[3], shp[4], us, 2e-6 * shp[0] * shp[1] * shp[2] / us))
            ms_each = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in prof)     # every launch bracketed on its own
            flops = sum(f for _, _, f, _, _ in prof)
            alg_bytes = sum(4.0 * (m * k + k * n + m * n) for _, _, _, (m, n, k, _, _), _ in prof) / len(prof)
            # The dominant kernel's launches of ONE step issued again back to back (identical arguments and
            # buffers) between two events: the per-launch figure above also counts the stream's idle time
            # between an event record and the launch it brackets (a few us each on an eager stream), this one
            # is kernel time (memset / deferred epilogue of a split product included), which is what the
            # rocprofv3 kernel trace of the same command reports.
            per_step = len(prof) // nprof
            calls = [c for _, _, _, _, c in prof[:per_step]]
            reps = 20
            for c in calls:
                c()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                for c in calls:
                    c()
            e1.record()
            torch.cuda.synchronize()
            eng.Zp.zero_()               # the replayed first-layer products added into it
            ms = e0.elapsed_time(e1) / reps * nprof
            achieved = flops / (ms * 1e-3) / 1e12
            # HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of this workload
            # (tools/pmc_traffic.py, gfx950 correction applied); only valid for the default workload
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'r01_gemm_traffic.json')
            default_wl = (args.table == 'T1M' and args.rows is None and args.batch == 8192 and args.pac == 8
                          and args.dims == '256,256' and args.d_steps == 1 and args.dp_mode == 'dpsgd')
            if default_wl and os.path.exists(tpath):
                traffic = json.load(open(tpath))['traffic_bytes_per_launch']
            out['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                               'traffic_unit': 'bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, PMC passes)',
                               'algorithmic_bytes_per_launch': alg_bytes,
                               'kernel': 'gemm_f32_kernel', 'launches_per_step': len(prof) / nprof,
                               'gflop_per_launch': flops / len(prof) / 1e9,
                               'avg_launch_us': 1e3 * ms / len(prof),
                               'gflop_per_step': flops / nprof / 1e9, 'gemm_ms_per_step': ms / nprof,
                               'avg_launch_us_bracketed_singly': 1e3 * ms_each / len(prof),
                               'method': 'one step\'s GEMM launches replayed back to back, %d repetitions, HIP events' % reps}
        ops.GEMM_PROFILE = None
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
                                 'frac': gbs / HBM_PEAK_GBS, 'elements': n_el}
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
                               'kernel': 'persample_gram_kernel'}
        del act, dl, n2

    if dist.rank == 0 and dist.world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(eng, span, args)

    dist.barrier()
    if dist.rank == 0:
        print(json.dumps(out))
    dist.shutdown()


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
    while t_total < 10.0 and steps < 20:
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
