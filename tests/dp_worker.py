"""Worker of the data-parallel tests (one process per rank; launched by test_parallel_*.py).

mode 'replay': every rank replays its shard of the reference trace (rows [r*B/R, (r+1)*B/R) of every
injected tensor) through StepEngine with the collectives of parallel.Dist; rank 0 checks gradients
and weights against the single-process values recorded from the reference.
mode 'graph': Philox-driven steps, eager engine vs graph-segment engine on every rank.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def shard(x, r, R, groups=False):
    n = x.shape[0] // R
    return x[r * n:(r + 1) * n]


def replay(dist, dev):
    from golden_util import Trace
    from dp_cgans_amd.engine import StepEngine
    from golden_util import AdamAwareCompare
    from test_engine_gpu import close, injected
    tr = Trace('trace_np.npz')
    R, r = dist.world, dist.rank
    Bl = tr.B // R
    eng = StepEngine(tr.span_info, tr.E, tr.gen_dim, tr.disc_dim, Bl, tr.pac, dev, private=False, dist=dist,
                     global_batch=tr.B)
    eng.disc.load(tr.params(tr.kinds.index('D'), 'p_before'))
    eng.gen.load(tr.params(tr.kinds.index('G'), 'p_before'))
    cmp_d, cmp_g = AdamAwareCompare(), AdamAwareCompare()
    for s, kind in enumerate(tr.kinds):
        full = injected(tr, s, dev)
        n = tr.B // tr.pac
        inj = {k: shard(full[k], r, R) for k in ('z', 'cond', 'mask', 'exp')}
        # dropout masks are stacked [fake groups | real groups | interpolate groups]: shard each block
        nblk = 3 if kind == 'D' else 1
        inj['keep'] = [torch.cat([shard(m[b * n:(b + 1) * n], r, R) for b in range(nblk)]) for m in full['keep']]
        if kind == 'D':
            inj['real'] = shard(full['real'], r, R)
            inj['real_cond'] = shard(full['cond'][full['perm']], r, R)
            inj['alpha'] = shard(full['alpha'], r, R)
            eng.d_step(inj)
            ref_g, ref_p = tr.params(s, 'g'), tr.params(s, 'p_after')
            cmp_d.steps += 1
            losses = eng.global_losses()
            if r == 0:
                for k in tr.dkeys:
                    close(eng.disc.G(k), ref_g[k], rtol=2e-5, what='s%d D grad %s' % (s, k))
                    cmp_d.check(k, eng.disc.P(k), ref_p[k], ref_g[k], 's%d D param %s' % (s, k))
                close(losses[1:2], [float(tr.get(s, 'pen'))], what='pen')
                ysc = max(abs(float(tr.get(s, 'd_out', 0).mean())), abs(float(tr.get(s, 'd_out', 1).mean())))
                close(losses[0:1], [float(tr.get(s, 'backward_value', 1))], scale=ysc, what='loss_d')
        else:
            eng.g_step(inj)
            ref_g, ref_p = tr.params(s, 'g'), tr.params(s, 'p_after')
            cmp_g.steps += 1
            losses = eng.global_losses()
            if r == 0:
                close(losses[3:4], [float(tr.get(s, 'ce'))], what='ce')
                close([losses[2] + losses[3]], [float(tr.get(s, 'backward_value', 0))], what='loss_g')
                for k in tr.gkeys:
                    sc = np.abs(ref_g[k.replace('fc.bias', 'fc.weight')]).max() if 'fc.bias' in k else None
                    close(eng.gen.G(k), ref_g[k], rtol=2e-5, scale=sc, what='s%d G grad %s' % (s, k))
                    if 'fc.bias' not in k:
                        cmp_g.check(k, eng.gen.P(k), ref_p[k], ref_g[k], 's%d G param %s' % (s, k))
    # replicas stay identical
    a = eng.disc.p.clone()
    dist.allreduce(a)
    assert torch.allclose(a, eng.disc.p * R, rtol=0, atol=0), 'discriminator replicas diverged'
    sd = eng.generator_state_dict()
    if r == 0:
        for k in tr.z.files:
            if k.startswith('gen_final/') and 'running' in k:
                sc = None       # as in test_engine_gpu: a batch mean near zero is judged on the feature's spread
                if k.endswith('running_mean'):
                    sc = max(np.abs(tr.z[k]).max(), float(np.sqrt(tr.z[k.replace('running_mean', 'running_var')]).max()))
                close(sd[k[len('gen_final/'):]], tr.z[k], scale=sc, what=k)


def replay_dpsgd(dist, dev, clip_quantile):
    """DP-SGD sharding proof: every rank runs its row shard of the reference trace's first discriminator update in
    dp_mode='dpsgd' (per-rank per-example norms and clip factors, clipped sums all-reduced, ONE noise tensor added by
    every rank after the all-reduce) and compares with oracle/dpsgd.py run single-process on the GLOBAL batch:
    norms and clip factors of its own examples, the clipped mean gradient, the noised + Adam-updated parameters."""
    from golden_util import Trace
    from dp_cgans_amd.engine import StepEngine
    from oracle import dpsgd
    from oracle import nets as onets
    from test_engine_gpu import _flat_noise, close, injected
    tr = Trace('trace_np.npz')
    R, r = dist.world, dist.rank
    s = tr.kinds.index('D')
    dp = {k: torch.from_numpy(v) for k, v in tr.params(s, 'p_before').items()}
    gp = {k: torch.from_numpy(v) for k, v in tr.params(tr.kinds.index('G'), 'p_before').items()}
    full = injected(tr, s, dev)
    cpu = {k: (v.cpu() if torch.is_tensor(v) else [m.cpu() for m in v]) for k, v in full.items()}
    n = tr.B // tr.pac
    mk = lambda g: [m[g * n:(g + 1) * n] for m in cpu['keep']]
    args = (gp, dp, tr.span_info, tr.pac, cpu['z'], cpu['cond'], cpu['perm'], cpu['real'], cpu['exp'], mk(0), mk(1), mk(2),
            cpu['alpha'])
    norms = dpsgd.d_step_grads_dpsgd(*args, clip=1e30)['norms'].numpy()
    clip = {'none': float(norms.max()) * 2, 'median': float(np.median(norms)), 'all': float(norms.min()) * 0.5}[clip_quantile]
    sigma = 0.8
    out = dpsgd.d_step_grads_dpsgd(*args, clip=clip)          # the global batch, one process
    Bl, nl = tr.B // R, n // R
    eng = StepEngine(tr.span_info, tr.E, tr.gen_dim, tr.disc_dim, Bl, tr.pac, dev, private=True, dp_mode='dpsgd',
                     dp_sigma=sigma, dp_clip=clip, dist=dist, global_batch=tr.B)
    eng.disc.load(dp)
    eng.gen.load(gp)
    per, flat = _flat_noise(eng)                              # same seed on every rank: the shared noise tensor
    inj = {k: shard(full[k], r, R) for k in ('z', 'cond', 'mask', 'exp')}
    inj['keep'] = [torch.cat([shard(m[b * n:(b + 1) * n], r, R) for b in range(3)]) for m in full['keep']]
    inj['real'] = shard(full['real'], r, R)
    inj['real_cond'] = shard(full['cond'][full['perm']], r, R)
    inj['alpha'] = shard(full['alpha'], r, R)
    inj['dp_noise'] = flat
    eng.d_step(inj)
    mine = slice(r * nl, (r + 1) * nl)
    close(eng.norms, out['norms'][mine], what='rank %d: per-example gradient norms' % r)
    close(eng.rowc[0:nl], out['factors'][mine], what='rank %d: clip factors' % r)
    for k in tr.dkeys:          # the all-reduced clipped sums, scaled by the GLOBAL example count
        close(eng.disc.G(k), out['grads'][k], rtol=2e-5, what='rank %d: clipped mean grad %s' % (r, k))
    noisy = dpsgd.add_noise(out['grads'], per, sigma, clip, n)
    dp2 = onets.apply_update(dp, noisy, {'t': 0, 'm': {}, 'v': {}})
    for k in tr.dkeys:
        close(eng.disc.P(k), dp2[k], what='rank %d: noised + Adam param %s' % (r, k))
    chk = eng.disc.p.clone()
    dist.allreduce(chk)
    assert torch.equal(chk, eng.disc.p * R), 'replicas diverged: the noise was not added identically on every rank'


def local_bn(dist, dev):
    """bn_sync=False - every rank's BatchNorm over its OWN rows, which is what DistributedDataParallel computes: the
    first discriminator update and the first generator update of the reference trace, sharded by rows.  Expected
    gradients: the mean over ranks of the single-process oracle run on each rank's shard as a batch of its own (the
    loss means are global means = means of the shard means); running statistics: the oracle's on this rank's shard."""
    from golden_util import Trace
    from dp_cgans_amd.engine import StepEngine
    from oracle import nets as onets
    from test_engine_gpu import close, injected
    tr = Trace('trace_np.npz')
    R, r = dist.world, dist.rank
    sD, sG = tr.kinds.index('D'), tr.kinds.index('G')
    dp = {k: torch.from_numpy(v) for k, v in tr.params(sD, 'p_before').items()}
    gp = {k: torch.from_numpy(v) for k, v in tr.params(sG, 'p_before').items()}
    n = tr.B // tr.pac
    eng = StepEngine(tr.span_info, tr.E, tr.gen_dim, tr.disc_dim, tr.B // R, tr.pac, dev, private=False, dist=dist,
                     global_batch=tr.B, bn_sync=False)
    assert not eng.bn_dist and eng.bn_count == tr.B // R
    eng.disc.load(dp)
    eng.gen.load(gp)
    L = onets.n_residual(gp)
    bn_state = {}
    for l in range(L):
        F_ = gp['seq.%d.bn.weight' % l].shape[0]
        bn_state['seq.%d.bn.running_mean' % l] = torch.zeros(F_)
        bn_state['seq.%d.bn.running_var' % l] = torch.ones(F_)
        bn_state['seq.%d.bn.num_batches_tracked' % l] = torch.zeros((), dtype=torch.int64)

    # --- discriminator update
    full = injected(tr, sD, dev)
    cpu = {k: (v.cpu() if torch.is_tensor(v) else [m.cpu() for m in v]) for k, v in full.items()}
    real_cond = cpu['cond'][cpu['perm']]

    def d_grads(q, state):
        sh = lambda x: shard(x, q, R)
        mk = lambda b: [sh(m[b * n:(b + 1) * n]) for m in cpu['keep']]
        dpl = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
        fake = onets.gen_forward(gp, torch.cat([sh(cpu['z']), sh(cpu['cond'])], 1), state)
        act = onets.apply_activate(fake, tr.span_info, sh(cpu['exp']))
        fake_cat, real_cat = torch.cat([act, sh(cpu['cond'])], 1), torch.cat([sh(cpu['real']), sh(real_cond)], 1)
        fake_cat = fake_cat.detach().requires_grad_(True)      # (the penalty differentiates with respect to its input rows)
        loss = -(onets.disc_forward(dpl, real_cat, tr.pac, mk(1)).mean() - onets.disc_forward(dpl, fake_cat, tr.pac, mk(0)).mean())
        pen = onets.gradient_penalty(dpl, real_cat, fake_cat, tr.pac, sh(cpu['alpha']), mk(2))
        keys = list(dpl)
        gs = torch.autograd.grad(loss + pen, [dpl[k] for k in keys], allow_unused=True)
        return {k: (torch.zeros_like(dpl[k]) if g is None else g.detach()) for k, g in zip(keys, gs)}
    per_rank = [d_grads(q, bn_state if q == r else None) for q in range(R)]
    want = {k: sum(g[k] for g in per_rank) / R for k in per_rank[0]}
    inj = {k: shard(full[k], r, R) for k in ('z', 'cond', 'mask', 'exp')}
    inj['keep'] = [torch.cat([shard(m[b * n:(b + 1) * n], r, R) for b in range(3)]) for m in full['keep']]
    inj['real'] = shard(full['real'], r, R)
    inj['real_cond'] = shard(full['cond'][full['perm']], r, R)
    inj['alpha'] = shard(full['alpha'], r, R)
    eng.d_step(inj)
    for k in tr.dkeys:
        close(eng.disc.G(k), want[k], rtol=2e-5, what='rank %d: D grad %s' % (r, k))

    # --- generator update (the engine's discriminator back on the trace's weights for that step)
    dnow = {k: eng.disc.P(k).detach().cpu().clone() for k in tr.dkeys}
    full = injected(tr, sG, dev)
    cpu = {k: (v.cpu() if torch.is_tensor(v) else [m.cpu() for m in v]) for k, v in full.items()}
    per_rank = []
    for q in range(R):
        sh = lambda x: shard(x, q, R)
        out = onets.g_step_grads(gp, dnow, tr.span_info, tr.pac, sh(cpu['z']), sh(cpu['cond']), sh(cpu['mask']),
                                 sh(cpu['exp']), [sh(m) for m in cpu['keep']], bn_state=bn_state if q == r else None,
                                 loop_loss=False)
        per_rank.append(out['grads'])
    want = {k: sum(g[k] for g in per_rank) / R for k in per_rank[0]}
    inj = {k: shard(full[k], r, R) for k in ('z', 'cond', 'mask', 'exp')}
    inj['keep'] = [shard(m, r, R) for m in full['keep']]
    eng.g_step(inj)
    for k in tr.gkeys:
        sc = want[k.replace('fc.bias', 'fc.weight')].abs().max().item() if 'fc.bias' in k else None
        close(eng.gen.G(k), want[k], rtol=2e-5, scale=sc, what='rank %d: G grad %s' % (r, k))
    sd = eng.generator_state_dict()
    for k, v in bn_state.items():
        if 'running' in k:
            sc = None
            if k.endswith('running_mean'):
                sc = max(float(v.abs().max()), float(bn_state[k.replace('running_mean', 'running_var')].sqrt().max()))
            close(sd[k], v, scale=sc, what='rank %d: %s (this rank\'s rows only)' % (r, k))
    # the parameters stay replicated (the running statistics are per rank, as under DistributedDataParallel)
    for net in (eng.disc, eng.gen):
        chk = net.p.clone()
        dist.allreduce(chk)
        assert torch.equal(chk, net.p * R), 'replicas diverged'


def graph(dist, dev, mode, bn_sync=True):
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']

    def make():
        eng = StepEngine(span, 24, (40, 24), (48, 16), 60, 10, dev, private=True, dp_mode=mode, seed=3, dist=dist,
                         bn_sync=bn_sync)
        eng.sampler = DataSampler(enc, span, device=dev, seed=9 + dist.rank)
        return eng
    from dp_cgans_amd import ops as _ops
    _ops.set_deterministic(True)      # bit-equality below needs the atomics-free GEMM path
    a, b, c = make(), make(), make()
    c.pair_forwards = False           # two separate generator forwards (one all-reduce per layer each)
    for _ in range(4):
        a.train_step(1)
        c.train_step(1)
    gs = GraphStep(b, d_steps=1)
    for _ in range(4):
        gs.run()
    torch.cuda.synchronize()
    if os.environ.get('DPCGAN_GRAPH_COLLECTIVES') != '0' and dist.backend == 'nccl':
        assert len(gs.items) == 1 and gs.collectives_in_graph, 'RCCL collectives are captured: one graph per step'
    else:
        assert len(gs.items) > 1, 'a data-parallel step must be cut into graph segments at its collectives'
    if not bn_sync:
        n_ar = sum(1 for k, _ in gs.items if k != 'graph')
        assert gs.collectives_in_graph or n_ar == 2, 'per-rank statistics: the two gradient all-reduces are the only exchanges (%d)' % n_ar
    assert torch.equal(a.disc.p, b.disc.p) and torch.equal(a.gen.p, b.gen.p)
    assert torch.equal(a.disc.p, c.disc.p) and torch.equal(a.gen.p, c.gen.p), 'paired forwards changed the step'
    sa, sc = a.generator_state_dict(), c.generator_state_dict()
    for k in sa:
        assert torch.equal(sa[k].cpu(), sc[k].cpu()), k
    chk = a.disc.p.clone()
    dist.allreduce(chk)
    assert torch.equal(chk, a.disc.p * dist.world), 'replicas diverged'
    assert torch.isfinite(a.gen.p).all()


def fallback(dist, dev, mode):
    """GraphStep's capture-failure fallback: the first collective of the in-graph capture raises (test hook), the step
    must warn, end that capture cleanly (branches were open) and capture again with eager collectives between graph
    segments - bit-equal to the eager step."""
    import warnings
    from golden_util import load
    from dp_cgans_amd import ops as _ops
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    assert dist.backend == 'nccl', 'the in-graph capture (and so its fallback) exists for RCCL only'
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    _ops.set_deterministic(True)

    def make():
        eng = StepEngine(span, 24, (40, 24), (48, 16), 60, 10, dev, private=True, dp_mode=mode, seed=3, dist=dist)
        eng.sampler = DataSampler(enc, span, device=dev, seed=9 + dist.rank)
        return eng
    a, b = make(), make()
    for _ in range(4):
        a.train_step(1)
    gs = GraphStep(b, d_steps=1)
    GraphStep.INJECT_CAPTURE_FAULT = True
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            gs.run()          # eager
            gs.run()          # capture: in-graph attempt fails, segments follow
        assert any('falling back to eager collectives' in str(x.message) for x in w), [str(x.message) for x in w]
    finally:
        GraphStep.INJECT_CAPTURE_FAULT = False
    assert not gs.collectives_in_graph and len(gs.items) > 1
    gs.run()
    gs.run()
    torch.cuda.synchronize()
    assert torch.equal(a.disc.p, b.disc.p) and torch.equal(a.gen.p, b.gen.p), 'the segment path after a failed capture differs from eager'
    sa, sb = a.generator_state_dict(), b.generator_state_dict()
    for k in sa:
        assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
    # and a fresh in-graph capture still works afterwards on the same streams
    c = make()
    gc = GraphStep(c, d_steps=1)
    for _ in range(4):
        gc.run()
    torch.cuda.synchronize()
    assert gc.collectives_in_graph and len(gc.items) == 1
    assert torch.equal(a.disc.p, c.disc.p)


def fused(dist, dev, mode):
    """The default (non-reproducible) data-parallel step: BatchNorm sums out of the GEMM epilogues in slots, folded on
    the device and all-reduced as one vector per layer, the two generator forwards paired - against the same step
    with separate forwards and (one rank) against the engine without any collective.  One training step from
    identical parameters and draws; the gradient buffers it leaves behind within the step's 2e-5 bar."""
    import copy
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']

    def make(d):
        eng = StepEngine(span, 24, (40, 24), (48, 16), 60, 10, dev, private=True, dp_mode=mode, seed=3, dist=d)
        eng.sampler = DataSampler(enc, span, device=dev, seed=9 + dist.rank)
        return eng
    a, b, g = make(dist), make(dist), make(dist)
    assert a.fused_stats and a.stat_slots > 1
    b.pair_forwards = False
    a.train_step(1)
    b.train_step(1)
    gs = GraphStep(g, d_steps=1)
    gs.run()
    engines = [('separate forwards', b), ('graph step (first run: eager)', g)]
    if dist.world == 1:
        off = copy.copy(dist)
        off.active = False
        q = make(off)
        q.train_step(1)
        engines.append(('no collectives', q))
    torch.cuda.synchronize()

    def close(x, y, what):
        err = float((x.double() - y.double()).abs().max()) / max(float(y.double().abs().max()), 1e-30)
        assert err <= 2e-5, '%s: %.3e' % (what, err)
    for name, e in engines:
        for k in a.disc.keys[:-1]:          # (the head bias gradient is an exact zero + rounding residue)
            close(a.disc.G(k), e.disc.G(k), '%s: D grad %s' % (name, k))
        for k in a.gen.keys:
            if 'fc.bias' not in k:
                close(a.gen.G(k), e.gen.G(k), '%s: G grad %s' % (name, k))
        sa, se = a.generator_state_dict(), e.generator_state_dict()
        for k in sa:
            if 'running' in k:
                close(sa[k], se[k], '%s: %s' % (name, k))


def main():
    mode = sys.argv[1]
    from dp_cgans_amd.parallel import Dist
    dist = Dist(backend=os.environ.get('DP_WORKER_BACKEND', 'gloo'))
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    if mode == 'replay':
        replay(dist, dev)
    elif mode == 'replay_dpsgd':
        replay_dpsgd(dist, dev, sys.argv[2])
    elif mode == 'local_bn':
        local_bn(dist, dev)
    elif mode == 'graph_local_bn':
        graph(dist, dev, sys.argv[2], bn_sync=False)
    elif mode == 'fused':
        fused(dist, dev, sys.argv[2])
    elif mode == 'fallback':
        fallback(dist, dev, sys.argv[2])
    else:
        graph(dist, dev, sys.argv[2])
    dist.barrier()
    dist.shutdown()
    print('rank %d ok' % dist.rank)


if __name__ == '__main__':
    main()
