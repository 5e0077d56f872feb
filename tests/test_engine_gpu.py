"""Whole-step parity: the HIP step engine replays the reference's recorded training steps
(tests/golden/trace_*.npz: every random draw injected) and must reproduce losses, gradient
penalty, generator outputs, gradients and updated weights within 1e-5 relative (north_star)."""
import numpy as np
import pytest
import torch

from golden_util import Trace

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(autouse=True, params=['penalty-direct', 'penalty-gram', 'penalty-gram,dense-cond'])
def step_form(request):
    """Every test of this file runs with both forms of the discriminator update's first layer - the
    direct one (interpolates, g0 = u W1 and v0 formed) and the Gram one (W1 W1^T; engine._d_back) - and
    with the generator's cond columns as row gathers / scatters (the default) or inside the dense
    products."""
    from dp_cgans_amd.engine import StepEngine
    StepEngine.GP_GRAM = 'penalty-gram' in request.param
    StepEngine.SPARSE_COND = False if 'dense-cond' in request.param else None
    yield request.param
    StepEngine.GP_GRAM = None
    StepEngine.SPARSE_COND = None


def close(a, b, rtol=RTOL, what='', scale=None):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    a, b = a.reshape(-1), b.reshape(-1)
    assert a.shape == b.shape, '%s: %s vs %s' % (what, a.shape, b.shape)
    sc = max(np.abs(b).max(), 1e-30) if scale is None else scale
    err = np.abs(a - b).max() / sc
    assert err <= rtol, '%s: max err / scale = %.3e' % (what, err)


from golden_util import AdamAwareCompare  # noqa: E402,F401  (shared with the CPU oracle tests and dp_worker.py)


def make_engine(tr, device='cuda:0'):
    from dp_cgans_amd.engine import StepEngine
    eng = StepEngine(tr.span_info, tr.E, tr.gen_dim, tr.disc_dim, tr.B, tr.pac, device, private=bool(tr.private))
    eng.want_interp_scores = True      # the trace holds the discriminator's scores of the interpolates too (a diagnostic in the step)
    first_d, first_g = tr.kinds.index('D'), tr.kinds.index('G')
    eng.disc.load(tr.params(first_d, 'p_before'))
    eng.gen.load(tr.params(first_g, 'p_before'))
    return eng


def injected(tr, s, dev):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    inj = dict(z=t(tr.get(s, 'normal')), exp=t(tr.exp_full(s)))
    if tr.has_cond:
        inj.update(cond=t(tr.get(s, 'cond')), mask=t(tr.get(s, 'mask')))
    groups = tr.masks(s)
    nh = len(tr.disc_dim)
    inj['keep'] = [t(np.concatenate([grp[i] for grp in groups], axis=0)) for i in range(nh)]
    if tr.kinds[s] == 'D':
        if tr.has_cond:
            inj['perm'] = t(tr.get(s, 'shuffle').astype(np.int64))
        inj['real'] = t(tr.get(s, 'real').astype(np.float32))
        inj['alpha'] = t(tr.get(s, 'rand').reshape(-1).astype(np.float32))
        if tr.private:
            tot = np.zeros(len(tr.dkeys), dtype=np.float64)
            for per_bwd in tr.hook_noise(s):
                for pid, v in per_bwd.items():
                    tot[pid] += v
            inj['hook_noise'] = t((tot / tr.B).astype(np.float32))
    return inj


# largest share of one tensor's entries AdamAwareCompare may hold to |diff| <= lr in the trace replays (what the runs
# need + a margin; the shares themselves are in the test report: golden_util.AdamAwareCompare.report)
EXCUSED_CEILING = {'trace_priv.npz': 0.07, 'trace_cfg2.npz': 0.03}       # seen: 0.056 (output layer, 56 x 678) and 0.0195 (5 of 256 biases)


@pytest.mark.parametrize('name', ['trace_np.npz', 'trace_priv.npz', 'trace_num.npz', 'trace_cfg1.npz', 'trace_cfg2.npz'])
def test_engine_replays_reference_training_steps(name):
    tr = Trace(name)
    eng = make_engine(tr)
    dev = eng.dev
    B, Dd, n = tr.B, eng.Dd, eng.n
    # a trace that holds ONE update of a network (BASELINE configs 1 and 2's first iteration) checks that update with
    # the first-step sensitivity bound: at t = 1 Adam's update is lr*g/(|g|+eps) and among the 10^5 entries of a
    # full-size layer a few have |g| within rounding of eps (1.1e-5 seen against the 1e-5 bar)
    cmp_d = AdamAwareCompare(first_step_sensitivity=tr.kinds.count('D') == 1, ceiling=EXCUSED_CEILING.get(name, 0.02))
    cmp_g = AdamAwareCompare(first_step_sensitivity=tr.kinds.count('G') == 1, ceiling=EXCUSED_CEILING.get(name, 0.02))
    for s, kind in enumerate(tr.kinds):
        inj = injected(tr, s, dev)
        if kind == 'D':
            eng.d_step(inj)
            close(eng.fake, tr.get(s, 'fake'), what='s%d fake' % s)
            if tr.has_cond:
                close(eng.fake_act, tr.get(s, "fakeact"), what="s%d fakeact" % s)
            else:       # Q5: the activation is computed and dropped, the discriminator sees the raw output
                close(eng.dfake, tr.get(s, "fakeact"), what="s%d fakeact (unused)" % s)
                close(eng.fake_act, tr.get(s, "fake"), what="s%d discriminator input" % s)
            close(eng.y[0:n], tr.get(s, 'd_out', 0), what='s%d y_fake' % s)
            close(eng.y[n:2 * n], tr.get(s, 'd_out', 1), what='s%d y_real' % s)
            close(eng.y[2 * n:], tr.get(s, 'd_out', 2), what='s%d y_interp' % s)
            losses = eng.losses.cpu().numpy()
            close(losses[1:2], [float(tr.get(s, 'pen'))], what='s%d pen' % s)
            # loss_d is a difference of two means: judge it on the scale of those means
            ysc = max(abs(float(tr.get(s, 'd_out', 0).mean())), abs(float(tr.get(s, 'd_out', 1).mean())))
            close(losses[0:1], [float(tr.get(s, 'backward_value', 1))], scale=ysc, what='s%d loss_d' % s)
            ref_g = tr.params(s, 'g')
            for i, k in enumerate(tr.dkeys):
                mine = eng.disc.G(k)
                if tr.private:
                    mine = mine + inj['hook_noise'][i]
                close(mine, ref_g[k], rtol=2e-5, what='s%d D grad %s' % (s, k))
            ref_p = tr.params(s, 'p_after')
            cmp_d.steps += 1
            for k in tr.dkeys:
                ref = np.clip(ref_p[k], -0.01, 0.01) if tr.private else ref_p[k]   # recorded before the clamp
                cmp_d.check(k, eng.disc.P(k), ref, ref_g[k], 's%d D param %s' % (s, k))
        else:
            eng.g_step(inj)
            close(eng.fake, tr.get(s, 'fake'), what='s%d fake' % s)
            close(eng.fake_act, tr.get(s, "fakeact"), what="s%d fakeact" % s)
            close(eng.y_g, tr.get(s, 'd_out', 0), what='s%d y_fake' % s)
            losses = eng.losses.cpu().numpy()
            if tr.has_cond:
                close(losses[3:4], [float(tr.get(s, 'ce'))], what='s%d ce' % s)
            close([losses[2] + losses[3]], [float(tr.get(s, 'backward_value', 0))], what='s%d loss_g' % s)
            ref_g = tr.params(s, 'g')
            for k in tr.gkeys:
                # Linear bias in front of BatchNorm: exactly-zero gradient, rounding noise on both sides
                sc = np.abs(ref_g[k.replace('fc.bias', 'fc.weight')]).max() if 'fc.bias' in k else None
                close(eng.gen.G(k), ref_g[k], rtol=2e-5, scale=sc, what='s%d G grad %s' % (s, k))
            ref_p = tr.params(s, 'p_after')
            cmp_g.steps += 1
            for k in tr.gkeys:
                if 'fc.bias' in k:      # Adam amplifies that noise to up to lr per step (see test_oracle_golden)
                    assert np.abs(eng.gen.P(k).cpu().numpy() - ref_p[k]).max() < 2e-4 * cmp_g.steps
                else:
                    cmp_g.check(k, eng.gen.P(k), ref_p[k], ref_g[k], 's%d G param %s' % (s, k))
    sd = eng.generator_state_dict()
    for k in tr.z.files:
        if k.startswith('gen_final/') and 'running' in k:
            # a batch mean is known to rounding of the VALUES it averages: judge running_mean on the
            # scale of the feature's spread when the mean itself is near zero
            sc = None
            if k.endswith('running_mean'):
                sc = max(np.abs(tr.z[k]).max(), float(np.sqrt(tr.z[k.replace('running_mean', 'running_var')]).max()))
            close(sd[k[len('gen_final/'):]], tr.z[k], scale=sc, what=k)
    assert int(sd['seq.0.bn.num_batches_tracked']) == int(tr.z['gen_final/seq.0.bn.num_batches_tracked'])


def test_engine_sampler_path_matches_oracle_on_same_draws():
    """Philox-driven step: read back the draws the kernels used (dpc_fill exposes the same map),
    feed them to the CPU oracle, compare losses / gradients / weights."""
    from golden_util import load
    from dp_cgans_amd import ops
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import StepEngine
    z = load('adult_sampler.npz')
    _check_step_against_oracle(z['encoded'], z['span_info'], 24, (40, 24), (48, 16), 120, 10)


def _synthetic_table(rng, n_rows, cont_modes, disc_dims):
    """Encoded matrix + span_info: continuous columns (value, one-hot of `m` modes), then discrete ones."""
    cols, span = [], []
    for m in cont_modes:
        cols.append(np.clip(rng.normal(0, 0.4, (n_rows, 1)), -0.99, 0.99))
        cols.append(np.eye(m)[rng.integers(m, size=n_rows)])
        span += [(1, 0, 2), (m, 1, 2)]
    for k in disc_dims:
        p = rng.dirichlet(np.ones(k) * 2.0)
        cols.append(np.eye(k)[rng.choice(k, size=n_rows, p=p)])
        span.append((k, 1, 1))
    return np.concatenate(cols, axis=1), np.array(span, dtype=np.int32)


@pytest.mark.parametrize('cfg', [
    # E, generator dims, discriminator dims, batch, pac, continuous modes, discrete dims
    dict(E=7, g=(37,), d=(21,), B=36, pac=3, cont=(1, 3), disc=(2, 5)),                   # one hidden layer each, odd sizes
    dict(E=16, g=(24, 8, 40), d=(32, 12, 20), B=64, pac=1, cont=(2,), disc=(3, 17, 2)),    # three layers, pac 1, a 17-way span
    dict(E=5, g=(9, 11), d=(13, 6), B=20, pac=10, cont=(), disc=(2, 2, 4)),                # discrete columns only
    dict(E=32, g=(64, 64), d=(64, 64), B=512, pac=8, cont=(1, 2, 3, 1), disc=(2, 3, 4, 6, 8)),
    dict(E=12, g=(20, 16), d=(24, 8), B=40, pac=4, cont=(2, 1, 3), disc=()),               # no discrete columns (Q5)
])
def test_step_against_oracle_on_odd_shapes(cfg):
    """The Philox-driven step against the CPU oracle on table / network shapes that exercise the
    unaligned and edge paths (row lengths not multiples of 4, single layers, pac 1, wide spans)."""
    rng = np.random.default_rng(len(cfg['disc']) * 7 + cfg['B'])
    enc, span = _synthetic_table(rng, 400, cfg['cont'], cfg['disc'])
    _check_step_against_oracle(enc, span, cfg['E'], cfg['g'], cfg['d'], cfg['B'], cfg['pac'])


def _check_step_against_oracle(enc, span, E, gdim, ddim, B, pac, loop_loss=True, adam_sensitivity=False, ceiling=0.03):
    from dp_cgans_amd import ops
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import StepEngine
    from oracle import nets as onets
    dev = torch.device('cuda:0')
    eng = StepEngine(span, E, gdim, ddim, B, pac, dev, private=False, seed=5)
    eng.sampler = DataSampler(enc, span, device=dev)
    gp = {k: v.cpu() for k, v in eng.gen.state_dict().items()}
    dp = {k: v.cpu() for k, v in eng.disc.state_dict().items()}
    n, Dd = eng.n, eng.Dd

    def draws(rows):
        e = torch.empty(B, Dd, device=dev)
        ops.fill(e, ops.FILL_EXP, eng.rng, ops.SITE_GUMBEL)
        zz = torch.empty(B, E, device=dev)
        ops.fill(zz, ops.FILL_NORMAL, eng.rng, ops.SITE_Z)
        keeps = []
        for i, w in enumerate(eng.ddim):
            kb = torch.empty(rows, w, dtype=torch.uint8, device=dev)
            ops.fill(kb, ops.FILL_KEEP, eng.rng, ops.SITE_DROP0 + i)
            keeps.append(kb.cpu())
        al = torch.empty(n, device=dev)
        ops.fill(al, ops.FILL_UNIFORM, eng.rng, ops.SITE_ALPHA)
        return e.cpu(), zz.cpu(), keeps, al.cpu()

    # ---- discriminator update
    e, zz, keeps, al = draws(3 * n)
    eng.d_step()
    real = eng.Xd[B:2 * B, :Dd].cpu()
    rid = eng.row_id.cpu().numpy()
    assert np.array_equal(real.numpy(), enc[rid].astype(np.float32))
    if eng.has_cond:
        cond = eng.cond.cpu()
        perm = eng.perm.cpu()
        sel = eng.sel.cpu().numpy()[perm.numpy()]
        cat_col = eng.sampler.cat_col.cpu().numpy()
        assert (enc[rid, cat_col[sel[:, 0]]] == 1).all() and (enc[rid, cat_col[sel[:, 1]]] == 1).all()
    else:               # no discrete columns: no cond vector, uniformly drawn rows
        cond, perm = torch.zeros(B, 0), torch.arange(B)
    out = onets.d_step_grads(gp, dp, span, pac, zz, cond, perm, real, e, [k[0:n] for k in keeps],
                             [k[n:2 * n] for k in keeps], [k[2 * n:] for k in keeps], al)
    losses = eng.losses.cpu().numpy()
    close(losses[0:1], [float(out['loss_d'])], scale=float(out['y_fake'].abs().mean()), what='loss_d')
    close(losses[1:2], [float(out['pen'])], what='pen')
    for k in eng.disc.keys:
        close(eng.disc.G(k), out['grads'][k], rtol=2e-5, what='D grad ' + k)
    mine_g = {k: eng.disc.G(k).clone() for k in eng.disc.keys}
    dp2 = onets.apply_update(dp, out['grads'], {'t': 0, 'm': {}, 'v': {}})
    cmp_d = AdamAwareCompare(first_step_sensitivity=adam_sensitivity, ceiling=ceiling)      # (odd shapes: 0.023 seen)
    cmp_d.steps = 1
    for k in eng.disc.keys:
        cmp_d.check(k, eng.disc.P(k), dp2[k].numpy(), out['grads'][k].numpy(), 'D param ' + k, mine_g=mine_g[k])
    # ---- generator update, both sides with the SAME discriminator: the two updated copies agree to the bars above,
    # but where those allow +-lr (noise-sized gradient entries) a wide first layer differs by up to 10 % of single
    # weights, and the generator's gradient would inherit that instead of showing its own error
    eng.disc.load(dp2)
    e, zz, keeps, al = draws(n)
    eng.g_step()
    cond, mask = (eng.cond.cpu(), eng.mask.cpu()) if eng.has_cond else (torch.zeros(B, 0), torch.zeros(B, 0, dtype=torch.int32))
    out = onets.g_step_grads(gp, dp2, span, pac, zz, cond, mask, e, keeps, loop_loss=loop_loss)
    losses = eng.losses.cpu().numpy()
    close([losses[2] + losses[3]], [float(out['loss_g'])], what='loss_g')
    for k in eng.gen.keys:
        sc = float(out['grads'][k.replace('fc.bias', 'fc.weight')].abs().max()) if 'fc.bias' in k else None
        close(eng.gen.G(k), out['grads'][k], rtol=2e-5, scale=sc, what='G grad ' + k)


def _flat_noise(eng, seed=11):
    """N(0,1) noise per discriminator parameter: ({key: cpu tensor}, flat device tensor in the engine's layout)."""
    g = torch.Generator().manual_seed(seed)
    per = {k: torch.randn(eng.disc.shapes[k], generator=g) for k in eng.disc.keys}
    flat = torch.zeros(eng.disc.n)
    for k in eng.disc.keys:
        o = eng.disc.offsets[k]
        flat[o:o + per[k].numel()] = per[k].reshape(-1)
    return per, flat.to(eng.dev)


@pytest.mark.parametrize('clip_quantile', [None, 0.5, 0.0])
def test_dpsgd_discriminator_update_matches_per_example_oracle(clip_quantile):
    """DP-SGD mode: per-example (pac group) norms from the Gram kernel, clip factors, fused
    clip-scale-sum gradient, injected Gaussian noise and Adam, against oracle/dpsgd.py's per-example
    autograd loop on the inputs of the reference trace.  Clip bounds: above every norm (None),
    at the median norm, below every norm."""
    from dp_cgans_amd.engine import StepEngine
    from oracle import dpsgd
    from oracle import nets as onets
    tr = Trace('trace_np.npz')
    s = tr.kinds.index('D')
    dev = torch.device('cuda:0')
    dp = {k: torch.from_numpy(v) for k, v in tr.params(s, 'p_before').items()}
    gp = {k: torch.from_numpy(v) for k, v in tr.params(tr.kinds.index('G'), 'p_before').items()}
    inj = injected(tr, s, dev)
    cpu = {k: (v.cpu() if torch.is_tensor(v) else [m.cpu() for m in v]) for k, v in inj.items()}
    n = tr.B // tr.pac
    mk = lambda g: [m[g * n:(g + 1) * n] for m in cpu['keep']]
    base = dpsgd.d_step_grads_dpsgd(gp, dp, tr.span_info, tr.pac, cpu['z'], cpu['cond'], cpu['perm'], cpu['real'],
                                    cpu['exp'], mk(0), mk(1), mk(2), cpu['alpha'], clip=1e30)
    norms = base['norms'].numpy()
    clip = {None: float(norms.max()) * 2, 0.5: float(np.median(norms)), 0.0: float(norms.min()) * 0.5}[clip_quantile]
    sigma = 0.8
    out = dpsgd.d_step_grads_dpsgd(gp, dp, tr.span_info, tr.pac, cpu['z'], cpu['cond'], cpu['perm'], cpu['real'],
                                   cpu['exp'], mk(0), mk(1), mk(2), cpu['alpha'], clip=clip)
    eng = StepEngine(tr.span_info, tr.E, tr.gen_dim, tr.disc_dim, tr.B, tr.pac, dev, private=True, dp_mode='dpsgd',
                     dp_sigma=sigma, dp_clip=clip)
    eng.disc.load(dp)
    eng.gen.load(gp)
    per, flat = _flat_noise(eng)
    inj['dp_noise'] = flat
    eng.d_step(inj)
    close(eng.norms, out['norms'], what='per-example gradient norms')
    close(eng.rowc[0:n], out['factors'], what='clip factors')
    assert torch.equal(eng.rowc[0:n], eng.rowc[n:2 * n]) and torch.equal(eng.rowc[0:n], eng.rowc[2 * n:])
    for k in tr.dkeys:
        close(eng.disc.G(k), out['grads'][k], rtol=2e-5, what='clipped mean grad ' + k)
    noisy = dpsgd.add_noise(out['grads'], per, sigma, clip, n)
    dp2 = onets.apply_update(dp, noisy, {'t': 0, 'm': {}, 'v': {}})
    for k in tr.dkeys:
        close(eng.disc.P(k), dp2[k], what='noised + Adam param ' + k)      # no weight clamp in this mode


def test_reference_mode_hook_noise_on_device_has_the_reference_variance():
    """dp_mode='reference' without injection: seg_add[s] = z_s*sqrt(fires_s*(t+1))/B read back through dpc_fill's map."""
    from dp_cgans_amd import ops
    from dp_cgans_amd.engine import StepEngine
    tr = Trace('trace_priv.npz')
    dev = torch.device('cuda:0')
    eng = StepEngine(tr.span_info, tr.E, tr.gen_dim, tr.disc_dim, tr.B, tr.pac, dev, private=True)
    nseg = len(eng.disc.keys)
    for t in range(3):
        zs = torch.empty(nseg, device=dev)
        ops.fill(zs, ops.FILL_NORMAL, eng.rng_shared, ops.SITE_HOOK)
        eng.disc.t.fill_(t)
        ops.hook_noise(eng.hook_add, eng.hook_fires, eng.disc.t, 1.0 / eng.gB, eng.rng_shared)
        want = zs.cpu().numpy() * np.sqrt(eng.hook_fires.cpu().numpy() * (t + 1)) / eng.gB
        close(eng.hook_add, want, what='hook noise at t=%d' % t)
        ops.state_advance(eng.rng_shared)


@pytest.mark.parametrize('mode', ['reference', 'dpsgd'])
def test_graph_replay_equals_eager_steps(mode):
    """The captured step (HIP graph replay) must produce bit-identical parameters to the same
    number of eager steps from the same seeds."""
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')

    def make():
        eng = StepEngine(span, 24, (40, 24), (48, 16), 120, 10, dev, private=True, dp_mode=mode, seed=3)
        eng.sampler = DataSampler(enc, span, device=dev, seed=9)
        return eng
    from dp_cgans_amd import ops as _ops
    _ops.set_deterministic(True)      # bit-equality below needs the atomics-free GEMM path
    a, b = make(), make()
    steps = 5
    for _ in range(steps):
        a.train_step(2)
    gs = GraphStep(b, d_steps=2)
    for _ in range(steps):
        gs.run()
    torch.cuda.synchronize()
    assert len(gs.items) == 1
    assert torch.equal(a.losses, b.losses)
    assert a.bn_batches == b.bn_batches == steps * 3
    assert torch.isfinite(a.disc.p).all() and torch.isfinite(a.gen.p).all()
    _ops.set_deterministic(False)


@pytest.mark.parametrize('d_steps', [1, 2])
def test_overlapped_generator_forward_equals_the_sequential_step(d_steps):
    """train_step runs the generator update's forward beside the last discriminator update (own buffer
    set, running statistics applied afterwards in sequence order).  Bit-identical to the strictly
    sequential step: parameters, losses, and BatchNorm running statistics (whose two momentum updates
    per step do not commute)."""
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    from dp_cgans_amd import ops as _ops
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')

    def make():
        eng = StepEngine(span, 24, (40, 24), (48, 16), 120, 10, dev, private=True, dp_mode='dpsgd', seed=5)
        eng.sampler = DataSampler(enc, span, device=dev, seed=2)
        return eng
    _ops.set_deterministic(True)
    seq, ovl, gr = make(), make(), make()
    seq.concurrent = False
    for _ in range(3):
        for k in range(d_steps):
            seq.d_step(sub=k)
        seq.g_step(sub=d_steps)
        _ops.state_advance(seq.rng, rng_b=seq.rng_shared, t_a=seq.disc.t, add_a=d_steps, t_b=seq.gen.t, add_b=1)
        ovl.train_step(d_steps)
    gs = GraphStep(gr, d_steps=d_steps)
    for _ in range(3):
        gs.run()
    torch.cuda.synchronize()
    for other in (ovl, gr):
        assert torch.equal(seq.disc.p, other.disc.p) and torch.equal(seq.gen.p, other.gen.p)
        assert torch.equal(seq.losses, other.losses)
        sa, sb = seq.generator_state_dict(), other.generator_state_dict()
        for k in sa:
            assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
    _ops.set_deterministic(False)


@pytest.mark.parametrize('cfg', [
    dict(E=7, g=(37,), d=(21,), B=36, pac=3, cont=(1, 3), disc=(2, 5)),
    dict(E=16, g=(24, 8, 40), d=(32, 12, 20), B=64, pac=1, cont=(2,), disc=(3, 17, 2)),
    dict(E=32, g=(64, 64), d=(64, 64), B=512, pac=8, cont=(1, 2, 3, 1), disc=(2, 3, 4, 6, 8)),
])
def test_dpsgd_against_oracle_on_odd_shapes(cfg):
    """DP-SGD discriminator update, Philox-driven, against the per-example autograd loop on shapes with
    unaligned rows, one / three layers and pac 1: per-example norms, clip factors (bound at the median
    norm), clipped mean gradient."""
    rng = np.random.default_rng(len(cfg['disc']) * 7 + cfg['B'])
    enc, span = _synthetic_table(rng, 400, cfg['cont'], cfg['disc'])
    _check_dpsgd_against_oracle(enc, span, cfg['E'], cfg['g'], cfg['d'], cfg['B'], cfg['pac'])


def _check_dpsgd_against_oracle(enc, span, E, gdim, ddim, B, pac):
    from dp_cgans_amd import ops
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import StepEngine
    from oracle import dpsgd
    dev = torch.device('cuda:0')

    def run(clip):
        eng = StepEngine(span, E, gdim, ddim, B, pac, dev, private=True, dp_mode='dpsgd', dp_clip=clip, seed=5)
        eng.sampler = DataSampler(enc, span, device=dev)
        gp = {k: v.cpu() for k, v in eng.gen.state_dict().items()}
        dp = {k: v.cpu() for k, v in eng.disc.state_dict().items()}
        n, Dd = eng.n, eng.Dd
        e = torch.empty(B, Dd, device=dev)
        ops.fill(e, ops.FILL_EXP, eng.rng, ops.SITE_GUMBEL)
        zz = torch.empty(B, E, device=dev)
        ops.fill(zz, ops.FILL_NORMAL, eng.rng, ops.SITE_Z)
        keeps = []
        for i, w in enumerate(eng.ddim):
            kb = torch.empty(3 * n, w, dtype=torch.uint8, device=dev)
            ops.fill(kb, ops.FILL_KEEP, eng.rng, ops.SITE_DROP0 + i)
            keeps.append(kb.cpu())
        al = torch.empty(n, device=dev)
        ops.fill(al, ops.FILL_UNIFORM, eng.rng, ops.SITE_ALPHA)
        eng.d_step(update=False)
        mk = lambda g: [k[g * n:(g + 1) * n] for k in keeps]
        out = dpsgd.d_step_grads_dpsgd(gp, dp, span, pac, zz.cpu(), eng.cond.cpu(), eng.perm.cpu(),
                                       eng.Xd[B:2 * B, :Dd].cpu(), e.cpu(), mk(0), mk(1), mk(2), al.cpu(), clip=clip)
        return eng, out

    eng, out = run(1e30)
    close(eng.norms, out['norms'], what='per-example gradient norms')
    clip = float(np.median(out['norms'].numpy()))
    eng, out = run(clip)                          # same seeds, same draws
    n = eng.n
    close(eng.norms, out['norms'], what='per-example gradient norms')
    close(eng.rowc[0:n], out['factors'], what='clip factors')
    assert float(eng.rowc.min()) < 1.0 and float(eng.rowc.max()) == 1.0
    for k in eng.disc.keys:
        close(eng.disc.G(k), out['grads'][k], rtol=2e-5, what='clipped mean grad ' + k)


def test_dpsgd_graph_replayed_training_steps_match_the_oracle_chain():
    """Two full Philox-driven DP-SGD training steps (discriminator update with per-example clipping and Gaussian
    noise, then generator update) replayed through GraphStep, against the oracle chain oracle/dpsgd.py +
    oracle/nets.py run on the SAME draws: z, Gumbel noise, Dropout bits, alpha and the DP noise are read back
    through dpc_fill's index map (call sites offset by 64 per update, one Philox step per training step), the
    sampled cond vectors / permutation / real rows from the engine's buffers.  Parameters after every step."""
    from golden_util import load
    from dp_cgans_amd import ops
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    from oracle import dpsgd
    from oracle import nets as onets
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')
    E, B, pac, sigma, clip = 24, 120, 10, 0.7, 0.02
    eng = StepEngine(span, E, (40, 24), (48, 16), B, pac, dev, private=True, dp_mode='dpsgd', dp_sigma=sigma, dp_clip=clip,
                     seed=21)
    eng.sampler = DataSampler(enc, span, device=dev, seed=4)
    n, Dd = eng.n, eng.Dd
    gp = {k: v.cpu() for k, v in eng.gen.state_dict().items()}
    dp = {k: v.cpu() for k, v in eng.disc.state_dict().items()}
    sd, sg = {'t': 0, 'm': {}, 'v': {}}, {'t': 0, 'm': {}, 'v': {}}
    seed, seed_shared = int(eng.rng[0]), int(eng.rng_shared[0])
    gs = GraphStep(eng, d_steps=1)
    cmp_d, cmp_g = AdamAwareCompare(ceiling=0.02), AdamAwareCompare(ceiling=0.04)      # (0.028 seen on the 56 x 104 output layer)

    def draws(step, k, rows):
        rng = torch.tensor([seed, step], dtype=torch.int64, device=dev)
        so = 64 * k
        e = torch.empty(B, Dd, device=dev)
        ops.fill(e, ops.FILL_EXP, rng, ops.SITE_GUMBEL + so)
        zz = torch.empty(B, E, device=dev)
        ops.fill(zz, ops.FILL_NORMAL, rng, ops.SITE_Z + so)
        keeps = []
        for i, w in enumerate(eng.ddim):
            kb = torch.empty(rows, w, dtype=torch.uint8, device=dev)
            ops.fill(kb, ops.FILL_KEEP, rng, ops.SITE_DROP0 + i + so)
            keeps.append(kb.cpu())
        al = torch.empty(n, device=dev)
        ops.fill(al, ops.FILL_UNIFORM, rng, ops.SITE_ALPHA + so)
        return e.cpu(), zz.cpu(), keeps, al.cpu()

    for step in range(3):                # eager, capture, replay
        gs.run()
        torch.cuda.synchronize()
        # ---- discriminator update (update 0 of the step)
        e, zz, keeps, al = draws(step, 0, 3 * n)
        cond, perm, real = eng.cd.cond.cpu(), eng.perm.cpu(), eng.Xd[B:2 * B, :Dd].cpu()
        mk = lambda g_: [k[g_ * n:(g_ + 1) * n] for k in keeps]
        out = dpsgd.d_step_grads_dpsgd(gp, dp, span, pac, zz, cond, perm, real, e, mk(0), mk(1), mk(2), al, clip=clip)
        flat = torch.empty(eng.disc.n, device=dev)
        ops.fill(flat, ops.FILL_NORMAL, torch.tensor([seed_shared, step], dtype=torch.int64, device=dev), ops.SITE_NOISE_D)
        flat = flat.cpu()
        per = {k: flat[eng.disc.offsets[k]:eng.disc.offsets[k] + dp[k].numel()].view(dp[k].shape) for k in eng.disc.keys}
        noisy = dpsgd.add_noise(out['grads'], per, sigma, clip, n)
        dp = onets.apply_update(dp, noisy, sd)
        cmp_d.steps += 1
        for k in eng.disc.keys:
            cmp_d.check(k, eng.disc.P(k), dp[k].numpy(), noisy[k].numpy(), 'step %d D param %s' % (step, k))
        # ---- generator update (update 1 of the step)
        e, zz, keeps, al = draws(step, 1, n)
        outg = onets.g_step_grads(gp, dp, span, pac, zz, eng.cg.cond.cpu(), eng.cg.mask.cpu(), e, keeps)
        gp = onets.apply_update(gp, outg['grads'], sg)
        cmp_g.steps += 1
        for k in eng.gen.keys:
            if 'fc.bias' in k:      # exactly-zero gradient in front of BatchNorm: Adam turns rounding noise into +-lr steps
                assert np.abs(eng.gen.P(k).cpu().numpy() - gp[k].numpy()).max() < 2e-4 * cmp_g.steps
            else:
                cmp_g.check(k, eng.gen.P(k), gp[k].numpy(), outg['grads'][k].numpy(), 'step %d G param %s' % (step, k))
    assert len(gs.items) == 1 and int(eng.disc.t) == 3 and int(eng.gen.t) == 3


def test_lazy_cond_and_mask_equal_what_the_dense_launch_writes():
    """A discriminator update that carries the cond vector as two hot ids does not write the dense cond matrix and
    the mask (engine.GenContext.cond): whoever reads them afterwards - after an eager update and after graph replays,
    which rewrite the ids without running the engine's Python code - gets what the dense-writing launch of an engine
    with the same seeds leaves (the draws depend on the seeds and the step count only)."""
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import GraphStep, StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')

    def make(lazy):
        eng = StepEngine(span, 24, (40, 24), (48, 16), 120, 10, dev, private=True, dp_mode='dpsgd', seed=5)
        eng.lazy_cond = lazy
        eng.sampler = DataSampler(enc, span, device=dev, seed=4)
        return eng
    a, b = make(True), make(False)
    a.d_step()
    b.d_step()
    assert a.cd._cond_stale == bool(a.sparse_cond and a.sparse_d) and not b.cd._cond_stale      # (the direct step form writes them anyway)
    assert torch.equal(a.cd.cond, b.cd.cond) and torch.equal(a.cd.mask, b.cd.mask)
    assert a.cd.cond.sum().item() == 2 * a.B
    ga, gb = GraphStep(a, 1), GraphStep(b, 1)
    for _ in range(4):
        ga.run()
        gb.run()
        assert torch.equal(a.cd.cond, b.cd.cond) and torch.equal(a.cd.mask, b.cd.mask)
        assert torch.equal(a.cg.cond, b.cg.cond) and torch.equal(a.cg.mask, b.cg.mask)


def test_penalty_product_behind_the_group_equals_the_one_beside_it():
    """StepEngine.pen_serial (wide first layers: (u^T diag(f c) u) W1 added to W1's gradient AFTER the grouped
    weight-gradient launch with plain stores, instead of beside it with atomics on both sides): same gradient of
    every discriminator tensor on the same draws, to the 2e-5 gradient bar."""
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')
    grads = []
    for serial in (False, True):
        eng = StepEngine(span, 24, (40, 24), (48, 16), 120, 10, dev, private=True, dp_mode='dpsgd', seed=5)
        eng.pen_serial = serial
        eng.sampler = DataSampler(enc, span, device=dev, seed=4)
        eng.d_step(update=False)
        torch.cuda.synchronize()
        grads.append({k: eng.disc.G(k).clone() for k in eng.disc.keys})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-30), k
    assert float(grads[0]['seq.0.weight'].abs().max()) > 0


def span_cond_width(span_info):
    from oracle.nets import discrete_columns
    return sum(k for _, _, k in discrete_columns(span_info))


def test_reproducible_mode_generator_gradient_sees_the_cond_columns():
    """deterministic=True keeps the dense weight-gradient products of the generator, which read the cond columns of
    Hg - so the input launch of a GENERATOR update must still write them there (it skips them in the default mode,
    where nothing reads them).  Same seeds, Philox-drawn inputs: the two modes' generator gradients agree to the
    2e-5 gradient bar, cond columns included."""
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')
    grads = []
    for det in (False, True):
        eng = StepEngine(span, 24, (40, 24), (48, 16), 120, 10, dev, private=False, seed=5, deterministic=det)
        eng.sampler = DataSampler(enc, span, device=dev, seed=4)
        eng.g_step(update=False)
        torch.cuda.synchronize()
        grads.append({k: eng.gen.G(k).clone() for k in eng.gen.keys})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        if k.endswith('fc.bias'):        # a Linear bias in front of BatchNorm has gradient zero: rounding noise on both sides
            assert float(a.abs().max()) < 1e-7 and float(b.abs().max()) < 1e-7, k
            continue
        assert float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-30), k
    w0 = grads[1]['seq.0.fc.weight']
    assert float(w0[:, -span_cond_width(span):].abs().max()) > 0


@pytest.mark.parametrize('mode', ['dpsgd', 'reference'])
def test_reproducible_mode_discriminator_gradient_equals_the_default_mode(mode):
    """The reproducible mode (dense products, fixed-order reductions, no sparse first layer) and the default mode
    (hot-id forms, grouped launches, atomics) are two routes to the same discriminator gradient: same seeds,
    Philox-drawn inputs, every tensor to the 2e-5 gradient bar."""
    from golden_util import load
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.engine import StepEngine
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    dev = torch.device('cuda:0')
    grads = []
    for det in (False, True):
        eng = StepEngine(span, 24, (40, 24), (48, 16), 120, 10, dev, private=True, dp_mode=mode, seed=5, deterministic=det)
        eng.sampler = DataSampler(enc, span, device=dev, seed=4)
        eng.d_step(update=False)
        torch.cuda.synchronize()
        grads.append({k: eng.disc.G(k).clone() for k in eng.disc.keys})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        assert float((a - b).abs().max()) <= 2e-5 * max(float(a.abs().max()), 1e-30), k

