#!/usr/bin/env python3
"""Capture golden vectors from the *imported* reference (this container only).

Runs sunchang0124/dp_cgans from /root/reference/src on the 200-row UCI Adult
example, with every random draw on the hot path recorded in call order, and
writes small .npz fixtures under tests/golden/.  The fixtures are data only
(inputs, random draws, expected outputs); no reference source is stored.

Usage:  python tools/capture_goldens.py  [--out tests/golden]

What is recorded (reference file:line the value comes from):
  * adult_encode.npz   DataTransformer.fit/transform (data_transformer.py:73-150):
                       table handed to DPCGANSynthesizer.fit, fitted BayesianGMM
                       parameters, the uniform behind every np.random.choice(p=),
                       the encoded matrix.
  * adult_sampler.npz  DataSampler tables (data_sampler.py:9-198) and one recorded
                       sample_condvec_pair / sample_data_pair call (:245-343,:382-396).
  * trace_{np,priv}.npz  two full training steps (dp_cgan.py:570-722) with all
                       draws, intermediate tensors, losses, gradients, weights.
  * rdp.npz            compute_rdp / get_privacy_spent grid (rdp_accountant.py:341,594).
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np

REF = '/root/reference'


def import_reference():
    g = types.ModuleType('gensim')
    m = types.ModuleType('gensim.models')
    m.KeyedVectors = object
    g.models = m
    sys.modules['gensim'] = g
    sys.modules['gensim.models'] = m
    sys.path.insert(0, os.path.join(REF, 'src'))
    import dp_cgans  # noqa: F401


class Rec:
    """Ordered log of random draws and intermediate values, bucketed by name."""

    def __init__(self):
        self.buckets = {}
        self.enabled = True

    def add(self, name, value):
        if self.enabled:
            self.buckets.setdefault(name, []).append(np.array(value, copy=True))

    def pop_all(self):
        b = self.buckets
        self.buckets = {}
        return b


REC = Rec()


def t2n(t):
    return t.detach().cpu().numpy().copy()


def install_patches():
    import torch
    import torch.nn.functional as F
    from dp_cgans.synthesizers import dp_cgan as ref
    from dp_cgans.synthesizers import data_sampler as ref_ds

    # ---- torch draws ------------------------------------------------------------------
    _normal, _rand, _randn = torch.normal, torch.rand, torch.randn
    _exp = torch.Tensor.exponential_

    def normal(*a, **k):
        out = _normal(*a, **k)
        REC.add('normal', t2n(out))
        return out

    def rand(*a, **k):
        out = _rand(*a, **k)
        REC.add('rand', t2n(out))
        return out

    def randn(*a, **k):
        out = _randn(*a, **k)
        REC.add('randn', t2n(out))
        return out

    def exponential_(self, *a, **k):
        out = _exp(self, *a, **k)
        REC.add('exp', t2n(out))
        return out

    def dropout(input, p=0.5, training=True, inplace=False):
        # Same function as F.dropout (mask ~ Bernoulli(1-p), scale 1/(1-p)) with the mask exposed.
        if not training or p == 0.0:
            return input
        keep = torch.bernoulli(torch.full_like(input, 1.0 - p))
        REC.add('dropmask', t2n(keep).astype(np.uint8))
        return input * keep * (1.0 / (1.0 - p))

    torch.normal, torch.rand, torch.randn = normal, rand, randn
    torch.Tensor.exponential_ = exponential_
    F.dropout = dropout
    torch.nn.functional.dropout = dropout

    # ---- numpy draws ------------------------------------------------------------------
    _choice, _nrand, _shuffle = np.random.choice, np.random.rand, np.random.shuffle

    def choice(a, size=None, replace=True, p=None):
        if p is not None:
            # One uniform per call (legacy RandomState.choice); recover it by replaying the state.
            st = np.random.get_state()
            out = _choice(a, size=size, replace=replace, p=p)
            st2 = np.random.get_state()
            np.random.set_state(st)
            u = np.random.random_sample()
            np.random.set_state(st2)
            REC.add('choice_p_u', u)
            REC.add('choice_p_out', out)
            return out
        out = _choice(a, size=size, replace=replace)
        REC.add('choice', out)
        return out

    def nrand(*shape):
        out = _nrand(*shape)
        REC.add('nprand', out)
        return out

    def shuffle(x):
        _shuffle(x)
        REC.add('shuffle', x)

    np.random.choice, np.random.rand, np.random.shuffle = choice, nrand, shuffle

    # ---- structured wrappers ----------------------------------------------------------
    S = ref.DPCGANSynthesizer
    _act, _cl = S._apply_activate, S._cond_loss_pair

    def apply_activate(self, data):
        out = _act(self, data)
        REC.add('fake', t2n(data))
        REC.add('fakeact', t2n(out))
        return out

    def cond_loss_pair(self, data, c, m):
        out = _cl(self, data, c, m)
        REC.add('ce', t2n(out))
        return out

    S._apply_activate, S._cond_loss_pair = apply_activate, cond_loss_pair

    D = ref.Discriminator
    _dfwd, _gp = D.forward, D.calc_gradient_penalty

    def dforward(self, input):
        out = _dfwd(self, input)
        REC.add('d_out', t2n(out))
        return out

    def gp(self, real_data, fake_data, device='cpu', pac=10, lambda_=10):
        out = _gp(self, real_data, fake_data, device, pac, lambda_)
        REC.add('pen', t2n(out))
        return out

    D.forward, D.calc_gradient_penalty = dforward, gp

    DS = ref_ds.DataSampler
    _scp, _sdp = DS.sample_condvec_pair, DS.sample_data_pair

    def scp(self, batch):
        out = _scp(self, batch)
        if out is not None:
            for k, v in zip(('cond', 'mask', 'pair_id', 'val_idx'), out):
                REC.add(k, v)
        return out

    def sdp(self, n, col, opt):
        out = _sdp(self, n, col, opt)
        REC.add('real', out)
        return out

    DS.sample_condvec_pair, DS.sample_data_pair = scp, sdp

    # ---- backward / hooks / optimiser --------------------------------------------------
    _backward = torch.Tensor.backward

    def backward(self, *a, **k):
        REC.add('backward_value', t2n(self))
        REC.add('hook_mark', np.array(-1.0))      # separates hook noise of successive backwards
        return _backward(self, *a, **k)

    torch.Tensor.backward = backward

    _reg = torch.Tensor.register_hook
    PARAM_IDS = {}

    def register_hook(self, fn):
        pid = PARAM_IDS.get(id(self), -1)

        def wrapped(grad):
            out = fn(grad)
            if pid >= 0:
                diff = (out - grad).reshape(-1)
                REC.add('hook_noise', np.array([pid, float(diff[0])]))
                REC.add('hook_mark', np.array(float(pid)))
            return out
        return _reg(self, wrapped)

    torch.Tensor.register_hook = register_hook

    _step = torch.optim.Adam.step

    def step(self, closure=None):
        ps = [p for g in self.param_groups for p in g['params']]
        REC.add('opt_nparams', len(ps))
        first = id(self) not in SEEN_OPT
        SEEN_OPT.add(id(self))
        for i, p in enumerate(ps):
            if first:                      # later steps: p_before == previous p_after
                REC.add('p_before_%d' % i, t2n(p))
            REC.add('g_%d' % i, t2n(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
        out = _step(self, closure)
        for i, p in enumerate(ps):
            REC.add('p_after_%d' % i, t2n(p))
        STEP_SINK.append(REC.pop_all())
        if LIMIT['n'] and len(STEP_SINK) >= LIMIT['n']:
            raise StopCapture()           # the trace holds what it was asked for: leave fit() here
        return out

    torch.optim.Adam.step = step

    # Register discriminator parameters for hook bookkeeping when it is built.
    _dinit = D.__init__

    def dinit(self, *a, **k):
        _dinit(self, *a, **k)
        for i, p in enumerate(self.parameters()):
            PARAM_IDS[id(p)] = i

    D.__init__ = dinit

    # Capture what DPCGANSynthesizer.fit receives, and the objects it builds.
    _fit = S.fit

    def fit(self, train_data, discrete_columns=tuple(), epochs=None):
        FIT_ARGS['train_data'] = train_data.copy()
        FIT_ARGS['discrete_columns'] = list(discrete_columns)
        FIT_ARGS['self'] = self
        return _fit(self, train_data, discrete_columns, epochs)

    S.fit = fit


STEP_SINK = []
SEEN_OPT = set()
FIT_ARGS = {}
LIMIT = {'n': 0}      # stop a fit after this many optimiser steps (0: run it to the end)


class StopCapture(Exception):
    pass


def flatten(prefix, bucket, out):
    for k, lst in bucket.items():
        for j, v in enumerate(lst):
            out['%s/%s/%d' % (prefix, k, j)] = v


def gm_params(gm):
    return dict(
        means=gm.means_.reshape(-1), covariances=gm.covariances_.reshape(-1),
        weights=gm.weights_, weight_concentration_a=gm.weight_concentration_[0],
        weight_concentration_b=gm.weight_concentration_[1], mean_precision=gm.mean_precision_,
        degrees_of_freedom=gm.degrees_of_freedom_,
        precisions_cholesky=gm.precisions_cholesky_.reshape(-1))


def run_trace(private, out_path, csv, numeric_only=False, batch_size=100, gen_dim=(48, 32), disc_dim=(32, 24),
              embedding_dim=16, d_steps=2, pac=10, max_opt_steps=0):
    """``max_opt_steps``: leave fit() after that many optimiser steps (the first iteration of BASELINE configs 1 and 2:
    their full-size parameter tensors make every recorded step a megabyte or more; config 2's epoch end would
    also raise in the reference's accountant at q = batch / rows = 2.5)."""
    import pandas as pd
    import torch
    from dp_cgans import DP_CGAN
    STEP_SINK.clear()
    SEEN_OPT.clear()
    REC.pop_all()
    np.random.seed(0)
    torch.manual_seed(0)
    df = pd.read_csv(csv)
    if numeric_only:          # a table without discrete columns (no cond vector, raw output to D)
        df = df.select_dtypes('number')
    model = DP_CGAN(epochs=1, batch_size=batch_size, generator_dim=tuple(gen_dim), discriminator_dim=tuple(disc_dim),
                    embedding_dim=embedding_dim, discriminator_steps=d_steps, pac=pac, private=private, cuda=False,
                    verbose=False)
    LIMIT['n'] = max_opt_steps
    try:
        model.fit(df)
    except StopCapture:
        pass
    finally:
        LIMIT['n'] = 0
    synth = FIT_ARGS['self']
    out = {}
    # Everything drawn before the first optimiser step that belongs to fit/transform is in step 0's
    # bucket; the step buckets are split at Adam.step boundaries (D,D,G,D,D,G for 2 steps/epoch).
    for s, bucket in enumerate(STEP_SINK):
        bucket.pop('choice_p_u', None)
        bucket.pop('choice_p_out', None)
        flatten('s%d' % s, bucket, out)
    out['n_opt_steps'] = np.array(len(STEP_SINK))
    sd = synth._generator.state_dict()
    for k, v in sd.items():
        out['gen_final/' + k] = t2n(v)
    for k, v in synth._discriminator.state_dict().items():
        out['disc_final/' + k] = t2n(v)
    info = []
    for col in synth._transformer.output_info_list:
        for sp in col:
            info.append((sp.dim, 0 if sp.activation_fn == 'tanh' else 1, len(col)))
    out['span_info'] = np.array(info, dtype=np.int32)   # (dim, is_softmax, spans_in_column)
    out['config'] = np.array([batch_size, embedding_dim, pac, d_steps, int(private)], dtype=np.int32)  # B,E,pac,d_steps,private
    out['gen_dim'] = np.array(list(gen_dim), np.int32)
    out['disc_dim'] = np.array(list(disc_dim), np.int32)
    np.savez_compressed(out_path, **out)
    return synth


def capture_encode_and_sampler(outdir, csv):
    """Fit on Adult through the public wrapper once more and dump transformer + sampler tables."""
    import pandas as pd
    import torch
    from dp_cgans import DP_CGAN
    STEP_SINK.clear()
    REC.pop_all()
    np.random.seed(1)
    torch.manual_seed(1)
    df = pd.read_csv(csv)
    model = DP_CGAN(epochs=1, batch_size=100, generator_dim=(16,), discriminator_dim=(16,),
                    embedding_dim=8, pac=10, private=False, cuda=False)
    model.fit(df)
    synth = FIT_ARGS['self']
    first = STEP_SINK[0]
    tr = synth._transformer
    table = FIT_ARGS['train_data']
    enc = {}
    enc['columns'] = np.array(list(table.columns))
    enc['discrete_columns'] = np.array(FIT_ARGS['discrete_columns'])
    enc['table'] = table.to_numpy(dtype=np.float64)
    enc['choice_u'] = np.array(first['choice_p_u'], dtype=np.float64)       # row-major per continuous column
    enc['choice_out'] = np.array(first['choice_p_out'], dtype=np.int64)
    cont_idx = []
    for ci, info in enumerate(tr._column_transform_info_list):
        if info.column_type == 'continuous':
            cont_idx.append(ci)
            for k, v in gm_params(info.transform).items():
                enc['gm%d/%s' % (ci, k)] = np.asarray(v, dtype=np.float64)
            enc['gm%d/valid' % ci] = np.asarray(info.transform_aux, dtype=np.bool_)
            x = table[[info.column_name]].values
            enc['gm%d/predict_proba' % ci] = info.transform.predict_proba(x)
        else:
            enc['ohe%d/dummies' % ci] = np.array(info.transform.dummies, dtype=np.float64)
    enc['continuous_columns'] = np.array(cont_idx, dtype=np.int32)
    enc['encoded'] = synth._data_sampler._data.astype(np.float64)
    info = []
    for col in tr.output_info_list:
        for sp in col:
            info.append((sp.dim, 0 if sp.activation_fn == 'tanh' else 1, len(col)))
    enc['span_info'] = np.array(info, dtype=np.int32)
    np.savez_compressed(os.path.join(outdir, 'adult_encode.npz'), **enc)

    ds = synth._data_sampler
    sm = {}
    sm['encoded'] = enc['encoded']
    sm['span_info'] = enc['span_info']
    sm['pair_prob'] = ds._discrete_column_pair_prob
    sm['pair_cond_st'] = ds._discrete_pair_cond_st
    sm['pair_n_category'] = ds._discrete_pair_n_category
    sm['column_cond_st'] = ds._discrete_column_cond_st
    sm['column_n_category'] = ds._discrete_column_n_category
    sm['categories_each_column'] = ds._categories_each_column
    sm['get_position'] = ds.get_position
    keys = sorted(ds.pair_id_dict.keys())
    sm['pair_id_keys'] = np.array(keys, dtype=np.int64)
    sm['pair_id_vals'] = np.array([ds.pair_id_dict[k] for k in keys], dtype=np.int64)
    sm['rid_pair_sizes'] = np.array([[len(x) for x in p] + [0] * (9 - len(p))
                                     for p in ds._rid_by_cat_cols_pair], dtype=np.int64)
    # One explicit recorded call, B=256 (with the permuted real-row draw as in dp_cgan.py:594-596).
    REC.pop_all()
    np.random.seed(7)
    B = 256
    cond, mask, pid, vid = ds.sample_condvec_pair(B)
    perm = np.arange(B)
    np.random.shuffle(perm)
    real = ds.sample_data_pair(B, pid[perm], vid[perm])
    b = REC.pop_all()
    sm['call/B'] = np.array(B)
    sm['call/pairs'] = np.array(b['choice'][:B], dtype=np.int64)            # ordered (first, second) draws
    sm['call/r'] = np.array(b['nprand'][0], dtype=np.float64)
    sm['call/cond'], sm['call/mask'], sm['call/pair_id'], sm['call/val_idx'] = cond, mask, pid, vid
    sm['call/perm'] = perm
    sm['call/row_ids'] = np.array(b['choice'][B:2 * B], dtype=np.int64)     # picked row ids
    sm['call/real'] = real
    np.savez_compressed(os.path.join(outdir, 'adult_sampler.npz'), **sm)

    # sample() path: sample_original_condvec with its randint draws recorded, and
    # inverse_transform of a generated-looking matrix (values beyond +-1 to exercise the clip).
    sp = {}
    drawn = []
    real_randint = np.random.randint

    def randint(*a, **k):
        v = real_randint(*a, **k)
        drawn.append(int(v))
        return v
    np.random.seed(11)
    np.random.randint = randint
    try:
        cond = ds.sample_original_condvec(128)
    finally:
        np.random.randint = real_randint
    sp['soc/rows'] = np.array(drawn[0::2], dtype=np.int64)
    sp['soc/cols'] = np.array(drawn[1::2], dtype=np.int64)
    sp['soc/cond'] = cond
    rs = np.random.RandomState(3)
    Dd = enc['encoded'].shape[1]
    gen = rs.rand(96, Dd)
    st = 0
    for col in tr.output_info_list:
        if len(col) == 2:                                   # continuous: value in (-1.3, 1.3)
            gen[:, st] = rs.uniform(-1.3, 1.3, 96)
        st += sum(x.dim for x in col)
    gen = gen.astype(np.float32)
    back = tr.inverse_transform(gen.astype(np.float64))
    sp['inv/in'] = gen
    sp['inv/out'] = back.to_numpy(dtype=np.float64)
    sp['inv/columns'] = np.array(list(back.columns))
    np.savez_compressed(os.path.join(outdir, 'adult_sample.npz'), **sp)
    return synth


def capture_rdp(outdir):
    from dp_cgans.functions.rdp_accountant import compute_rdp, get_privacy_spent
    orders = [1 + x / 10. for x in range(1, 100)]
    rows = []
    for q in (0.5, 0.1, 0.01, 100 / 32561.0, 1.0):
        for sigma in (1.0, 0.7, 2.0):
            for steps in (0, 1, 10, 1000):
                rdp = compute_rdp(q=q, noise_multiplier=sigma, steps=steps, orders=orders)
                eps, _, opt = get_privacy_spent(orders, rdp, target_delta=2e-6)
                rows.append((q, sigma, steps, eps, opt, rdp[0], rdp[50], rdp[98]))
    np.savez_compressed(os.path.join(outdir, 'rdp.npz'), grid=np.array(rows, dtype=np.float64),
                        orders=np.array(orders))


def capture_config_traces(out, csv, tmp):
    """The first iteration of the reference's loop at the hyper-parameters BASELINE.json names for configs 1 and 2:
    config 1 - batch 100, embedding 128, G/D dims (128,128,128), discriminator_steps 5 (tests/test_dp_cgans.py:17-28),
    private=False: 5 D updates + 1 G update; config 2 - batch 500, pac 10, default dims (256,256), private=True (the
    reference's hook noise + clamp): 1 D update + 1 G update."""
    run_trace(False, os.path.join(out, 'trace_cfg1.npz'), csv, batch_size=100, gen_dim=(128, 128, 128),
              disc_dim=(128, 128, 128), embedding_dim=128, d_steps=5, pac=10, max_opt_steps=6)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    run_trace(True, os.path.join(out, 'trace_cfg2.npz'), csv, batch_size=500, gen_dim=(256, 256),
              disc_dim=(256, 256), embedding_dim=128, d_steps=1, pac=10, max_opt_steps=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--only', default=None, choices=['configs'],
                    help="'configs': only the traces at BASELINE configs 1 and 2's own hyper-parameters (round 3)")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    csv = os.path.join(REF, 'resources', 'example_tabular_data_UCIAdult.csv')
    import_reference()
    install_patches()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)            # the reference writes fitted_transformer.pkl / loss_output_*.txt to cwd
        try:
            if args.only == 'configs':
                capture_config_traces(out, csv, tmp)
                return
            capture_rdp(out)
            capture_encode_and_sampler(out, csv)
            for f in os.listdir(tmp):
                os.remove(os.path.join(tmp, f))
            run_trace(False, os.path.join(out, 'trace_np.npz'), csv)
            for f in os.listdir(tmp):
                os.remove(os.path.join(tmp, f))
            run_trace(True, os.path.join(out, 'trace_priv.npz'), csv)
            for f in os.listdir(tmp):
                os.remove(os.path.join(tmp, f))
            run_trace(False, os.path.join(out, 'trace_num.npz'), csv, numeric_only=True)
        finally:
            os.chdir(cwd)
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == '__main__':
    main()
