"""The public API on the GPU: DataTransformer.transform bit-exact against the reference fixture,
the reference's own smoke test (tests/test_dp_cgans.py:10-36) on the Adult CSV, private modes,
save/load, the saved-generator format."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from golden_util import GOLDEN, load

pytestmark = pytest.mark.gpu


def golden_transformer(z, device):
    from dp_cgans_amd.data_transformer import (ColumnTransformInfo, DataTransformer, GaussianMixtureTable,
                                               OneHotCategories, SpanInfo)
    cols = [str(c) for c in z['columns']]
    disc = set(str(c) for c in z['discrete_columns'])
    table = pd.DataFrame(z['table'], columns=cols)
    tr = DataTransformer(device=device)
    tr.output_info_list, tr.output_dimensions, tr.dataframe = [], 0, True
    tr._column_raw_dtypes = table.infer_objects().dtypes
    tr._column_transform_info_list = []
    for ci, name in enumerate(cols):
        if name in disc:
            ohe = OneHotCategories(table[name])
            info = ColumnTransformInfo(name, 'discrete', ohe, None, [SpanInfo(len(ohe.dummies), 'softmax')], len(ohe.dummies))
        else:
            g = lambda k: z['gm%d/%s' % (ci, k)]
            gt = GaussianMixtureTable(g('means'), g('covariances'), g('precisions_cholesky'),
                                      (g('weight_concentration_a'), g('weight_concentration_b')), g('mean_precision'),
                                      g('degrees_of_freedom'), g('valid'))
            nv = int(g('valid').sum())
            info = ColumnTransformInfo(name, 'continuous', gt, g('valid'), [SpanInfo(1, 'tanh'), SpanInfo(nv, 'softmax')], 1 + nv)
        tr.output_info_list.append(info.output_info)
        tr.output_dimensions += info.output_dimensions
        tr._column_transform_info_list.append(info)
    return tr, table


def test_transform_bit_exact_against_reference_fixture():
    """DataTransformer.transform (data_transformer.py:133-150) with the fitted mixtures and the
    per-row uniforms the reference used: the fp64 encoded matrix must be identical."""
    z = load('adult_encode.npz')
    tr, table = golden_transformer(z, 'cuda:0')
    cont = [int(c) for c in z['continuous_columns']]
    us = z['choice_u'].reshape(len(cont), -1)
    uniforms = {table.columns[ci]: us[j] for j, ci in enumerate(cont)}
    enc = tr.transform(table, uniforms=uniforms)
    assert enc.dtype == np.float64 and np.array_equal(enc, z['encoded'])
    enc32 = tr.transform_device(table, uniforms=uniforms)
    assert enc32.dtype == torch.float32 and np.array_equal(enc32.cpu().numpy(), z['encoded'].astype(np.float32))
    # Philox-driven picks: valid one-hots, values inside the clip range
    free = tr.transform(table)
    assert free.shape == enc.shape and np.isfinite(free).all() and np.abs(free).max() <= 1.0


@pytest.fixture()
def in_tmp(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)      # fit writes fitted_transformer.pkl / loss_output_*.txt to the cwd (dp_cgan.py:483,547)
    return tmp_path


def adult():
    return pd.read_csv(os.path.join(GOLDEN, 'adult.csv'))


def test_reference_smoke_test_fit_and_sample_adult(in_tmp):
    """tests/test_dp_cgans.py:10-36 of the reference (fewer epochs): fit on the Adult CSV, batch 100,
    dims (128,128), discriminator_steps 5, then 100 sampled rows with the input's schema."""
    from dp_cgans_amd import DP_CGAN
    df = adult()
    model = DP_CGAN(epochs=4, batch_size=100, log_frequency=True, verbose=False, generator_dim=(128, 128),
                    discriminator_dim=(128, 128), generator_lr=2e-4, discriminator_lr=2e-4, discriminator_steps=5,
                    private=False)
    model.fit(df)
    out = model.sample(100)
    assert len(out) == 100 and list(out.columns) == list(df.columns)
    for c in df.columns:
        assert out[c].dtype == df[c].dtype, c
        if df[c].dtype == object:
            assert set(out[c].dropna().unique()) <= set(df[c].dropna().unique())
        else:
            assert out[c].min() >= df[c].min() and out[c].max() <= df[c].max()
    synth = model._model
    assert len(synth.loss_values) == 4 and list(synth.loss_values.columns) == ['Epoch', 'Generator Loss', 'Distriminator Loss']
    assert np.isfinite(synth.loss_values[['Generator Loss', 'Distriminator Loss']].to_numpy()).all()
    assert os.path.exists('fitted_transformer.pkl') and any(f.startswith('loss_output_') for f in os.listdir('.'))
    # through the wrapper: 24 synthesizer columns, 19 binary discrete ones, 38 categories
    assert synth._data_sampler._n_discrete_columns == 19 and synth._data_sampler.dim_cond_vec() == 38
    # ---- saved-generator format: keys / shapes of the reference Generator
    sd = synth._generator.state_dict()
    E, C = 128, 38
    Dd = synth._transformer.output_dimensions
    want = {'seq.0.fc.weight': (128, E + C), 'seq.0.fc.bias': (128,), 'seq.0.bn.weight': (128,), 'seq.0.bn.bias': (128,),
            'seq.0.bn.running_mean': (128,), 'seq.0.bn.running_var': (128,), 'seq.0.bn.num_batches_tracked': (),
            'seq.1.fc.weight': (128, E + C + 128), 'seq.1.fc.bias': (128,), 'seq.1.bn.weight': (128,),
            'seq.1.bn.bias': (128,), 'seq.1.bn.running_mean': (128,), 'seq.1.bn.running_var': (128,),
            'seq.1.bn.num_batches_tracked': (), 'seq.2.weight': (Dd, E + C + 256), 'seq.2.bias': (Dd,)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert sd['seq.0.bn.num_batches_tracked'].dtype == torch.int64
    assert int(sd['seq.0.bn.num_batches_tracked']) == 4 * 2 * 6 + 2     # 6 generator forwards per step + 2 sample batches
    dsd = synth._discriminator.state_dict()
    assert {k: tuple(v.shape) for k, v in dsd.items()} == {
        'seq.0.weight': (128, 10 * (Dd + C)), 'seq.0.bias': (128,), 'seq.3.weight': (128, 128), 'seq.3.bias': (128,),
        'seq.6.weight': (1, 128), 'seq.6.bias': (1,)}
    # a torch module laid out like the reference Generator (dp_cgan.py:71-100) loads it strictly and
    # reproduces the engine's forward
    gen = _torch_generator(E + C, (128, 128), Dd)
    gen.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
    eng = synth._engine
    cond = torch.as_tensor(synth._data_sampler.sample_original_condvec(eng.B)).to(eng.dev)
    eng.sample_forward(cond)
    x = eng.Hg[:, eng.start[0]:].cpu()
    ref = gen.train()(x)
    err = (eng.fake.cpu() - ref.detach()).abs().max() / ref.abs().max()
    assert err < 1e-5, err
    # ---- save / load round trip (base.py:500-527)
    sd = synth._generator.state_dict()          # (the forward above moved the running statistics)
    model.save('m.pkl')
    m2 = DP_CGAN.load('m.pkl')
    for k, v in m2._model._generator.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].cpu()), k
    out2 = m2.sample(50)
    assert len(out2) == 50 and list(out2.columns) == list(df.columns)
    # xai_discriminator scores len(data_samples) rows (dp_cgan.py:899-915): [rows / pac, 1], whatever the batch size
    score = model.xai_discriminator(df.iloc[:10].T)
    assert score.shape == (1, 1) and torch.isfinite(score).all()
    score = model.xai_discriminator(df.iloc[:150].T)         # more rows than one batch of 100
    assert score.shape == (15, 1) and torch.isfinite(score).all()
    with pytest.raises(AssertionError):
        model.xai_discriminator(df.iloc[:7].T)               # Discriminator.forward's pac assert (dp_cgan.py:67)
    # ---- device contract: cuda=False warns once and runs on the GPU; a CPU device string raises
    from dp_cgans_amd import DPCGANSynthesizer
    DPCGANSynthesizer._warned_no_cpu = False
    with pytest.warns(UserWarning, match='no CPU path'):
        s2 = DPCGANSynthesizer(cuda=False)
    assert s2._device.type == 'cuda'
    with pytest.raises(ValueError, match='not a GPU'):
        synth.set_device('cpu')
    synth.set_device('cuda:0')                              # same device: nothing to rebuild
    assert synth._engine is eng


def test_ops_refuse_a_call_whose_tensors_are_not_on_the_current_device():
    """Launches go to the current stream of the tensors' GPU; with another GPU current the wrappers raise
    instead of launching on the wrong device's stream (needs two visible GPUs)."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible GPUs')
    from dp_cgans_amd import DPCGANSynthesizer, ops
    x = torch.zeros(8, 8, device='cuda:1')
    with torch.cuda.device(0):
        with pytest.raises(RuntimeError, match='current device'):
            ops.fill(x, ops.FILL_UNIFORM, ops.new_rng(1, 'cuda:1'), 1)
    rng = np.random.default_rng(0)
    df = pd.DataFrame({'a': rng.normal(size=300), 'c': rng.integers(3, size=300).astype(str)})
    s = DPCGANSynthesizer(epochs=2, batch_size=60, pac=10, generator_dim=(32,), discriminator_dim=(32,), embedding_dim=16,
                          cuda='cuda:1', seed=3)
    with torch.cuda.device(0):                               # the caller's current device is NOT the model's
        s.fit(df, discrete_columns=['c'])
        out = s.sample(40)
    assert out.shape == (40, 2) and s._engine.gen.p.device.index == 1


def _torch_generator(in_dim, gdims, data_dim):
    import torch.nn as nn

    class Residual(nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.fc, self.bn, self.relu = nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU()

        def forward(self, x):
            return torch.cat([self.relu(self.bn(self.fc(x))), x], dim=1)

    class Generator(nn.Module):
        def __init__(self):
            super().__init__()
            seq, d = [], in_dim
            for w in gdims:
                seq.append(Residual(d, w))
                d += w
            seq.append(nn.Linear(d, data_dim))
            self.seq = nn.Sequential(*seq)

        def forward(self, x):
            return self.seq(x)
    return Generator()


@pytest.mark.parametrize('dp_mode', ['reference', 'dpsgd'])
def test_private_fit_both_modes(in_tmp, dp_mode):
    """BASELINE configs[1]: Adult, private=True, batch 500, pac 10 (200 rows < batch: 1 step per epoch, Q1)."""
    from dp_cgans_amd import DP_CGAN
    df = adult()
    model = DP_CGAN(epochs=6, batch_size=500, pac=10, private=True, dp_mode=dp_mode, dp_sigma=1.0, dp_clip=1.0)
    model.fit(df)
    synth = model._model
    assert np.isfinite(synth.loss_values[['Generator Loss', 'Distriminator Loss']].to_numpy()).all()
    assert np.isfinite(synth.epsilon) and synth.epsilon >= 0
    p = synth._engine.disc.p
    assert torch.isfinite(p).all()
    if dp_mode == 'reference':
        assert float(p.abs().max()) <= 0.01 + 1e-9          # weight clamp, dp_cgan.py:657-661
    assert len(model.sample(30)) == 30


def test_numeric_only_table_without_discrete_columns(in_tmp):
    """No discrete columns: no cond vector, raw generator output goes to the discriminator (Q5)."""
    from dp_cgans_amd import DPCGANSynthesizer
    rng = np.random.default_rng(0)
    df = pd.DataFrame({'a': rng.normal(size=300), 'b': rng.normal(3, 2, size=300)})
    s = DPCGANSynthesizer(epochs=3, batch_size=60, pac=10, generator_dim=(32,), discriminator_dim=(32,), embedding_dim=16)
    s.fit(df)
    out = s.sample(40)
    assert out.shape == (40, 2) and np.isfinite(out.to_numpy()).all()


def test_sample_path_kernels_match_the_host_restatement():
    """sample_original_condvec (data_sampler.py:348-363, with and without the zero-offset quirk) and
    inverse_transform (data_transformer.py:152-213) on the device against numpy, bit for bit."""
    from dp_cgans_amd.data_sampler import DataSampler
    z = load('adult_encode.npz')
    tr, table = golden_transformer(z, 'cuda:0')
    enc, span = z['encoded'], z['span_info']
    g = np.random.default_rng(3)
    B = 500
    rows, cols = g.integers(0, enc.shape[0], B), g.integers(0, 19, B)
    for quirks in (True, False):
        ds = DataSampler(enc, span, device='cuda:0', reference_quirks=quirks)
        cond = ds.sample_original_condvec(B, rows=rows, cols=cols)
        want = np.zeros((B, ds._n_categories), dtype=np.float32)
        dims, cst, mst = ds._categories_each_column, ds.cond_st.cpu().numpy(), ds._matrix_st
        for i in range(B):
            st = 0 if quirks else int(mst[cols[i]])
            pick = int(np.argmax(enc[rows[i], st:st + dims[cols[i]]]))
            want[i, pick + cst[cols[i]]] = 1
        assert cond.dtype == np.float32 and np.array_equal(cond, want), quirks
        free = ds.sample_original_condvec(B)
        assert (free.sum(1) == 1).all() and set(np.unique(free)) == {0.0, 1.0}
    # decode: random "generator outputs" (tanh range values, soft one-hots with ties)
    Dd = enc.shape[1]
    fake = g.random((1000, Dd)).astype(np.float32)
    fake[:, 0] = g.uniform(-1.2, 1.2, 1000)               # beyond the clip range too
    fake[::7, 1:3] = 0.5                                   # ties: first maximum wins
    host = tr.inverse_transform(fake)
    devd = tr.inverse_transform_device(torch.from_numpy(fake).to('cuda:0'))
    assert list(host.columns) == list(devd.columns)
    for c in host.columns:
        assert host[c].dtype == devd[c].dtype, c
        assert np.array_equal(host[c].to_numpy(), devd[c].to_numpy()), c


def test_sample_path_against_reference_fixture():
    """SURVEY 8f-1 pinned to the reference: sample_original_condvec with the reference's recorded randint
    draws (cond vectors bit-exact, zero-offset quirk Q9 included) and inverse_transform of a
    generated-looking matrix (argmax over the valid components, clip to +-1, de-normalise), on the
    device kernel and on the host path."""
    from dp_cgans_amd.data_sampler import DataSampler
    ze, zs, zf = load('adult_encode.npz'), load('adult_sampler.npz'), load('adult_sample.npz')
    ds = DataSampler(zs['encoded'], zs['span_info'], device='cuda:0')
    cond = ds.sample_original_condvec(len(zf['soc/rows']), rows=zf['soc/rows'], cols=zf['soc/cols'])
    assert cond.dtype == np.float32 and np.array_equal(cond, zf['soc/cond'])
    tr, _ = golden_transformer(ze, 'cuda:0')
    gen = zf['inv/in']
    want = zf['inv/out']
    host = tr.inverse_transform(gen.astype(np.float64))
    assert list(host.columns) == [str(c) for c in zf['inv/columns']]
    np.testing.assert_array_equal(host.to_numpy(dtype=np.float64), want)
    dev = tr.inverse_transform_device(torch.from_numpy(gen).to('cuda:0'))
    np.testing.assert_array_equal(dev.to_numpy(dtype=np.float64), want)
