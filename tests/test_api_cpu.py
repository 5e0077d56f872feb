"""Host logic of the public wrapper (no GPU): constructor surface, table pre-processing, category
order, inverse transform, error behaviour.  Reference: dp_cgan_modular.py:73-113, base.py:125-139,
data_transformer.py:152-239, Transformers/transformers.py:92-118."""
import inspect
import os

import numpy as np
import pandas as pd
import pytest

from golden_util import GOLDEN, load

REFERENCE_KWARGS = dict(
    field_names=None, field_types=None, field_transformers=None, anonymize_fields=None, primary_key=None,
    constraints=None, table_metadata=None, embedding_dim=128, generator_dim=(256, 256), discriminator_dim=(256, 256),
    generator_lr=2e-4, generator_decay=1e-6, discriminator_lr=2e-4, discriminator_decay=1e-6, batch_size=500,
    discriminator_steps=1, log_frequency=True, verbose=False, epochs=300, pac=10, cuda=True, rounding='auto',
    min_value='auto', max_value='auto', private=False, wandb=False, ontology=None, saved_transformer=None)


def test_constructor_surface_matches_reference():
    from dp_cgans_amd import DP_CGAN, DPCGANSynthesizer
    sig = inspect.signature(DP_CGAN.__init__).parameters
    names = list(sig)[1:]
    assert names[:len(REFERENCE_KWARGS)] == list(REFERENCE_KWARGS)       # same order: positional calls keep working
    for k, v in REFERENCE_KWARGS.items():
        assert sig[k].default == v, k
    ssig = inspect.signature(DPCGANSynthesizer.__init__).parameters
    for k in ('embedding_dim', 'generator_dim', 'discriminator_dim', 'generator_lr', 'generator_decay',
              'discriminator_lr', 'discriminator_decay', 'batch_size', 'discriminator_steps', 'log_frequency',
              'verbose', 'epochs', 'pac', 'cuda', 'private', 'wandb', 'ontology'):
        assert ssig[k].default == REFERENCE_KWARGS[k], k
    with pytest.raises(AssertionError):
        DPCGANSynthesizer(batch_size=101)                                 # dp_cgan.py:241
    m = DP_CGAN(batch_size=100)
    assert m._model_kwargs['batch_size'] == 100 and m._model_kwargs['pac'] == 10


def test_device_and_seed_contract_of_the_constructor():
    """cuda=False (the reference's own test passes it, tests/test_dp_cgans.py:27) warns and keeps the GPU;
    a CPU device string raises; seed defaults to None (OS entropy); a fixed seed with private=True warns
    that the noise becomes reproducible; a batch that does not split into pac groups per rank raises."""
    import warnings
    from dp_cgans_amd import DP_CGAN, DPCGANSynthesizer
    assert inspect.signature(DPCGANSynthesizer.__init__).parameters['seed'].default is None
    assert inspect.signature(DP_CGAN.__init__).parameters['seed'].default is None
    DPCGANSynthesizer._warned_no_cpu = False
    with pytest.warns(UserWarning, match='no CPU path'):
        s = DPCGANSynthesizer(cuda=False)
    assert s._device.type == 'cuda'
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        DPCGANSynthesizer(cuda=False)                                  # one warning per process
        DPCGANSynthesizer(private=True)                                # no seed: nothing to warn about
    for dev in ('cpu', 'cpu:0'):
        with pytest.raises(ValueError, match='not a GPU'):
            DPCGANSynthesizer(cuda=dev)
    assert DPCGANSynthesizer(cuda='cuda:3')._device.index == 3
    with pytest.warns(UserWarning, match='voids the privacy guarantee'):
        DPCGANSynthesizer(private=True, seed=1)
    a, b = DPCGANSynthesizer()._base_seed(), DPCGANSynthesizer()._base_seed()
    assert a != b and 0 <= a < 2 ** 63
    assert DPCGANSynthesizer(seed=5)._base_seed() == 5
    with pytest.raises(ValueError, match='multiple of world size x pac'):
        DPCGANSynthesizer(batch_size=500, pac=8)._build_engine(np.array([[2, 1, 1]], dtype=np.int32))
    from dp_cgans_amd.synthesizer import _derive_seed
    assert _derive_seed(5, b'transformer') != 5 and _derive_seed(5, b'transformer') == _derive_seed(5, b'transformer')


def test_table_preprocessing_matches_what_the_reference_hands_to_the_synthesizer():
    """Through the reference wrapper the Adult frame becomes 24 float columns, 19 of them binary
    one-hot columns treated as discrete; fixture captured from the reference."""
    from dp_cgans_amd.model import _TableLite
    z = load('adult_encode.npz')
    df = pd.read_csv(os.path.join(GOLDEN, 'adult.csv'))
    t = _TableLite()
    t.fit(df)
    out = t.transform(df)
    assert list(out.columns) == [str(c) for c in z['columns']]
    assert np.array_equal(out.to_numpy(dtype=np.float64), z['table'])
    disc = [c for c in out.columns if c not in t.get_fields() and set(out[c].unique()) == {0.0, 1.0}]
    assert disc == [str(c) for c in z['discrete_columns']]
    back = t.reverse_transform(out)
    assert list(back.columns) == list(df.columns)
    for c in df.columns:
        a, b = back[c], df[c]
        assert ((a == b) | (a.isna() & b.isna())).all(), c
    assert back['age'].dtype == df['age'].dtype


def test_inverse_transform_and_condition_lookup():
    from dp_cgans_amd.data_transformer import (ColumnTransformInfo, DataTransformer, GaussianMixtureTable,
                                               OneHotCategories, SpanInfo)
    z = load('adult_encode.npz')
    cols = [str(c) for c in z['columns']]
    disc = set(str(c) for c in z['discrete_columns'])
    table = pd.DataFrame(z['table'], columns=cols)
    tr = DataTransformer(device='cpu')
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
    assert np.array_equal(tr.span_info(), z['span_info'])
    enc = z['encoded']
    back = tr.inverse_transform(enc)
    # discrete columns come back exactly; continuous ones up to the +-0.99 clip of the encode
    for name in cols:
        if name in disc:
            assert np.array_equal(back[name].to_numpy(), table[name].to_numpy()), name
        else:
            ci = cols.index(name)
            st = int(sum(i.output_dimensions for i in tr._column_transform_info_list[:ci]))
            unclipped = np.abs(enc[:, st]) < 0.99
            assert unclipped.any()
            np.testing.assert_allclose(back[name].to_numpy()[unclipped], table[name].to_numpy()[unclipped], rtol=1e-9,
                                       atol=1e-9)
    info = tr.convert_column_name_value_to_id('sex.value0', 0.0)
    assert info['discrete_column_id'] == 17 and info['column_id'] == cols.index('sex.value0') and info['value_id'] in (0, 1)
    with pytest.raises(ValueError):
        tr.convert_column_name_value_to_id('nope', 1)
    with pytest.raises(ValueError):
        tr.convert_column_name_value_to_id('sex.value0', 7.0)


def test_one_hot_category_order_nulls_last():
    from dp_cgans_amd.data_transformer import OneHotCategories
    o = OneHotCategories(pd.Series(['b', None, 'a', 'b', 'c']))
    assert o.dummies[:3] == ['b', 'a', 'c'] and pd.isna(o.dummies[3])
    assert o.codes(pd.Series(['a', 'zzz', None, 'c'])).tolist() == [1, -1, 3, 2]


def test_validation_errors():
    from dp_cgans_amd import DPCGANSynthesizer
    s = DPCGANSynthesizer(batch_size=20)
    with pytest.raises(ValueError, match='Invalid columns found'):
        s._validate_discrete_columns(pd.DataFrame({'a': [1, 2]}), ['b'])
    with pytest.raises(TypeError):
        s._validate_discrete_columns([[1, 2]], [])
    with pytest.raises(ValueError, match='Invalid columns found'):
        s._validate_discrete_columns(np.zeros((3, 2)), [5])


def test_host_inverse_transform_against_reference_fixture():
    """DataTransformer.inverse_transform (host path, data_transformer.py:152-213) on the matrix the
    reference inverted when tests/golden/adult_sample.npz was captured: identical values."""
    from golden_util import load
    from test_api_gpu import golden_transformer
    ze, zf = load('adult_encode.npz'), load('adult_sample.npz')
    tr, _ = golden_transformer(ze, 'cpu')
    back = tr.inverse_transform(zf['inv/in'].astype(np.float64))
    assert list(back.columns) == [str(c) for c in zf['inv/columns']]
    np.testing.assert_array_equal(back.to_numpy(dtype=np.float64), zf['inv/out'])
