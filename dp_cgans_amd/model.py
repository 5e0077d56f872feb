"""DP_CGAN: the public tabular-model wrapper (reference:
/root/reference/src/dp_cgans/synthesizers/dp_cgan_modular.py:69-113 and base.py:94-139,357-407,500-527).

The reference routes the DataFrame through a vendored SDV ``Table`` + RDT HyperTransformer before
the synthesizer sees it.  Only the part of that pre-processing the training step depends on is
restated here (the rest of Table/constraints/copulas is out of scope):

* object / category / bool columns -> one float column ``<name>.value<i>`` per category (first
  appearance order, nulls last), each of which the synthesizer then treats as its own binary
  discrete column (dp_cgan_modular.py:36-48);
* integer / float columns -> float, reversed with the FloatFormatter rules for
  ``rounding='auto'``, ``min_value='auto'``, ``max_value='auto'``.
"""
import pickle

import numpy as np
import pandas as pd

from .data_transformer import OneHotCategories
from .synthesizer import DPCGANSynthesizer


class _TableLite(object):
    def __init__(self, field_names=None, rounding='auto', min_value='auto', max_value='auto'):
        self.field_names = field_names
        self.rounding, self.min_value, self.max_value = rounding, min_value, max_value
        self.fitted = False

    def fit(self, data):
        self.columns = [c for c in data.columns if self.field_names is None or c in self.field_names]
        self.dtypes = data[self.columns].dtypes
        self.fields = {}
        for c in self.columns:
            kind = self.dtypes[c].kind
            if kind in 'iuf':
                col = data[c].astype(float)
                meta = {'type': 'numerical', 'subtype': 'integer' if kind in 'iu' else 'float', 'mean': float(col.mean()),
                        'min': float(col.min()), 'max': float(col.max())}
                if kind in 'iu':
                    meta['decimals'] = 0
                else:
                    dec = 0
                    vals = col.dropna().to_numpy()
                    while dec < 15 and not np.allclose(np.round(vals, dec), vals, rtol=0, atol=1e-12):
                        dec += 1
                    meta['decimals'] = dec
                self.fields[c] = meta
            elif kind in 'Ob' or str(self.dtypes[c]) == 'category':
                self.fields[c] = {'type': 'categorical', 'ohe': OneHotCategories(data[c])}
            else:
                raise ValueError('Unsupported dtype {} in column {}'.format(self.dtypes[c], c))   # table.py:388-389
        self.fitted = True

    def get_fields(self):
        return self.fields

    def transform(self, data):
        out = {}
        for c in self.columns:
            meta = self.fields[c]
            if meta['type'] == 'numerical':
                out[c] = data[c].astype(float).fillna(meta['mean']).to_numpy()
            else:
                codes = meta['ohe'].codes(data[c])
                for i in range(len(meta['ohe'].dummies)):
                    out['%s.value%d' % (c, i)] = (codes == i).astype(float)
        return pd.DataFrame(out)

    def reverse_transform(self, data):
        out = {}
        for c in self.columns:
            meta = self.fields[c]
            if meta['type'] == 'numerical':
                v = np.asarray(data[c], dtype=float)
                if self.min_value == 'auto':
                    v = np.maximum(v, meta['min'])
                elif self.min_value is not None:
                    v = np.maximum(v, self.min_value)
                if self.max_value == 'auto':
                    v = np.minimum(v, meta['max'])
                elif self.max_value is not None:
                    v = np.minimum(v, self.max_value)
                if self.rounding == 'auto':
                    v = np.round(v, meta['decimals'])
                elif self.rounding is not None:
                    v = np.round(v, self.rounding)
                out[c] = v.astype(self.dtypes[c]) if meta['subtype'] == 'integer' else v
            else:
                n = len(meta['ohe'].dummies)
                block = np.stack([np.asarray(data['%s.value%d' % (c, i)], dtype=float) for i in range(n)], axis=1)
                out[c] = meta['ohe'].decode(block).to_numpy()
        return pd.DataFrame(out, columns=self.columns)


class DP_CGAN(object):
    """Drop-in for ``dp_cgans.DP_CGAN``: same constructor arguments and defaults
    (dp_cgan_modular.py:73-80), ``fit(DataFrame)``, ``sample(num_rows)``, ``save``/``load``.
    Extra keyword arguments of this build: ``dp_mode``, ``dp_sigma``, ``dp_clip``, ``seed`` (None = OS
    entropy; a fixed seed is a test hook that voids the privacy guarantee of ``private=True``)."""

    _MODEL_CLASS = DPCGANSynthesizer

    def __init__(self, field_names=None, field_types=None, field_transformers=None,
                 anonymize_fields=None, primary_key=None, constraints=None, table_metadata=None,
                 embedding_dim=128, generator_dim=(256, 256), discriminator_dim=(256, 256),
                 generator_lr=2e-4, generator_decay=1e-6, discriminator_lr=2e-4,
                 discriminator_decay=1e-6, batch_size=500, discriminator_steps=1,
                 log_frequency=True, verbose=False, epochs=300, pac=10, cuda=True,
                 rounding='auto', min_value='auto', max_value='auto', private=False,
                 wandb=False, ontology=None, saved_transformer=None, dp_mode='reference', dp_sigma=1.0,
                 dp_clip=1.0, seed=None):
        for name, arg in (('field_types', field_types), ('field_transformers', field_transformers),
                          ('anonymize_fields', anonymize_fields), ('primary_key', primary_key),
                          ('constraints', constraints), ('table_metadata', table_metadata)):
            if arg:
                raise NotImplementedError('%s: the SDV Table metadata layer is outside this build (SURVEY section 2); '
                                          'pass a plain DataFrame' % name)
        self._metadata = _TableLite(field_names, rounding, min_value, max_value)
        self._metadata_fitted = False
        self._num_rows = None
        self._model = None
        self._model_kwargs = {
            'embedding_dim': embedding_dim, 'generator_dim': generator_dim, 'discriminator_dim': discriminator_dim,
            'generator_lr': generator_lr, 'generator_decay': generator_decay, 'discriminator_lr': discriminator_lr,
            'discriminator_decay': discriminator_decay, 'batch_size': batch_size,
            'discriminator_steps': discriminator_steps, 'log_frequency': log_frequency, 'verbose': verbose,
            'epochs': epochs, 'pac': pac, 'cuda': cuda, 'private': private, 'wandb': wandb, 'ontology': ontology,
            'saved_transformer': saved_transformer, 'dp_mode': dp_mode, 'dp_sigma': dp_sigma, 'dp_clip': dp_clip,
            'seed': seed,
        }

    def _build_model(self):
        return self._MODEL_CLASS(**self._model_kwargs)

    def get_metadata(self):
        return self._metadata

    def fit(self, data):
        """Fit this model to the data (base.py:125-139 + dp_cgan_modular.py:18-55)."""
        if not self._metadata_fitted:
            self._metadata.fit(data)
            self._metadata_fitted = True
        self._num_rows = len(data)
        table_data = self._metadata.transform(data)
        self._model = self._build_model()
        categoricals = []
        fields = self._metadata.get_fields()
        for field in table_data.columns:
            if field in fields:
                if fields[field]['type'] == 'categorical':
                    categoricals.append(field)
            else:
                field_data = table_data[field].dropna()
                if set(field_data.unique()) == {0.0, 1.0}:
                    categoricals.append(field)          # booleans encoded as floats are modeled as bool
        self._model.fit(table_data, discrete_columns=categoricals)

    def sample(self, num_rows=None, randomize_samples=True, max_retries=100, max_rows_multiplier=10, conditions=None,
               float_rtol=0.01, graceful_reject_sampling=False):
        """Sample rows from this table (base.py:357-407).  Conditions are not supported by this
        model (dp_cgan_modular.py:57-61)."""
        if conditions is not None:
            raise NotImplementedError(f"{self._MODEL_CLASS} doesn't support conditional sampling.")
        if self._model is None:
            raise RuntimeError('fit the model before sampling')
        num_rows = num_rows or self._num_rows
        sampled = self._model.sample(num_rows)
        return self._metadata.reverse_transform(sampled)

    def xai_discriminator(self, data_samples):
        return self._model.xai_discriminator(data_samples)

    def save(self, path):
        """Save this model instance to the given path using pickle (base.py:500-508)."""
        with open(path, 'wb') as output:
            pickle.dump(self, output)

    @classmethod
    def load(cls, path):
        """Load a model that ``save`` of this class wrote (base.py:514-527)."""
        with open(path, 'rb') as f:
            return pickle.load(f)
