"""DataTransformer: mode-specific normalisation of continuous columns and one-hot encoding of
discrete ones.  Same interface as the reference class
(/root/reference/src/dp_cgans/synthesizers/data_transformer.py:16-239); ``fit`` runs the reference's
variational Bayesian mixture (:38-47, scikit-learn's algorithm) for all continuous columns at once on
the device (gmm_fit.py, csrc/gmm.hip), ``transform`` runs on the device (csrc/sampler.hip: encode_continuous_kernel,
encode_onehot_kernel) and ``inverse_transform`` restates :152-213 in numpy.
"""
from collections import namedtuple

import numpy as np
import pandas as pd
import torch
from scipy.special import digamma

from . import ops

SpanInfo = namedtuple('SpanInfo', ['dim', 'activation_fn'])
ColumnTransformInfo = namedtuple(
    'ColumnTransformInfo', ['column_name', 'column_type', 'transform', 'transform_aux', 'output_info',
                            'output_dimensions'])


class OneHotCategories(object):
    """Categories of one discrete column in first-appearance order, nulls last - the order the
    reference's OneHotEncoder builds (Transformers/transformers.py:92-118)."""

    def __init__(self, series):
        values = pd.Series(series).reset_index(drop=True)
        null = pd.isna(values).to_numpy()
        self.uniques = list(pd.unique(values[~null]))
        self.has_null = bool(null.any())
        self.dummies = list(self.uniques) + ([np.nan] if self.has_null else [])

    def codes(self, series):
        """int32 code per row; -1 for a category unseen at fit time (encodes as all zeros)."""
        values = pd.Series(series).reset_index(drop=True)
        codes = pd.Categorical(values, categories=self.uniques).codes.astype(np.int32)
        if self.has_null:
            codes[pd.isna(values).to_numpy()] = len(self.uniques)
        return codes

    def decode(self, onehot):
        idx = np.argmax(onehot, axis=1)
        return pd.Series(np.asarray(self.dummies, dtype=object)[idx])


class GaussianMixtureTable(object):
    """The numbers of a fitted BayesianGaussianMixture that the encode needs, as a [4, K] fp64 table
    (mu, sigma, precision_cholesky, A) plus the indices of the kept components.  A_k collects every
    x-independent term of sklearn's weighted log-probability (dirichlet_process weights, one
    feature, 'full' covariance):  log w_k + log pc_k - .5 log 2pi - .5 log dof_k
    + .5 (log 2 + psi(dof_k / 2) - 1 / mean_precision_k)."""

    def __init__(self, means, covariances, precisions_cholesky, weight_concentration, mean_precision,
                 degrees_of_freedom, valid):
        mu = np.asarray(means, dtype=np.float64).reshape(-1)
        self.means = mu
        self.stds = np.sqrt(np.asarray(covariances, dtype=np.float64).reshape(-1))
        pc = np.asarray(precisions_cholesky, dtype=np.float64).reshape(-1)
        a, b = (np.asarray(v, dtype=np.float64) for v in weight_concentration)
        dof = np.asarray(degrees_of_freedom, dtype=np.float64)
        stick = digamma(b) - digamma(a + b)
        logw = digamma(a) - digamma(a + b) + np.concatenate(([0.0], np.cumsum(stick)[:-1]))
        A = (logw + np.log(pc) - 0.5 * np.log(2.0 * np.pi) - 0.5 * np.log(dof)
             + 0.5 * (np.log(2.0) + digamma(0.5 * dof) - 1.0 / np.asarray(mean_precision, dtype=np.float64)))
        self.table = np.stack([mu, self.stds, pc, A])
        self.valid = np.asarray(valid, dtype=bool)
        self.valid_idx = np.nonzero(self.valid)[0].astype(np.int32)

    @classmethod
    def from_sklearn(cls, gm, valid):
        return cls(gm.means_, gm.covariances_, gm.precisions_cholesky_, gm.weight_concentration_, gm.mean_precision_,
                   gm.degrees_of_freedom_, valid)


class DataTransformer(object):
    """Data Transformer.

    Model continuous columns with a BayesianGMM and normalized to a scalar [0, 1] and a vector.
    Discrete columns are one-hot encoded.
    """

    def __init__(self, max_clusters=5, weight_threshold=0.001, device='cuda:0', seed=0):
        self._max_clusters = max_clusters
        self._weight_threshold = weight_threshold
        self._device = torch.device(device)
        self._seed = seed

    def _mixture(self):
        from .gmm_fit import DeviceBayesianGMM
        return DeviceBayesianGMM(n_components=self._max_clusters, weight_concentration_prior=1e-3, n_init=5,
                                 max_iter=2000, device=self._device, seed=self._seed)

    def _continuous_info(self, column_name, fit):
        valid = fit.weights > self._weight_threshold
        table = GaussianMixtureTable(fit.means, fit.covariances, fit.precisions_cholesky, fit.weight_concentration,
                                     fit.mean_precision, fit.degrees_of_freedom, valid)
        table.lower_bound, table.n_iter, table.converged, table.weights = fit.lower_bound, fit.n_iter, fit.converged, fit.weights
        num_components = int(valid.sum())
        return ColumnTransformInfo(
            column_name=column_name, column_type='continuous', transform=table,
            transform_aux=valid, output_info=[SpanInfo(1, 'tanh'), SpanInfo(num_components, 'softmax')],
            output_dimensions=1 + num_components)

    def _fit_continuous(self, column_name, raw_column_data):
        """Train Bayesian GMM for continuous column (data_transformer.py:36-57): the reference's
        BayesianGaussianMixture hyper-parameters (:38-47), fitted on the device (gmm_fit.py)."""
        x = np.asarray(raw_column_data, dtype=np.float64).reshape(1, -1)
        return self._continuous_info(column_name, self._mixture().fit_columns(x)[0])

    def _fit_discrete(self, column_name, raw_column_data):
        """Fit one hot encoder for discrete column (data_transformer.py:59-71)."""
        ohe = OneHotCategories(raw_column_data)
        n = len(ohe.dummies)
        return ColumnTransformInfo(column_name=column_name, column_type='discrete', transform=ohe, transform_aux=None,
                                   output_info=[SpanInfo(n, 'softmax')], output_dimensions=n)

    def fit(self, raw_data, discrete_columns=tuple()):
        """Fit GMM for continuous columns and One hot encoder for discrete columns; also counts the
        columns of the matrix data and records the span information (data_transformer.py:73-100)."""
        self.output_info_list = []
        self.output_dimensions = 0
        self.dataframe = True
        if not isinstance(raw_data, pd.DataFrame):
            self.dataframe = False
            discrete_columns = [str(c) for c in discrete_columns]
            raw_data = pd.DataFrame(raw_data, columns=[str(i) for i in range(raw_data.shape[1])])
        self._column_raw_dtypes = raw_data.infer_objects().dtypes
        self._column_transform_info_list = []
        # every continuous column and every initialisation in one batch of device fits
        cont = [c for c in raw_data.columns if c not in discrete_columns]
        fits = {}
        if cont:
            x = np.stack([np.asarray(raw_data[c].values, dtype=np.float64) for c in cont])
            fits = dict(zip(cont, self._mixture().fit_columns(x)))
        for column_name in raw_data.columns:
            if column_name in discrete_columns:
                info = self._fit_discrete(column_name, raw_data[column_name])
            else:
                info = self._continuous_info(column_name, fits[column_name])
            self.output_info_list.append(info.output_info)
            self.output_dimensions += info.output_dimensions
            self._column_transform_info_list.append(info)

    # ---- transform: device kernels ---------------------------------------------------------------
    def transform_device(self, raw_data, dtype=torch.float32, uniforms=None, chunk=1 << 22):
        """Encoded matrix [N, output_dimensions] as a device tensor (fp32 for training, fp64 for the
        reference-shaped ``transform``).  ``uniforms`` = {column name: fp64 [N]} injects the draw
        behind the reference's per-row ``np.random.choice(p=)`` (data_transformer.py:117)."""
        if not isinstance(raw_data, pd.DataFrame):
            raw_data = pd.DataFrame(raw_data, columns=[str(i) for i in range(raw_data.shape[1])])
        dev = self._device
        N = len(raw_data)
        out = torch.empty(N, self.output_dimensions, dtype=dtype, device=dev)
        o64, o32 = (out, None) if dtype == torch.float64 else (None, out)
        col = 0
        for j, info in enumerate(self._column_transform_info_list):
            if info.column_type == 'continuous':
                gt = info.transform
                tab = torch.from_numpy(gt.table).to(dev)
                vidx = torch.from_numpy(gt.valid_idx).to(dev)
                x_all = np.ascontiguousarray(raw_data[info.column_name].to_numpy(dtype=np.float64))
                u_all = None if uniforms is None else np.ascontiguousarray(uniforms[info.column_name], dtype=np.float64)
                # one Philox stream per column (the key), one step per row chunk: the 16-bit call-site field
                # carries neither, so no table size makes two chunks or columns share their draws
                rng = ops.new_rng((self._seed ^ 0xe2c0de) + 0x9E3779B97F4A7C15 * (j + 1), dev)
                for r0 in range(0, N, chunk):
                    r1 = min(N, r0 + chunk)
                    x = torch.from_numpy(x_all[r0:r1]).to(dev)
                    u = None if u_all is None else torch.from_numpy(u_all[r0:r1]).to(dev)
                    ops.encode_continuous(x, tab, vidx, out64=None if o64 is None else o64[r0:r1],
                                          out32=None if o32 is None else o32[r0:r1], col_off=col, u_inj=u, rng=rng,
                                          site=ops.SITE_ENCODE)
                    ops.state_advance(rng)
            else:
                codes_all = info.transform.codes(raw_data[info.column_name])
                for r0 in range(0, N, chunk):
                    r1 = min(N, r0 + chunk)
                    codes = torch.from_numpy(codes_all[r0:r1]).to(dev)
                    ops.encode_onehot(codes, info.output_dimensions, out64=None if o64 is None else o64[r0:r1],
                                      out32=None if o32 is None else o32[r0:r1], col_off=col)
            col += info.output_dimensions
        torch.cuda.synchronize(dev)
        return out

    def transform(self, raw_data, uniforms=None):
        """Take raw data and output a matrix data (numpy float64, as the reference returns)."""
        return self.transform_device(raw_data, dtype=torch.float64, uniforms=uniforms).cpu().numpy()

    # ---- inverse transform (host; data_transformer.py:152-213) -----------------------------------
    def _inverse_transform_continuous(self, info, column_data, sigmas, st):
        gt = info.transform
        value = column_data[:, 0]
        if sigmas is not None:
            value = np.random.normal(value, sigmas[st])
        value = np.clip(value, -1, 1)
        probs = np.ones((len(column_data), self._max_clusters)) * -100
        probs[:, gt.valid] = column_data[:, 1:]
        k = np.argmax(probs, axis=1)
        return value * 4 * gt.stds[k] + gt.means[k]

    def inverse_transform(self, data, sigmas=None):
        """Take matrix data and output raw data (DataFrame or array, as given to ``fit``)."""
        st = 0
        cols, names = [], []
        for info in self._column_transform_info_list:
            dim = info.output_dimensions
            block = data[:, st:st + dim]
            if info.column_type == 'continuous':
                cols.append(self._inverse_transform_continuous(info, block, sigmas, st))
            else:
                cols.append(info.transform.decode(block).to_numpy())
            names.append(info.column_name)
            st += dim
        recovered = pd.DataFrame(dict(zip(names, cols)), columns=names).astype(self._column_raw_dtypes)
        return recovered if self.dataframe else recovered.values

    def _decode_tables(self):
        """(desc int32 [ncol,4], params fp64, Kc) for dpc_inverse_transform."""
        Kc = self._max_clusters
        desc, params, st = [], [], 0
        for info in self._column_transform_info_list:
            if info.column_type == 'continuous':
                gt = info.transform
                desc.append((0, st, info.output_dimensions, len(params)))
                params += list(gt.means) + list(gt.stds) + [float(v) for v in gt.valid_idx]
                params += [0.0] * (Kc - len(gt.valid_idx))
            else:
                desc.append((1, st, info.output_dimensions, 0))
            st += info.output_dimensions
        return (np.array(desc, dtype=np.int32).reshape(-1, 4), np.array(params if params else [0.0], dtype=np.float64), Kc)

    def inverse_transform_device(self, data):
        """``inverse_transform`` (sigmas=None) of a device fp32 matrix on the device: one kernel decodes
        every column (argmax + de-normalise), one [n, n_columns] fp64 matrix comes back to the host,
        where category indices become labels."""
        desc, params, Kc = self._decode_tables()
        dev = data.device
        out = torch.empty(data.shape[0], desc.shape[0], dtype=torch.float64, device=dev)
        ops.inverse_transform(data, torch.from_numpy(desc).to(dev), torch.from_numpy(params).to(dev), Kc, out)
        host = out.cpu().numpy()
        cols, names = [], []
        for j, info in enumerate(self._column_transform_info_list):
            if info.column_type == 'continuous':
                cols.append(host[:, j])
            else:
                dummies = np.asarray(info.transform.dummies, dtype=object)     # labels by category index
                cols.append(pd.Series(dummies[host[:, j].astype(np.int64)]).infer_objects().to_numpy())
            names.append(info.column_name)
        recovered = pd.DataFrame(dict(zip(names, cols)), columns=names).astype(self._column_raw_dtypes)
        return recovered if self.dataframe else recovered.values

    def convert_column_name_value_to_id(self, column_name, value):
        discrete_counter = 0
        column_id = 0
        for info in self._column_transform_info_list:
            if info.column_name == column_name:
                break
            if info.column_type == 'discrete':
                discrete_counter += 1
            column_id += 1
        else:
            raise ValueError(f"The column_name `{column_name}` doesn't exist in the data.")
        code = int(info.transform.codes([value])[0])
        if code < 0:
            raise ValueError(f"The value `{value}` doesn't exist in the column `{column_name}`.")
        return {'discrete_column_id': discrete_counter, 'column_id': column_id, 'value_id': code}

    def span_info(self):
        """Flattened output_info_list: int32 [n_spans, 3] = (dim, is_softmax, spans_in_column)."""
        rows = []
        for col in self.output_info_list:
            for sp in col:
                rows.append((sp.dim, 1 if sp.activation_fn == 'softmax' else 0, len(col)))
        return np.array(rows, dtype=np.int32).reshape(-1, 3)
