"""DataSampler on the device: same interface as the reference class
(/root/reference/src/dp_cgans/synthesizers/data_sampler.py:6-406) over bitmap / co-occurrence
tables instead of per-pair row-id lists (see csrc/sampler.hip for the data layout and why)."""
import numpy as np
import torch

from . import ops
from .engine import span_tables


class DataSampler(object):
    """DataSampler samples the conditional vector and corresponding data (device-resident).

    ``output_info`` is either the reference's list of lists of SpanInfo(dim, activation_fn) or the
    flattened int array [n_spans, 3] = (dim, is_softmax, spans_in_column).
    """

    def __init__(self, data, output_info, log_frequency=True, device='cuda:0', seed=0, reference_quirks=True,
                 superblock_bytes=4 << 30):
        self.dev = torch.device(device)
        # The reference never fills _discrete_column_matrix_st (data_sampler.py:19-20), so
        # sample_original_condvec / generate_cond_from_condition_column_info read column 0's span for
        # every discrete column.  reference_quirks=True keeps that; False uses the real offsets.
        self.reference_quirks = reference_quirks
        span_info = flatten_output_info(output_info)
        self.span_info = span_info
        _, cols, _, _ = span_tables(span_info)
        self._n_discrete_columns = len(cols)
        self._n_categories = int(sum(c[2] for c in cols))
        self._categories_each_column = np.array([c[2] for c in cols], dtype=np.int64)
        self._matrix_st = np.array([c[0] for c in cols], dtype=np.int64)
        self._log_frequency = log_frequency     # kept for interface parity; the pair path never reads it
        if torch.is_tensor(data):
            self.table = data.to(self.dev, torch.float32).contiguous()
        else:
            self.table = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(self.dev)
        self.N, self.Dd = self.table.shape
        if self.Dd != int(span_info[:, 0].sum()):
            raise AssertionError('data width does not match output_info')       # data_sampler.py:42
        self.rng = ops.new_rng(seed ^ 0x5a17, self.dev)
        nd, C = self._n_discrete_columns, self._n_categories
        if nd == 0:
            return
        cat_col = np.concatenate([np.arange(ds, ds + k) for ds, _, k in cols]).astype(np.int32)
        self.cat_col = torch.from_numpy(cat_col).to(self.dev)
        self.cond_st = torch.tensor([c[1] for c in cols], dtype=torch.int32, device=self.dev)
        self.dims = torch.tensor([c[2] for c in cols], dtype=torch.int32, device=self.dev)
        self._matrix_st_dev = torch.tensor([c[0] for c in cols], dtype=torch.int32, device=self.dev)
        self.W = (self.N + 63) // 64
        self.bitmaps = torch.zeros(C, self.W, dtype=torch.int64, device=self.dev)
        self.cnt = torch.zeros(C, C, dtype=torch.int32, device=self.dev)
        # superblock split of the counts for the row pick: S <= 64 superblocks, at least 16 bitmap words
        # each, at most ~4 GiB of table (288 GB of HBM: 20 MB at T1M, 3 superblocks at C = 18 000)
        S = int(max(1, min(64, self.W // 16, superblock_bytes // (4 * C * C))))
        self.cntblk = torch.zeros(C, C, S, dtype=torch.int32, device=self.dev) if S > 1 else None
        ops.build_bitmaps(self.table, self.cat_col, self.bitmaps)
        ops.cooccurrence(self.bitmaps, self.cnt, self.cntblk)
        self._last_sel = None

    def dim_cond_vec(self):
        return self._n_categories

    # -- reference-shaped host API (numpy in / numpy out), computed by the kernels -----------------
    def sample_condvec_pair(self, batch, pairs=None, r=None):
        """-> cond [B,C] f32, mask [B,nd] i32, pair_id [B] i64, val_idx [B] i64 (or None without
        discrete columns).  ``pairs`` / ``r`` inject the draws (parity tests)."""
        if self._n_discrete_columns == 0:
            return None
        d, B = self.dev, batch
        cond = torch.empty(B, self._n_categories, device=d)
        mask = torch.empty(B, self._n_discrete_columns, dtype=torch.int32, device=d)
        pid = torch.empty(B, dtype=torch.int64, device=d)
        vid = torch.empty(B, dtype=torch.int64, device=d)
        sel = torch.empty(B, 2, dtype=torch.int32, device=d)
        gcount = torch.empty(B, dtype=torch.int64, device=d)
        pj = torch.as_tensor(np.asarray(pairs, dtype=np.int32)).to(d) if pairs is not None else None
        rj = torch.as_tensor(np.asarray(r, dtype=np.float64)).to(d) if r is not None else None
        ops.condvec_pair_sample(self.cnt, self.cond_st, self.dims, self.N, B, [cond], mask, pid, vid, sel, gcount,
                                pairs_inj=pj, r_inj=rj, rng=self.rng)
        ops.state_advance(self.rng)
        out = cond.cpu().numpy(), mask.cpu().numpy(), pid.cpu().numpy(), vid.cpu().numpy()
        self._last_sel = (sel, gcount, out[2], out[3])
        return out

    def sample_data_pair(self, n, col, opt, k=None):
        """Rows satisfying the pairs of the latest sample_condvec_pair call, requested in the order
        ``col``/``opt`` give (the reference passes pair_id[perm], val_idx[perm])."""
        d = self.dev
        if col is None:
            idx = torch.empty(n, dtype=torch.float64, device=d)
            ops.fill(idx, ops.FILL_UNIFORM64, self.rng, ops.SITE_ROWPICK)
            ops.state_advance(self.rng)
            rid = (idx * self.N).long().clamp_(max=self.N - 1)
            return self.table[rid].double().cpu().numpy()
        # (pair, value) ids -> the category pair / group size recorded by the latest condvec call
        sel, gcount, pid, vid = self._last_sel
        key = pid.astype(np.int64) * (1 << 32) + vid.astype(np.int64)
        uniq, first = np.unique(key, return_index=True)
        want = np.asarray(col, dtype=np.int64) * (1 << 32) + np.asarray(opt, dtype=np.int64)
        pos = np.searchsorted(uniq, want)
        if (pos >= len(uniq)).any() or (uniq[np.minimum(pos, len(uniq) - 1)] != want).any():
            raise KeyError('sample_data_pair: (pair, value) ids must come from the latest sample_condvec_pair call')
        src = torch.from_numpy(first[pos].astype(np.int64)).to(d)
        kd = torch.as_tensor(np.asarray(k, dtype=np.int64)).to(d) if k is not None else None
        rid = torch.empty(n, dtype=torch.int64, device=d)
        ops.row_select(self.bitmaps, sel, gcount, rid, perm=src, k_inj=kd, rng=self.rng, cntblk=self.cntblk)
        out = torch.empty(n, self.Dd, device=d)
        ops.row_gather(self.table, rid, out, self.Dd)
        ops.state_advance(self.rng)
        self.last_row_ids = rid.cpu().numpy()
        return out.double().cpu().numpy()

    def rows_for(self, perm, k=None):
        """Device form of sample_data_pair(B, pair_id[perm], val_idx[perm]): -> (row_id, rows fp32)."""
        sel, gcount = self._last_sel[:2]
        d = self.dev
        B = sel.shape[0]
        permd = torch.as_tensor(np.asarray(perm, dtype=np.int64)).to(d)
        kd = torch.as_tensor(np.asarray(k, dtype=np.int64)).to(d) if k is not None else None
        rid = torch.empty(B, dtype=torch.int64, device=d)
        ops.row_select(self.bitmaps, sel, gcount, rid, perm=permd, k_inj=kd, rng=self.rng, cntblk=self.cntblk)
        out = torch.empty(B, self.Dd, device=d)
        ops.row_gather(self.table, rid, out, self.Dd)
        ops.state_advance(self.rng)
        return rid.cpu().numpy(), out.cpu().numpy()

    def sample_original_condvec_device(self, batch, rows=None, cols=None):
        """Cond vector for generation (data_sampler.py:348-363) as a device tensor [batch, C].  The
        reference reads column 0's span for every discrete column (its _discrete_column_matrix_st is
        never filled); reference_quirks=True keeps that."""
        if self._n_discrete_columns == 0:
            return None
        d = self.dev
        cond = torch.empty(batch, self._n_categories, device=d)
        mst = None if self.reference_quirks else self._matrix_st_dev
        rj = torch.as_tensor(np.asarray(rows, dtype=np.int64)).to(d) if rows is not None else None
        cj = torch.as_tensor(np.asarray(cols, dtype=np.int64)).to(d) if cols is not None else None
        ops.condvec_original(self.table, self.cond_st, self.dims, mst, cond, rows_inj=rj, cols_inj=cj, rng=self.rng)
        ops.state_advance(self.rng)
        return cond

    def sample_original_condvec(self, batch, rows=None, cols=None):
        """Reference-shaped host API: numpy float32 [batch, C] (or None without discrete columns)."""
        cond = self.sample_original_condvec_device(batch, rows, cols)
        return None if cond is None else cond.cpu().numpy()

    def generate_cond_from_condition_column_info(self, condition_info, batch):
        """data_sampler.py:401-406 (the reference adds value_id to a matrix offset that is always 0, Q9)."""
        vec = np.zeros((batch, self._n_categories), dtype='float32')
        c = condition_info['discrete_column_id']
        if self.reference_quirks:
            idx = condition_info['value_id']
        else:
            idx = int(self.cond_st[c].item()) + condition_info['value_id']
        vec[:, idx] = 1
        return vec


def flatten_output_info(output_info):
    """list of lists of SpanInfo(dim, activation_fn) -> int array [n,3]; arrays pass through."""
    if isinstance(output_info, np.ndarray):
        return output_info.astype(np.int32)
    rows = []
    for col in output_info:
        for sp in col:
            dim, fn = (sp.dim, sp.activation_fn) if hasattr(sp, 'dim') else (sp[0], sp[1])
            if fn not in ('tanh', 'softmax'):
                raise ValueError('Unexpected activation function %s.' % fn)      # dp_cgan.py:325
            rows.append((int(dim), 1 if fn == 'softmax' else 0, len(col)))
    return np.array(rows, dtype=np.int32).reshape(-1, 3)
