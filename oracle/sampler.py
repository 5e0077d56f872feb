"""Oracle (TEST INFRASTRUCTURE ONLY): DataSampler restated in numpy with injected draws.

Reference: /root/reference/src/dp_cgans/synthesizers/data_sampler.py
  tables            :9-198    (row-id lists per (pair, joint value); pair probability tables)
  condvec pairs     :245-343  (+ _random_choice_prob_pairs :209-212)
  real-row pick     :382-396
The reference draws with the global numpy RNG; here the draws are arguments:
  pairs [B,2]  the ordered result of np.random.choice(nd, 2, replace=False) per row
  r     [B]    np.random.rand(B)
  k     [B]    rank of the picked row inside the ascending row-id list of (pair, joint value)
"""
import numpy as np

from .nets import discrete_columns


def pack_pair(data, a, b):
    """MSB-first bit packing of the concatenated one-hots of two columns (data_sampler.py:64-65):
    value = sum_j bit_j * 2^(width-1-j), computed in fp64 exactly as reduce(2a+b)."""
    ds_a, _, ka = a
    ds_b, _, kb = b
    bits = np.concatenate([data[:, ds_a:ds_a + ka], data[:, ds_b:ds_b + kb]], axis=1)
    acc = np.zeros(len(data), dtype=np.float64)
    for j in range(bits.shape[1]):
        acc = 2.0 * acc + bits[:, j]
    return acc


class Tables(object):
    """Everything DataSampler.__init__ derives from the encoded matrix."""

    def __init__(self, data, span_info, with_row_lists=True):
        data = np.asarray(data, dtype=np.float64)
        cols = discrete_columns(span_info)
        nd = len(cols)
        self.data, self.cols, self.nd = data, cols, nd
        self.dims = np.array([c[2] for c in cols], dtype=np.int64)
        self.cond_st = np.array([c[1] for c in cols], dtype=np.int64)
        self.n_categories = int(self.dims.sum())
        n_pairs = nd * (nd - 1) // 2
        if nd >= 2:
            srt = np.sort(self.dims)
            width = int((srt[-1] + 1) * (srt[-2] + 1))       # data_sampler.py:121-127
        else:
            width = 0
        self.pair_prob = np.zeros((n_pairs, width), dtype=np.float64)
        self.pair_code = np.zeros((n_pairs, width), dtype=np.int32)
        self.pair_n = np.zeros(n_pairs, dtype=np.int32)
        self.pair_rows = []
        self.pair_of = {}
        p = 0
        for i in range(nd):
            for j in range(i + 1, nd):
                code = pack_pair(data, cols[i], cols[j])
                uniq, counts = np.unique(code, return_counts=True)
                self.pair_prob[p, :len(uniq)] = counts / np.sum(counts)
                self.pair_code[p, :len(uniq)] = uniq
                self.pair_n[p] = len(uniq)
                if with_row_lists:
                    self.pair_rows.append([np.where(code == u)[0] for u in uniq])
                self.pair_of[(i, j)] = p
                p += 1

    def pair_index(self, i, j):
        """Closed form of the nested-loop numbering (i < j)."""
        return i * self.nd - i * (i + 1) // 2 + (j - i - 1)


def sample_condvec_pair(tb, pairs, r):
    """-> cond [B,C] f32, mask [B,nd] i32, pair_id [B] i64, val_idx [B] i64."""
    pairs = np.asarray(pairs, dtype=np.int64)
    B = len(pairs)
    lo, hi = pairs.min(1), pairs.max(1)
    pair_id = np.array([tb.pair_of[(a, b)] for a, b in zip(lo, hi)], dtype=np.int64)
    cond = np.zeros((B, tb.n_categories), dtype=np.float32)
    mask = np.zeros((B, tb.nd), dtype=np.int32)
    mask[np.arange(B), lo] = 1
    mask[np.arange(B), hi] = 1
    cum = tb.pair_prob[pair_id].cumsum(axis=1)
    val_idx = (cum > np.asarray(r)[:, None]).argmax(axis=1).astype(np.int64)
    code = tb.pair_code[pair_id, val_idx].astype(np.int64)
    for b in range(B):
        ka, kb = int(tb.dims[lo[b]]), int(tb.dims[hi[b]])
        v = int(code[b])
        pa = pb = 0
        if v != 0:
            # first set bit, scanning from the MSB side, inside each column's segment
            for t in range(ka):
                if (v >> (ka + kb - 1 - t)) & 1:
                    pa = t
                    break
            for t in range(kb):
                if (v >> (kb - 1 - t)) & 1:
                    pb = t
                    break
        cond[b, tb.cond_st[lo[b]] + pa] = 1
        cond[b, tb.cond_st[hi[b]] + pb] = 1
    return cond, mask, pair_id, val_idx


def rank_of_row(tb, pair_id, val_idx, row_ids):
    """k such that pair_rows[pair][val][k] == row id (turns a recorded pick into a rank)."""
    return np.array([int(np.searchsorted(tb.pair_rows[p][v], rid))
                     for p, v, rid in zip(pair_id, val_idx, row_ids)], dtype=np.int64)


def sample_data_pair(tb, pair_id, val_idx, k):
    """The k-th smallest matching row id per request, and the gathered rows (fp64)."""
    idx = np.array([tb.pair_rows[p][v][kk] for p, v, kk in zip(pair_id, val_idx, k)], dtype=np.int64)
    return idx, tb.data[idx]


def group_sizes(tb, pair_id, val_idx):
    return np.array([len(tb.pair_rows[p][v]) for p, v in zip(pair_id, val_idx)], dtype=np.int64)
