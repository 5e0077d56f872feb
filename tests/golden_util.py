"""Helpers to read the fixtures written by tools/capture_goldens.py."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def gen_keys(n_res):
    ks = []
    for l in range(n_res):
        ks += ['seq.%d.fc.weight' % l, 'seq.%d.fc.bias' % l, 'seq.%d.bn.weight' % l, 'seq.%d.bn.bias' % l]
    return ks + ['seq.%d.weight' % n_res, 'seq.%d.bias' % n_res]


def disc_keys(n_hidden):
    ks = []
    for l in range(n_hidden + 1):
        ks += ['seq.%d.weight' % (3 * l), 'seq.%d.bias' % (3 * l)]
    return ks


class Trace(object):
    """Two reference training steps (d_steps=2): optimiser steps D,D,G,D,D,G."""

    def __init__(self, name):
        z = load(name)
        self.z = z
        self.B, self.E, self.pac, self.d_steps, self.private = [int(v) for v in z['config']]
        self.span_info = z['span_info']
        self.gen_dim = [int(v) for v in z['gen_dim']]
        self.disc_dim = [int(v) for v in z['disc_dim']]
        self.n_steps = int(z['n_opt_steps'])
        self.gkeys = gen_keys(len(self.gen_dim))
        self.dkeys = disc_keys(len(self.disc_dim))
        self.kinds = []
        for s in range(self.n_steps):
            self.kinds.append('D' if ('s%d/pen/0' % s) in z.files else 'G')
        self.has_cond = 's0/cond/0' in z.files        # False for a table without discrete columns (trace_num)

    def get(self, s, name, j=0):
        return self.z['s%d/%s/%d' % (s, name, j)]

    def count(self, s, name):
        n = 0
        while ('s%d/%s/%d' % (s, name, n)) in self.z.files:
            n += 1
        return n

    def params(self, s, which):
        keys = self.dkeys if self.kinds[s] == 'D' else self.gkeys
        return {k: self.get(s, '%s_%d' % (which, i)) for i, k in enumerate(keys)}

    def exp_full(self, s):
        """[B, data_dim] Exp(1) draws laid out at the softmax spans' columns (1 elsewhere)."""
        dd = int(self.span_info[:, 0].sum())
        out = np.ones((self.B, dd), dtype=np.float32)
        st, j = 0, 0
        for dim, sm, _ in self.span_info.tolist():
            if sm:
                out[:, st:st + dim] = self.get(s, 'exp', j)
                j += 1
            st += dim
        return out

    def masks(self, s):
        nh = len(self.disc_dim)
        m = [self.get(s, 'dropmask', j) for j in range(self.count(s, 'dropmask'))]
        return [m[i:i + nh] for i in range(0, len(m), nh)]

    def hook_noise(self, s):
        """Per-parameter total of the scalar noise the reference's grad hooks added in step s,
        split by backward pass: list (one per backward) of {param index: sum of randn draws}."""
        marks = [float(self.get(s, 'hook_mark', j)) for j in range(self.count(s, 'hook_mark'))]
        draws = [float(self.get(s, 'randn', j)[0]) for j in range(self.count(s, 'randn'))]
        out, di = [], 0
        for m in marks:
            if m < 0:
                out.append({})
            else:
                out[-1][int(m)] = out[-1].get(int(m), 0.0) + draws[di]
                di += 1
        assert di == len(draws)
        return out
