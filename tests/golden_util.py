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


RTOL = 1e-5
SHARES_LOG = os.environ.get('DPCGAN_EXCUSED_LOG')      # a file every AdamAwareCompare appends its excused shares to
LIVE = []      # the comparers a test created: tests/conftest.py reports each at the test's end


class AdamAwareCompare(object):
    """Adam's update lr*m/(sqrt(v)+eps) is sign-like: for a gradient entry of rounding-noise size
    (|g| within ~100 eps of zero, e.g. a unit that dropout silenced in every row) the update is
    decided by summation order, on the reference's side too.  Entries whose reference gradient is
    below 1e-5 of the tensor's largest are therefore held to |diff| <= lr per step taken so far, all
    others to 1e-5 relative.  ``first_step_sensitivity`` (single-update checks of wide layers, where
    thousands of entries sit near Adam's eps = 1e-8): at t = 1 the update is lr*g/(|g|+eps), so
    d(update)/dg = lr*eps/(|g|+eps)^2; entries where a gradient error at the gradient's own bar (2e-5 of
    the largest, checked separately and without any excuse) moves the update by more than half the
    parameter bar are held to the looser bound as well.

    What the excuse covers is bounded and on record: ``ceiling`` is the largest share of one tensor's entries
    (tensors of 64 entries or more) a test lets it cover - exceeded, the check fails, with the share in the message -
    and ``report`` hands the largest share seen to pytest's ``record_property`` (and to the file named by
    DPCGAN_EXCUSED_LOG), so a green run shows how much of each comparison ran at the looser bound."""

    def __init__(self, lr=2e-4, first_step_sensitivity=False, ceiling=0.02):
        self.loose, self.steps, self.lr, self.sens, self.ceiling = {}, 0, lr, first_step_sensitivity, ceiling
        self.worst, self.worst_key = 0.0, None
        LIVE.append(self)

    def check(self, key, mine, ref_p, ref_g, what, mine_g=None):
        # (exact zeros - e.g. cond columns no row of the batch selected - are NOT excused: both sides must leave them
        # alone; with ``mine_g``, the one exception is a sum that cancels to exactly 0 in the reference's summation
        # order and to rounding residue - below 1e-6 of the tensor's largest entry - in this one)
        ref_p, ref_g = np.asarray(ref_p), np.asarray(ref_g)
        gmax, sc = max(np.abs(ref_g).max(), 1e-30), max(np.abs(ref_p).max(), 1e-30)
        thr = 1e-5 * gmax
        if self.sens:
            assert self.steps == 1
            thr = max(thr, np.sqrt(self.lr * 1e-8 * 2e-5 * gmax / (0.5 * RTOL * sc)))
        tiny = (np.abs(ref_g) < thr) & (ref_g != 0)
        if mine_g is not None:
            mg = np.abs(mine_g.detach().cpu().numpy() if hasattr(mine_g, 'detach') else np.asarray(mine_g))
            tiny |= (ref_g == 0) & (mg != 0) & (mg < 1e-6 * gmax)
        self.loose[key] = tiny | self.loose.get(key, np.zeros_like(tiny))
        lo = self.loose[key]
        mine = mine.detach().cpu().numpy() if hasattr(mine, 'detach') else np.asarray(mine)
        err = np.abs(mine - ref_p)
        assert (err[~lo] / sc).max(initial=0.0) <= RTOL, '%s: %.3e' % (what, (err[~lo] / sc).max())
        assert err[lo].max(initial=0.0) <= self.lr * self.steps, '%s (noise-sized grads): %.3e' % (what, err[lo].max())
        share = float(lo.mean()) if lo.size >= 64 else 0.0
        if share > self.worst:
            self.worst, self.worst_key = share, what
        assert share <= self.ceiling, ('%s: %.4f of the entries were held to the looser |diff| <= lr bound, above this '
                                       "test's ceiling of %.4f" % (what, share, self.ceiling))

    def report(self, record_property, name):
        """Largest excused share of a tensor so far -> the test report (junit property) and the shares log."""
        if record_property is not None:
            record_property('excused_share[%s]' % name, round(self.worst, 6))
        if SHARES_LOG:
            with open(SHARES_LOG, 'a') as f:
                f.write('%s\t%.6f\t%s\tceiling %.4f\n' % (name, self.worst, self.worst_key, self.ceiling))
        return self.worst


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
