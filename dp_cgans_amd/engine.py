"""Host-side executor of one DP-CGAN training step on the HIP kernels.

Mirrors what DPCGANSynthesizer.fit does per step (reference dp_cgan.py:577-661 discriminator
update, :673-722 generator update) but issues hand-written gfx950 kernels through the C-ABI
instead of ATen ops + autograd.  torch is used for device memory and streams only.

Memory layout (all fp32, row-major, resident in HBM for the whole fit):
  gen params / grads / Adam m,v   one flat buffer each; state_dict tensors are views into it
  disc params / grads / Adam m,v  one flat buffer each
  Hg  [B, sum(g)+E+C]             generator activations, concat-free: [h_{L-1} | .. | h_0 | z | c];
                                  layer l reads the suffix that starts after its own output block,
                                  which is exactly torch.cat([out, input]) of Residual (dp_cgan.py:83).
                                  Two sets (GenContext): one for the discriminator update's forward,
                                  one for the generator update's, which runs beside the former
  Xd  [3B, Dd+C] = [3n, K0]       stacked discriminator input: rows 0..n-1 fake groups, n..2n-1 real
                                  groups, 2n..3n-1 interpolates (later overwritten by the penalty's
                                  v-chain so that one GEMM per layer yields loss + penalty weight grads)
  H_i, S_i, D_i [3n, d_i]         activations, slopes (mask*2*{1,.2}), backward deltas per hidden layer
  Zp [2n, d_0], G1 [d_0, d_0]     Gram-form penalty (gp_gram): first-layer plain products x W1^T of the fake
                                  and real rows, and W1 W1^T; the interpolates and the K0-wide penalty
                                  gradient are then never formed (third block of Xd unused)
  sparse_d (Gram-form penalty + two-hot cond vector): the discriminator input keeps only its DATA columns,
                                  Xd [2B, P] = [2n, pac*P] (P = Dd rounded up to 32, zero padded); the cond
                                  columns of every pac sub-row travel as two hot ids (GenContext.sel, sel_real)
                                  and their share of the first layer is row gathers of W1cT = W1[:, cond
                                  columns]^T [pac*C, d_0] (forward) and sorted row sums (weight gradient);
                                  W1 itself stays in the reference's layout, addressed through segment maps
"""
import contextlib
import math
import os

import numpy as np
import torch

from . import ops

GP_LAMBDA = 10.0
LEAK = 0.2
TAU = 0.2


def span_tables(span_info):
    """span_info int [n,3] (dim, is_softmax, spans_in_column) -> spans int32 [n,3] (offset, dim, kind),
    discrete columns [(data_off, cond_off, dim)], col2d / col2c int32 [Dd]."""
    spans, cols, st, cst = [], [], 0, 0
    for dim, sm, ncol in np.asarray(span_info).tolist():
        spans.append((st, dim, 1 if sm else 0))
        if sm and ncol == 1:
            cols.append((st, cst, dim))
            cst += dim
        st += dim
    Dd = st
    col2d = -np.ones(Dd, dtype=np.int32)
    col2c = np.zeros(Dd, dtype=np.int32)
    for d, (ds, cs, k) in enumerate(cols):
        col2d[ds:ds + k] = d
        col2c[ds:ds + k] = np.arange(cs, cs + k)
    return np.array(spans, dtype=np.int32).reshape(-1, 3), cols, col2d, col2c


class FlatParams(object):
    """Named views into one flat fp32 buffer (+ grads and Adam moments of the same layout)."""

    def __init__(self, shapes, device):
        self.keys = [k for k, _ in shapes]
        self.shapes = dict(shapes)
        sizes = [int(np.prod(s)) for _, s in shapes]
        # keep every tensor 16-byte aligned inside the flat buffer (float4 loads in the GEMM)
        self.offsets, off = {}, 0
        for k, n in zip(self.keys, sizes):
            self.offsets[k] = off
            off += (n + 3) // 4 * 4
        self.n = off
        self.p = torch.zeros(off, device=device)
        self.g = torch.zeros(off, device=device)
        self.m = torch.zeros(off, device=device)
        self.v = torch.zeros(off, device=device)
        self.t = torch.zeros(1, dtype=torch.int32, device=device)
        seg = [self.offsets[k] for k in self.keys] + [off]
        self.seg_off = torch.tensor(seg, dtype=torch.int64, device=device)
        self.loaded = True       # parameters were written from outside (init, load): derived copies are stale

    def view(self, buf, k):
        o = self.offsets[k]
        return buf[o:o + int(np.prod(self.shapes[k]))].view(self.shapes[k])

    def P(self, k):
        return self.view(self.p, k)

    def G(self, k):
        return self.view(self.g, k)

    def load(self, sd):
        for k in self.keys:
            self.P(k).copy_(torch.as_tensor(sd[k]).to(self.p.device, torch.float32))
        self.loaded = True

    def state_dict(self):
        return {k: self.P(k).detach().clone() for k in self.keys}


def _linear_init(w, b, gen):
    """torch.nn.Linear default init (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for both)."""
    bound = 1.0 / math.sqrt(w.shape[1])
    w.copy_((torch.rand(w.shape, generator=gen) * 2 - 1) * bound)
    b.copy_((torch.rand(b.shape, generator=gen) * 2 - 1) * bound)


def _zero_pool(spec, device):
    """One byte buffer (zeroed by a single fill) carved into typed tensors: spec = [(name, shape, dtype)]
    -> (buffer, {name: tensor}); every tensor starts on a 256-byte boundary."""
    off, places = 0, []
    for name, shape, dtype in spec:
        nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        places.append((name, shape, dtype, off, nbytes))
        off += (nbytes + 255) // 256 * 256
    buf = torch.zeros(max(off, 256), dtype=torch.uint8, device=device)
    views = {name: buf[o:o + nb].view(dtype).view(shape) for name, shape, dtype, o, nb in places}
    return buf, views


class NoDist(object):
    world = 1
    rank = 0
    active = False

    def allreduce(self, t):
        return t


class GenContext(object):
    """Buffers one generator forward writes: its inputs (z | cond inside ``Hg``, cond vector, mask and
    the sampler's per-row picks), activations, BatchNorm statistics, the raw output ``fake`` and the
    discriminator-input rows ``X`` [B, Dd+C] that receive the activated output and the cond vector.
    The discriminator update and the generator update own one each, so that the generator update's
    forward can run beside the discriminator update (StepEngine.train_step)."""

    def __init__(self, eng, X):
        d, B = eng.dev, eng.B
        f32 = dict(device=d, dtype=torch.float32)
        self.Hg = torch.zeros(B, (eng.Wtot + 3) // 4 * 4, **f32)[:, :eng.Wtot]       # rows padded to 16 bytes, as fake below
        self.Z = [torch.zeros(B, w, **f32) for w in eng.gdim]
        self.bn_mean = [torch.zeros(w, **f32) for w in eng.gdim]
        self.bn_invstd = [torch.zeros(w, **f32) for w in eng.gdim]
        ldd = (eng.Dd + 3) // 4 * 4          # rows padded to 16 bytes: keeps the GEMM's vector path
        self.fake = torch.zeros(B, ldd, **f32)[:, :eng.Dd]
        self.X = X
        self.X3 = X.view(eng.n, eng.K0x)
        self._cond = torch.zeros(B, max(eng.C, 1), **f32)
        self._cond_stale = False      # this context's input launch writes the two hot ids (`sel`) only: `cond` / `mask` are rebuilt from them on access
        self._mask = torch.zeros(B, max(eng.nd, 1), device=d, dtype=torch.int32)
        self._eng = eng
        self.pair_id = torch.zeros(B, device=d, dtype=torch.int64)
        self.val_idx = torch.zeros(B, device=d, dtype=torch.int64)
        self.sel = torch.zeros(B, 2, device=d, dtype=torch.int32)
        self.gcount = torch.zeros(B, device=d, dtype=torch.int64)
        # (row, hot id) pairs of sel sorted by id (ops.hot_sort): the backward's cond-column row sums
        self.order = torch.zeros(2 * B, device=d, dtype=torch.int32)
        self.sid = torch.zeros(2 * B, device=d, dtype=torch.int32)
        self.n_items = torch.zeros(1, device=d, dtype=torch.int32)
        maxF = max(eng.gdim + [1])
        # one BatchNorm-sum workspace per layer (the deferred running-statistics update re-reads them)
        self.part = [torch.zeros(64 * 2 * maxF, device=d, dtype=torch.float64) for _ in eng.gdim]
        self.sums = [None] * len(eng.gdim)       # (workspace, row chunks) of the latest forward's BatchNorm sums
        self.bnacc = None                        # fp64 [2, F] accumulators per layer (inside the engine's zero pools)


    @property
    def cond(self):
        """The dense two-hot cond matrix [B, C].  A discriminator update that keeps the cond columns off the matrix
        cores (sparse_cond + sparse_d) never reads it - the ids in ``sel`` are what travels - so its input launch does
        not write it (590 MB per update at config 5); whoever asks for it afterwards gets it built from the ids."""
        self._materialise()
        return self._cond

    @property
    def mask(self):
        """[B, nd] int32, 1 at the two discrete columns the cond vector was drawn for (the cond loss reads it); lazy
        like ``cond``."""
        self._materialise()
        return self._mask

    def _materialise(self):
        if self._cond_stale:
            ids = self.sel.long()
            self._cond.zero_()
            self._cond.scatter_(1, ids, 1.0)
            # discrete column of a cond id: the last column whose first id is <= it
            st = self._eng.sampler.cond_st.long()
            col = torch.searchsorted(st, ids.reshape(-1), right=True).reshape(ids.shape) - 1
            self._mask.zero_()
            self._mask.scatter_(1, col, 1)
            # (the flag stays: a replayed step graph rewrites the ids without running this Python code, so as long as
            # the context's inputs come from launches that skip the dense copies every access rebuilds them)

class StepEngine(object):
    GP_GRAM = None      # test hook: True / False overrides the automatic choice of the penalty's form
    SPARSE_COND = None  # test hook: False keeps the cond columns inside the generator's dense products

    def __init__(self, span_info, embedding_dim, generator_dim, discriminator_dim, batch_size, pac, device,
                 generator_lr=2e-4, generator_decay=1e-6, discriminator_lr=2e-4, discriminator_decay=1e-6,
                 private=False, seed=None, dist=None, global_batch=None, dp_mode='reference', dp_sigma=1.0, dp_clip=1.0,
                 weight_clip=None, gp_gram=None, sparse_cond=None, deterministic=None, bn_sync=True):
        self.dev = torch.device(device)
        self.dist = dist or NoDist()
        if seed is None:
            # OS entropy keys every stream (parameter init, batch draws, DP noise), rank 0's on every rank:
            # replicas must start from the same parameters and add the same noise.  A fixed seed is a test
            # hook - it makes the private mode's Gaussian noise a publicly computable sequence.
            seed = int.from_bytes(os.urandom(8), 'little') >> 1
            if self.dist.active:
                seed = self.dist.broadcast_int(seed, self.dev)
        self.span_info_np = np.asarray(span_info, dtype=np.int32).reshape(-1, 3)
        spans, cols, col2d, col2c = span_tables(span_info)
        self.Dd = int(spans[:, 1].sum())
        self.nd = len(cols)
        self.C = int(sum(c[2] for c in cols))
        self.E, self.B, self.pac = embedding_dim, batch_size, pac
        if batch_size % pac != 0:
            raise AssertionError('batch_size %% pac != 0')      # dp_cgan.py:67
        self.gB = global_batch or batch_size * self.dist.world
        # Data parallel, the generator's BatchNorm: bn_sync=True normalises with the statistics of the GLOBAL batch (one
        # all-reduce per layer and direction: the step equals the single-process step on the concatenated batch);
        # False is what DistributedDataParallel does by default - every rank normalises its own rows, the only exchanges
        # left are the two gradient all-reduces, and the step runs on the single-GPU schedule.
        self.bn_sync = bool(bn_sync)
        self.bn_dist = self.dist.active and self.bn_sync
        self.bn_count = float(self.gB if (self.bn_sync or not self.dist.active) else batch_size)
        self.n = batch_size // pac
        self.gn = self.gB // pac
        self.gdim, self.ddim = list(generator_dim), list(discriminator_dim)
        self.private = private
        if dp_mode not in ('reference', 'dpsgd'):
            raise ValueError("dp_mode must be 'reference' or 'dpsgd'")
        # private=True, dp_mode 'reference': the reference's grad-hook noise + weight clamp +-0.01
        # (dp_cgan.py:638-661).  dp_mode 'dpsgd': per-example (pac group) clipping to dp_clip and one
        # N(0, (dp_sigma*dp_clip)^2) add to the summed gradient; weight clamp only if weight_clip is given.
        self.dp_mode = dp_mode if private else 'off'
        self.dp_sigma, self.dp_clip = float(dp_sigma), float(dp_clip)
        if weight_clip is None:
            weight_clip = 0.01 if self.dp_mode == 'reference' else 0.0
        self.weight_clip = float(weight_clip)
        self._ar_hook = None
        self._head_pending = None
        # the generator's weight-gradient products as ONE grouped launch behind its backward chain (0.664 -> 0.655 ms per
        # step at T1M: before, three products of 17-58 us each overlapped one another and the chain's last kernels)
        self.group_gweights = os.environ.get('DPCGAN_GROUP_GW', '1') == '1'
        self.advance_in_colsum = os.environ.get('DPCGAN_ADV_COLSUM', '1') == '1'
        self._adv_pending, self._adv_done = None, False
        self.lazy_cond = os.environ.get('DPCGAN_LAZY_COND', '1') == '1'
        self.want_interp_scores = False      # compute the discriminator's scores of the penalty's interpolates (diagnostic: _defer_head)
        self._ar_fault = None
        self._early_hook = None
        # where train_step forks the generator update's forward off the last discriminator update: 'd_forward' (in front
        # of the discriminator's forward), 'd_forward_done' (behind its head kernel: the default - round 3, same-box A/B
        # at T1M: 0.7235 / 0.7124 / 0.7574 / 0.7745 ms per step for d_forward / d_forward_done / d_chain_done /
        # d_grads_done; with the faster products the lane no longer pays for slowing the discriminator's first-layer
        # product and forward chain), 'd_chain_done' (in front of the weight gradients), 'd_grads_done'
        self.early_at = os.environ.get('DPCGAN_EARLY_AT', 'd_forward_done')
        self._inline = False
        self.lr_g, self.wd_g, self.lr_d, self.wd_d = generator_lr, generator_decay, discriminator_lr, discriminator_decay
        d = self.dev
        self.spans = torch.from_numpy(spans).to(d)
        self.col2d = torch.from_numpy(col2d).to(d)
        self.col2c = torch.from_numpy(col2c).to(d)
        self.has_cond = self.nd > 0
        L, M = len(self.gdim), len(self.ddim)
        self.L, self.M = L, M
        self.Wtot = sum(self.gdim) + self.E + self.C
        self.start = [sum(self.gdim[l:]) for l in range(L)] + [0]      # start[l] = first input column of layer l
        self.row = self.Dd + self.C
        self.K0 = pac * self.row
        # bit-reproducible steps (no split-K atomics, no scatter atomics): read once from the process default
        self.deterministic = bool(ops.DETERMINISTIC) if deterministic is None else bool(deterministic)
        # ---- parameters ------------------------------------------------------------------------
        gs = []
        for l in range(L):
            fan_in = self.Wtot - self.start[l]
            gs += [('seq.%d.fc.weight' % l, (self.gdim[l], fan_in)), ('seq.%d.fc.bias' % l, (self.gdim[l],)),
                   ('seq.%d.bn.weight' % l, (self.gdim[l],)), ('seq.%d.bn.bias' % l, (self.gdim[l],))]
        gs += [('seq.%d.weight' % L, (self.Dd, self.Wtot)), ('seq.%d.bias' % L, (self.Dd,))]
        self.gen = FlatParams(gs, d)
        # The cond vector is one-/two-hot (data_sampler.py:284-330, :348-363): its columns of every
        # generator Linear stay off the matrix cores.  Forward: the GEMM covers the dense input columns
        # and its epilogue adds the hot columns of W from WcT[l] = W_l[:, cond columns]^T by the row ids in
        # GenContext.sel.  Backward: the cond columns of dW_l are row sums of dZ_l scattered by the same
        # ids into dWcT[l] (zeroed with the gradient buffer: one fill), then transposed into place.
        if sparse_cond is None:
            sparse_cond = StepEngine.SPARSE_COND
        self.sparse_cond = self.nd > 0 and (True if sparse_cond is None else bool(sparse_cond))
        outs = self.gdim + [self.Dd]
        if self.sparse_cond:
            self.WcT = [torch.zeros(self.C, (w + 3) // 4 * 4, device=d)[:, :w] for w in outs]     # rows 16-byte aligned
        # BatchNorm's column sums are taken in the Linear's GEMM epilogue (fp64 atomics into [2, F] accumulators:
        # GemmArgs::colstat) unless the engine runs in reproducible mode (separate fixed-order reduction)
        self.fused_stats = not self.deterministic
        # their accumulators are SLOTS x [2, F]: tile row t adds into slot t % SLOTS (128 tile rows meeting at one fp64
        # address cost a product 4-5 us), and the normalisation kernels sum the slots like row-chunk partials
        self.stat_slots = int(os.environ.get('DPCGAN_STAT_SLOTS', '4')) if self.fused_stats else 1
        # BatchNorm backward's sums come out of the epilogue of the product that completes the block's output gradient -
        # for blocks up to this width (measured: 0.875 -> 0.852 ms per step at width 256, but 5.21 -> 5.37 ms at width
        # 1024, where the epilogue's extra y / z tile reads outweigh the two reduction launches they replace)
        self.bnb_max_width = 512
        # Everything the generator update accumulates into lies in ONE buffer that one fill zeroes (_g_front):
        # the flat gradient (split-K products add with beta = 1), the transposed cond-column gradients, the
        # BatchNorm statistics accumulators of the update's forward (and of both forwards of a data-parallel
        # step: `pair` holds [layer][context][2, F] so that one all-reduce per layer serves both).
        spec = [('g', (self.gen.n,), torch.float32)]
        if self.sparse_cond:
            spec += [('dWcT%d' % i, (self.C, w), torch.float32) for i, w in enumerate(outs)]
        S = self.stat_slots
        spec += [('bn%d' % l, (S * 2 * w,), torch.float64) for l, w in enumerate(self.gdim)]
        spec += [('pair%d' % l, (2, S * 2 * w), torch.float64) for l, w in enumerate(self.gdim)]
        spec += [('bnb%d' % l, (S * 2 * w,), torch.float64) for l, w in enumerate(self.gdim)]      # BatchNorm backward sums
        self._gen_gx, gv = _zero_pool(spec, d)
        self.gen.g = gv['g']
        if self.sparse_cond:
            self.dWcT = [gv['dWcT%d' % i] for i in range(len(outs))]
        self._bn_g = [gv['bn%d' % l] for l in range(L)]
        self._bn_pair = [gv['pair%d' % l] for l in range(L)]
        self._bn_b = [gv['bnb%d' % l] for l in range(L)]
        self._wct_stale = True
        ds, k = [], self.K0
        for i, w in enumerate(self.ddim):
            ds += [('seq.%d.weight' % (3 * i), (w, k)), ('seq.%d.bias' % (3 * i), (w,))]
            k = w
        ds += [('seq.%d.weight' % (3 * M), (1, k)), ('seq.%d.bias' % (3 * M), (1,))]
        self.disc = FlatParams(ds, d)
        self.bn_run = {}
        for l in range(L):
            self.bn_run['seq.%d.bn.running_mean' % l] = torch.zeros(self.gdim[l], device=d)
            self.bn_run['seq.%d.bn.running_var' % l] = torch.ones(self.gdim[l], device=d)
        self.bn_batches = 0
        self._init_params(seed)
        # ---- activations / workspaces ----------------------------------------------------------
        B, n = self.B, self.n
        f32 = dict(device=d, dtype=torch.float32)
        self.dHg = torch.zeros(B, (self.Wtot + 3) // 4 * 4, **f32)[:, :self.Wtot]
        self.dZ = [torch.zeros(B, w, **f32) for w in self.gdim]
        ldd = (self.Dd + 3) // 4 * 4          # rows padded to 16 bytes: keeps the GEMM's vector path
        self.dfake = torch.zeros(B, ldd, **f32)[:, :self.Dd]
        # Gradient penalty through the first layer's Gram matrix W1 W1^T (see _d_back): on by default
        # when the batch of pac groups is at least half as tall as the first hidden layer is wide
        # (the d0 x d0 x K0 Gram product then costs less than the three n x d0 x K0 products it replaces)
        if gp_gram is None:
            gp_gram = StepEngine.GP_GRAM
        self.gp_gram = bool(gp_gram) if gp_gram is not None else (2 * n >= self.ddim[0])
        # The discriminator's input is cat([data, cond]) per sub-row (dp_cgan.py:622-623) and cond is two-hot:
        # with the Gram-form penalty no K0-wide operand other than the input itself is left, so the cond
        # columns (45 % of K0 at T1M) leave the matrix cores here too.  Needs the padded data segment to fit
        # the stored one (P <= Dd + C) and the atomics of the sorted row sums (not in reproducible mode).
        self.P = (self.Dd + 31) // 32 * 32
        self.sparse_d = bool(self.sparse_cond and self.gp_gram and self.P <= self.row and not self.deterministic)
        self.seg = (self.Dd, self.P, self.row)              # segment map of W1's data columns (ops.gemm_desc)
        xrow = self.P if self.sparse_d else self.row
        self.K0x = pac * xrow                               # length of a stored discriminator-input row
        # _d_back: the penalty's (u^T diag u) W1 behind the grouped weight gradients instead of beside them (wide first layers)
        ps = os.environ.get('DPCGAN_PEN_SERIAL', 'auto')
        self.pen_serial = (self.K0 * self.ddim[0] >= (8 << 20)) if ps == 'auto' else ps == '1'
        self.Xd = torch.zeros((2 if self.sparse_d else 3) * B, xrow, **f32)
        self.Xd3 = self.Xd.view(-1, self.K0x)
        self.Xg = torch.zeros(B, xrow, **f32)               # discriminator input of the generator update
        self.cd = GenContext(self, self.Xd[0:B])            # forward of the discriminator update
        self.cg = GenContext(self, self.Xg)                 # forward of the generator update (its backward reads it)
        self._last = self.cd
        self.bn_scratch = [torch.zeros(2, w, **f32) for w in self.gdim]
        self.dX = torch.zeros(B, xrow, **f32)
        self.H = [torch.zeros(3 * n, w, **f32) for w in self.ddim]
        self.S = [torch.zeros(3 * n, w, **f32) for w in self.ddim]
        self.Dl = [torch.zeros(3 * n, w, **f32) for w in self.ddim]
        self.y = torch.zeros(3 * n, **f32)          # discriminator outputs of the latest D update (fake | real | xhat)
        self.y_g = torch.zeros(n, **f32)            # ... of the latest G update
        self.y_x = torch.zeros(n, **f32)            # ... of score_real_rows
        rs = torch.empty(3 * n, **f32)
        rs[:n] = 1.0 / self.gn
        rs[n:2 * n] = -1.0 / self.gn
        rs[2 * n:] = 1.0
        self.rowscale_d = rs
        self.rowscale_g = torch.full((n,), -1.0 / self.gn, **f32)
        self.ones_col = torch.ones(2 * n, 1, **f32)
        self.pen_rows = torch.zeros(n, device=d, dtype=torch.float64)
        self._losses = torch.zeros(4, **f32)         # loss_d, pen, -mean(y_fake), cond loss (see `losses`)
        self._loss_dirty = [False, False]            # D part / G part to be reduced from y, pen_rows, part_loss
        self._ce_parts = 0
        maxF = sum(self.gdim) + sum(self.ddim) + self.Dd + 2      # (one launch pair sums several matrices: ops.colsum_multi)
        self.part = torch.zeros(64 * 2 * maxF, device=d, dtype=torch.float64)
        # Independent kernels of a step (the real-row sampling chain beside the generator forward, the
        # weight-gradient GEMMs beside the input-gradient chain) run on side streams; inside a captured
        # graph that becomes parallel branches, which fill CUs that one small GEMM leaves idle.  Each
        # side stream has its own reduction workspace.
        self.side = [torch.cuda.Stream(device=d) for _ in range(2)] if self.dev.type == 'cuda' else []
        # the generator update's forward (inputs, generator, activation) depends on nothing the
        # discriminator update writes: train_step runs it on this stream beside the last discriminator update
        self.early = torch.cuda.Stream(device=d) if self.dev.type == 'cuda' else None
        self.part_side = [torch.zeros(64 * 2 * maxF, device=d, dtype=torch.float64) for _ in self.side]
        self.pair_part = torch.zeros(2 * 64 * 2 * max(self.gdim + [1]), device=d, dtype=torch.float64)
        self.pair_fold = torch.zeros(4 * max(self.gdim + [1]), device=d, dtype=torch.float64)
        self._open = []          # side streams with work the main stream has not joined yet
        self._pending = []       # side branches recorded but not yet issued (_Fork)
        # which fork sites issue their branch AFTER the main stream's next launch (_Fork; one bit per site: 0 real rows /
        # W1 W1^T, 1 the early lane, 2-3 the discriminator's weight-gradient branches, 4 hot-id sorts, 5 the output
        # layer's weight gradient, 6 the Residual blocks' weight gradients, 7 running statistics).  Measured one by one
        # at T1M (0.679 ms with none): 0 -9 us, 4 -5, 6 -4, 2 -4, 3 -3, 7 -2, but 1 +56 and 5 +23 (the lane then shares
        # a queue with the branches it should run beside); 0|4|6 together 0.664 ms, every larger set 0.664-0.667.
        self.defer_forks = int(os.environ.get('DPCGAN_DEFER_FORKS', '0x51'), 0)
        self._defer_ok = True    # (off while a step runs with injected draws: those bodies hold torch copies)
        self._g1_issued = False  # W1 W1^T of this discriminator update is already on its way (_draw_inputs)
        self._pair_sort = False  # _gen_forward_pair issues the hot-id sort at the end of its second lane
        self._pair_R = {}        # row chunks of the paired forwards' column reductions, per layer
        self._early_open = False # the early stream holds work of _gen_forward_pair the main stream has not joined
        self._so = 0             # Philox call-site offset of the update being issued (64 * its index in the step)
        self.concurrent = True
        self.group_dweights = os.environ.get('DPCGAN_GROUP_DW', '1') != '0'     # _d_back: the weight gradients as one grouped launch
        self.seed_in_epilogue = os.environ.get('DPCGAN_SEED_EPI', '1') != '0'   # _disc_forward: the chain seed out of the last hidden layer's product
        self.pair_forwards = True    # data-parallel train_step: one BatchNorm all-reduce per layer for both forwards
        self.part_loss = torch.zeros(2048, device=d, dtype=torch.float64)
        self.perm = torch.zeros(B, device=d, dtype=torch.int64)
        self.keys_ws = torch.zeros(2 * B, device=d, dtype=torch.int64)
        self.row_id = torch.zeros(B, device=d, dtype=torch.int64)
        self.hook_add = torch.zeros(len(self.disc.keys), **f32)
        # both backward passes (penalty, loss_d) fire every hook except the last bias's, which only
        # loss_d reaches [observed on the reference: 11 draws per hook set for 6 tensors]
        self.hook_fires = torch.tensor([2.0] * (len(self.disc.keys) - 1) + [1.0], **f32)
        d0 = self.ddim[0]
        # What the discriminator update accumulates into, zeroed by one fill at its start (_draw_inputs): the flat
        # gradient, the transposed cond-column gradient of the first layer, the Gram-form penalty's small
        # products (split-K adds with beta = 1, no memset nodes) and the BatchNorm accumulators of its forward.
        ldw = (d0 + 3) // 4 * 4
        spec = [('g', (self.disc.n,), torch.float32), ('P1', (d0, d0), torch.float32), ('G1', (d0, d0), torch.float32)]
        if self.sparse_d:
            spec.append(('dW1cT', (pac * self.C, ldw), torch.float32))
        spec += [('bn%d' % l, (self.stat_slots * 2 * w,), torch.float64) for l, w in enumerate(self.gdim)]
        self._disc_gx, dv = _zero_pool(spec, d)
        self.disc.g = dv['g']
        self._bn_d = [dv['bn%d' % l] for l in range(L)]
        self.cd.bnacc, self.cg.bnacc = self._bn_d, self._bn_g
        if self.sparse_d:
            self.W1cT = torch.zeros(pac * self.C, ldw, **f32)[:, :d0]       # W1[:, cond columns]^T, sub-row major
            self.dW1cT = dv['dW1cT'][:, :d0]                                # its gradient (sorted row sums)
            self.sel_real = torch.full((B, 2), -1, device=d, dtype=torch.int32)   # hot ids that travel with the real rows
            self.order_d = torch.zeros(4 * B, device=d, dtype=torch.int32)  # (row, id) pairs of both blocks sorted by id
            self.sid_d = torch.zeros(4 * B, device=d, dtype=torch.int32)
            self.n_items_d = torch.zeros(1, device=d, dtype=torch.int32)
        self.Zp = torch.zeros(2 * n, d0, **f32)        # first layer's plain products of the fake and real rows
        self.G1 = dv['G1']                             # W1 W1^T
        self.Tq = torch.zeros(n, d0, **f32)            # u G1
        self.P1 = dv['P1']                             # u^T diag(f c) u
        self.gp_f = torch.zeros(n, **f32)
        self.gp_fc = torch.zeros(n, **f32)
        self.gp_q = torch.zeros(n, device=d, dtype=torch.float64)
        self.norm2 = torch.zeros(n, device=d, dtype=torch.float64)      # DP-SGD: per-example squared gradient norms
        self.norms = torch.zeros(n, **f32)
        self.rowc = torch.ones(3 * n, **f32)                              # clip factor of each stacked row
        self.rowc_w = torch.zeros(3 * n, **f32)                           # ... times rowscale_d
        # per-rank streams for the batch shard (z, gumbel, dropout, sampling); one stream shared by all
        # ranks for the DP noise, so that it is added exactly once and replicas stay bit-identical
        self.rng = ops.new_rng(seed + 7919 * self.dist.rank, d)
        self.rng_shared = ops.new_rng(seed ^ 0x3c3c3c3c, d)
        self.sampler = None      # DeviceSampler, attached by the caller

    # ------------------------------------------------------------------------------------------
    def _init_params(self, seed):
        gen = torch.Generator().manual_seed(seed)
        for l in range(self.L):
            w = torch.empty(self.gen.shapes['seq.%d.fc.weight' % l])
            b = torch.empty(self.gen.shapes['seq.%d.fc.bias' % l])
            _linear_init(w, b, gen)
            self.gen.P('seq.%d.fc.weight' % l).copy_(w)
            self.gen.P('seq.%d.fc.bias' % l).copy_(b)
            self.gen.P('seq.%d.bn.weight' % l).fill_(1.0)
            self.gen.P('seq.%d.bn.bias' % l).zero_()
        w = torch.empty(self.gen.shapes['seq.%d.weight' % self.L])
        b = torch.empty(self.gen.shapes['seq.%d.bias' % self.L])
        _linear_init(w, b, gen)
        self.gen.P('seq.%d.weight' % self.L).copy_(w)
        self.gen.P('seq.%d.bias' % self.L).copy_(b)
        for i in range(self.M + 1):
            w = torch.empty(self.disc.shapes['seq.%d.weight' % (3 * i)])
            b = torch.empty(self.disc.shapes['seq.%d.bias' % (3 * i)])
            _linear_init(w, b, gen)
            self.disc.P('seq.%d.weight' % (3 * i)).copy_(w)
            self.disc.P('seq.%d.bias' % (3 * i)).copy_(b)

    def _mm(self, A, B, C, **kw):
        """A GEMM of this engine: split-K (fp32 atomics) only when the engine is not in reproducible mode."""
        kw.setdefault('splitk', 1 if self.deterministic else 0)
        return ops.gemm(A, B, C, **kw)

    def _mm_gather(self, A, B, C, gat, gidx, **kw):
        kw.setdefault('splitk', 1 if self.deterministic else 0)
        return ops.gemm_gather(A, B, C, gat, gidx, **kw)

    # buffers of the latest update's generator forward (the tests and the samplers' checks read them)
    Hg = property(lambda self: self._last.Hg)
    fake = property(lambda self: self._last.fake)
    fake_act = property(lambda self: self._last.X[:, :self.Dd])
    cond = property(lambda self: self._last.cond)
    mask = property(lambda self: self._last.mask)
    sel = property(lambda self: self._last.sel)
    gcount = property(lambda self: self._last.gcount)
    pair_id = property(lambda self: self._last.pair_id)
    val_idx = property(lambda self: self._last.val_idx)

    @property
    def losses(self):
        """[loss_d, penalty, -mean(y_fake), cond loss]: this rank's additive shares for the latest
        updates.  The step itself never needs these scalars (the reference reads them once per epoch,
        dp_cgan.py:735-736), so their final reductions run here, on demand, from the buffers the
        latest updates left behind (y, pen_rows, cond-loss block partials)."""
        n = self.n
        if self._loss_dirty[0]:
            ops.mean_pair(self.y, n, n / self.gn, -n / self.gn, self._losses[0:1])
            ops.sum_f64(self.pen_rows, GP_LAMBDA / self.gn, self._losses[1:2])
            self._loss_dirty[0] = False
        if self._loss_dirty[1]:
            ops.mean_pair(self.y_g, n, -n / self.gn, 0.0, self._losses[2:3])
            if self.has_cond:
                ops.sum_f64(self.part_loss[:self._ce_parts], 2.0 / (float(self.gB) * self.nd), self._losses[3:4])
            else:
                self._losses[3:4].zero_()
            self._loss_dirty[1] = False
        return self._losses

    def global_losses(self):
        """[loss_d, penalty, -mean(y_fake), cond loss] of the latest updates over the GLOBAL batch
        (every entry of ``losses`` is this rank's additive share)."""
        t = self.losses.clone()
        self.dist.allreduce(t)
        return t.cpu().numpy()

    def generator_state_dict(self):
        sd = self.gen.state_dict()
        for k, v in self.bn_run.items():
            sd[k] = v.detach().clone()
        for l in range(self.L):
            sd['seq.%d.bn.num_batches_tracked' % l] = torch.tensor(self.bn_batches, dtype=torch.int64)
        return sd

    def discriminator_state_dict(self):
        return self.disc.state_dict()

    def load_generator_state_dict(self, sd):
        """Accepts the reference Generator's state_dict (keys/shapes of SURVEY 3.5)."""
        self.gen.load(sd)
        self._wct_stale = True
        for k in self.bn_run:
            if k in sd:
                self.bn_run[k].copy_(torch.as_tensor(sd[k]).to(self.dev, torch.float32))
        k = 'seq.0.bn.num_batches_tracked'
        if k in sd:
            self.bn_batches = int(sd[k])

    # ------------------------------------------------------------------------------------------
    # pieces
    # ------------------------------------------------------------------------------------------
    def _zc(self, c):
        z0 = self.start[0]
        return c.Hg[:, z0:z0 + self.E], c.Hg[:, z0 + self.E:]

    def _draw_inputs(self, inj, need_real, c, zero=None):
        """z, cond vector (+ mask) and, for the discriminator update, the real rows.  ``zero``: the accumulator
        pool of the update these inputs belong to, cleared here (the discriminator update's own pool when
        ``need_real``)."""
        zv, cv = self._zc(c)
        B, Dd, C = self.B, self.Dd, self.C
        if need_real:
            zero = self._disc_gx       # the discriminator update's accumulators (gradients, Gram products, BatchNorm sums)
        if self.has_cond and not any(k in inj for k in ('z', 'cond', 'pairs', 'r')):
            # pool fill, latent draw and cond vectors are independent of each other: ONE launch (three nodes in a
            # row at the top of the step's critical path otherwise) with the same draws as the separate calls below
            s = self.sampler
            fake_c = None if self.sparse_d else c.X[:, Dd:]
            # dense copies of the cond vector only where something reads them: the cond columns of Hg when the
            # generator's Linears multiply them (not sparse_cond), the discriminator input's when its first Linear
            # does (not sparse_d), the matrix itself for the generator update's cond loss - a discriminator update
            # with both sparse forms needs none (config 5: 2 x 590 MB per update, 385 -> ~100 us for this launch)
            dsts = []
            c._cond_stale = False
            if not need_real or not (self.sparse_cond and self.sparse_d) or not self.lazy_cond:
                dsts.append(c._cond)
            else:
                c._cond_stale = True
            # (the reproducible mode's generator backward takes the DENSE weight-gradient product over all of Hg,
            # cond columns included: _gen_dweight)
            if not self.sparse_cond or not self.lazy_cond or (not need_real and self.deterministic):
                dsts.append(cv)
            if fake_c is not None:
                dsts.append(fake_c)
            ops.step_inputs(s.cnt, s.cond_st, s.dims, s.N, B, dsts, None if c._cond_stale else c._mask,
                            c.pair_id, c.val_idx, c.sel, c.gcount, self.rng, ops.SITE_CONDVEC + self._so, z=zv,
                            site_z=ops.SITE_Z + self._so, zero=zero)
            self._join()
            if need_real:
                self._draw_real(inj, need_real, c)
            return
        if zero is not None:
            zero.zero_()
        # the latent draw stays on the main stream, in front of the cond-vector sampling: forked beside it,
        # the fork / join pair cost more than the 5 us kernel it hid (0.958 -> 0.950 ms per step)
        if 'z' in inj:
            zv.copy_(inj['z'])
        else:
            ops.fill(zv, ops.FILL_NORMAL, self.rng, ops.SITE_Z + self._so, 0.0, 1.0)
        if not self.has_cond:
            self._join()
            if need_real:     # uniform rows (data_sampler.py:388-390)
                real = self.Xd[B:2 * B]
                if 'real' in inj:
                    real.copy_(inj['real'])
                else:
                    ops.row_uniform(self.sampler.N, self.row_id, u_inj=inj.get('u_row'), rng=self.rng,
                                    site=ops.SITE_ROWPICK + self._so)
                    ops.row_gather(self.sampler.table, self.row_id, real, Dd)
            return
        fake_c = None if self.sparse_d else c.X[:, Dd:]     # (sparse_d: the cond columns travel as c.sel)
        c._cond_stale = False
        if 'cond' in inj:
            c.cond.copy_(inj['cond'])
            c.mask.copy_(inj['mask'])
            cv.copy_(c.cond)
            if fake_c is not None:
                fake_c.copy_(c.cond)
            if self.sparse_cond:
                ops.cond_to_sel(c.cond, c.sel)
        else:
            s = self.sampler
            ops.condvec_pair_sample(s.cnt, s.cond_st, s.dims, s.N, B, [c.cond, cv] + ([] if fake_c is None else [fake_c]),
                                    c.mask, c.pair_id,
                                    c.val_idx, c.sel, c.gcount, pairs_inj=inj.get('pairs'),
                                    r_inj=inj.get('r'), rng=self.rng, site=ops.SITE_CONDVEC + self._so)
        self._join()
        if need_real:
            self._draw_real(inj, need_real, c)

    def _draw_real(self, inj, need_real, c):
        """Real rows that match the cond vectors of ``c`` (data_sampler.py:382-396) on a side branch.
        (Measured in round 3: the branch held back until the generator's output layer, so that config 5's 630 MB row
        gather does not run beside the forward's small kernels: 82.0 -> 82.3 ms there, 0.647 -> 0.670 ms at T1M.)"""
        B, Dd, C = self.B, self.Dd, self.C
        real = self.Xd[B:2 * B]
        with self._fork(0, 0):      # beside the generator forward; joined before the interpolation
            if 'real' in inj:
                real[:, :Dd].copy_(inj['real'])
                real_cond = inj['real_cond'] if 'real_cond' in inj else c.cond[inj['perm']]
                if self.sparse_d:
                    ops.cond_to_sel(real_cond.contiguous(), self.sel_real)
                else:
                    real[:, Dd:].copy_(real_cond)
            else:
                s = self.sampler
                if 'perm' in inj:
                    self.perm.copy_(inj['perm'])
                else:
                    ops.random_perm(self.rng, self.perm, self.keys_ws, site=ops.SITE_PERM + self._so)
                ops.row_select(s.bitmaps, c.sel, c.gcount, self.row_id, perm=self.perm, k_inj=inj.get('k'),
                               rng=self.rng, cntblk=s.cntblk, site=ops.SITE_ROWPICK + self._so)
                if self.sparse_d:
                    ops.row_gather(s.table, self.row_id, real, Dd, sel=c.sel, perm=self.perm, C=0, sel_out=self.sel_real)
                else:
                    ops.row_gather(s.table, self.row_id, real, Dd, sel=c.sel, perm=self.perm, C=C)
            if self.sparse_d and need_real != 'rows':
                # (row, cond id) pairs of the fake and real blocks sorted by id: the cond columns of the first
                # layer's weight gradient are row sums in that order (_d_back)
                ops.hot_sort(c.sel, self.C, self.order_d, self.sid_d, self.n_items_d, sel_b=self.sel_real, pac=self.pac)
                self._probe('real_rows_branch')
            if self.gp_gram:
                # W1 W1^T of the Gram-form penalty (see _d_back): issued here, beside the generator forward's
                # chain of small kernels, rather than beside the discriminator's first-layer product (there: 0.655 ->
                # 0.676 ms per step, round 3)
                W1 = self.disc.P('seq.0.weight')
                self._mm(W1, W1, self.G1, trans_b=True, beta=1.0)
                self._g1_issued = True

    def _gen_weight(self, l):
        return self.gen.P('seq.%d.fc.weight' % l if l < self.L else 'seq.%d.weight' % self.L)

    def _refresh_wct(self):
        """WcT[l] = W_l[:, cond columns]^T for every generator Linear, one launch."""
        if self.sparse_cond:
            ops.transpose_slices([(self._gen_weight(l)[:, -self.C:], self.WcT[l]) for l in range(self.L + 1)])
        self._wct_stale = False

    def _gen_linear(self, c, l, out, bias, colstat=None):
        """out = Hg[:, start[l]:] W_l^T + bias (layer L = the output Linear); ``colstat``: fp64 [stat_slots, 2, F]
        accumulator that receives the column sums / sums of squares of ``out`` in the same launch (BatchNorm statistics)."""
        W = self._gen_weight(l)
        st = dict(colstat=colstat, stat_slots=self.stat_slots) if colstat is not None else {}
        if not self.sparse_cond:
            self._mm(c.Hg[:, self.start[l]:], W, out, trans_b=True, bias=bias, **st)
            return
        if self._wct_stale or self.gen.loaded:       # parameters were (re)loaded outside an optimiser step
            self.gen.loaded = False
            self._refresh_wct()
        self._mm_gather(c.Hg[:, self.start[l]:self.Wtot - self.C], W[:, :W.shape[1] - self.C], out, self.WcT[l], c.sel,
                        bias=bias, **st)

    def _gen_dweight(self, c, l, dy, what=3):
        """dW_l (+)= dy^T Hg[:, start[l]:] into the zeroed gradient buffer (``what`` & 1: the product over the
        dense input columns; & 2: the cond columns' row sums).  Bias gradients: _gen_bias_grads."""
        gw = self.gen.G('seq.%d.fc.weight' % l if l < self.L else 'seq.%d.weight' % self.L)
        if self.sparse_cond and not self.deterministic:
            # The output layer's branch starts beside the input-gradient product that the whole backward chain waits
            # for: its small latency-bound kernel (the row sums) goes first, so the big product meets that chain
            # ~15 us later (0.802 -> 0.793 ms per step; a delay probe there used to make the step FASTER).  For the
            # other layers the product first is better (0.782 against 0.787).
            first = l == self.L
            if what & 2 and first:
                ops.hot_rows_sum(dy, c.order, c.sid, c.n_items, self.dWcT[l])
            if what & 1:
                self._mm(dy, c.Hg[:, self.start[l]:self.Wtot - self.C], gw[:, :gw.shape[1] - self.C], trans_a=True, beta=1.0)
            if what & 2 and not first:
                ops.hot_rows_sum(dy, c.order, c.sid, c.n_items, self.dWcT[l])
        elif what & 1:
            self._mm(dy, c.Hg[:, self.start[l]:], gw, trans_a=True, beta=1.0)

    def _gen_bias_grads(self, part):
        """Bias gradients of every generator Linear (column sums of dfake and the dZ_l) in one launch pair."""
        L = self.L
        probs = [(self.dZ[l], None, self.gen.G('seq.%d.fc.bias' % l)) for l in range(L)]
        probs.append((self.dfake, None, self.gen.G('seq.%d.bias' % L)))
        adv = None
        if self._adv_pending is not None:
            # the step's closing bookkeeping rides in this launch (train_step): the last launch but the Adam step's,
            # and nothing beside or behind it draws numbers or reads the counters except that Adam step (t_host = 0)
            adv = ops.step_advance(self.rng, rng_b=self.rng_shared, t_a=self.disc.t, add_a=self._adv_pending,
                                   t_b=self.gen.t, add_b=1)
            self._adv_pending, self._adv_done = None, True
        ops.colsum_multi(probs, part, atomic=not self.deterministic, advance=adv)      # (into the zeroed gradient pool)

    def _bn_sums(self, c, l, acc):
        """BatchNorm column sums of c.Z[l] for the fused normalisation kernel: -> (workspace, row chunks).  With
        fused statistics the Linear's epilogue has left the finished [2, F] sums in ``acc`` already."""
        if self.fused_stats:
            return acc, self.stat_slots
        R = ops.colreduce_partial(0, c.Z[l], c.part[l])
        return c.part[l], R

    def _gen_forward(self, c, running=True):
        """Generator forward into context ``c``.  ``running`` False leaves BatchNorm's running
        statistics alone (``_bn_running`` applies the update later, from the column sums kept for it)."""
        g = self.gen
        count = self.bn_count
        for l in range(self.L):
            self._gen_linear(c, l, c.Z[l], g.P('seq.%d.fc.bias' % l), colstat=c.bnacc[l] if self.fused_stats else None)
            part, R = self._bn_sums(c, l, c.bnacc[l])
            R = self._sync_sums(part, R, 2 * self.gdim[l])
            c.sums[l] = (part, R)
            ops.bn_relu_fwd_fused(c.Z[l], c.Hg[:, self.start[l + 1]:self.start[l]], part, R, count,
                                  g.P('seq.%d.bn.weight' % l), g.P('seq.%d.bn.bias' % l), c.bn_mean[l], c.bn_invstd[l],
                                  run_mean=self.bn_run['seq.%d.bn.running_mean' % l] if running else None,
                                  run_var=self.bn_run['seq.%d.bn.running_var' % l] if running else None)
        if running:
            self.bn_batches += 1
        self._gen_linear(c, self.L, c.fake, g.P('seq.%d.bias' % self.L))

    def _gen_forward_pair(self, ca, cb, so_a, so_b):
        """Two generator forwards in lockstep (the last discriminator update's and the generator
        update's: same weights, different inputs) for data-parallel steps, activations included: the
        BatchNorm column sums of both lie in ONE buffer ([context][2, F], finished sums) and take ONE
        all-reduce per layer instead of two.
        Same arithmetic and the same running-statistics order (ca then cb) as two _gen_forward calls.
        The two are independent chains of small kernels, so cb's runs on the `early` stream beside ca's
        (joined at every collective); cb's last Linear and activation are left running there when this
        returns - ``_join_early`` (called by the next collective and before the generator update's
        backward) brings them back."""
        g = self.gen
        count = self.bn_count
        main = torch.cuda.current_stream(self.dev)
        lanes = self.concurrent and self.early is not None
        if self.sparse_cond and (self._wct_stale or self.gen.loaded):
            self.gen.loaded = False
            self._refresh_wct()

        class lane_b(object):          # ``with lane_b():`` issues on the early stream, after what that stream holds
            def __enter__(s_):
                s_.ctx = torch.cuda.stream(self.early) if lanes else None
                if s_.ctx is not None:
                    s_.ctx.__enter__()

            def __exit__(s_, *a):
                if s_.ctx is not None:
                    s_.ctx.__exit__(*a)
                return False
        if lanes:
            self.early.wait_stream(main)            # whatever of cb's inputs was drawn on the main stream
        for l in range(self.L):
            F = self.gdim[l]
            b = g.P('seq.%d.fc.bias' % l)
            if self.fused_stats:
                acc = self._bn_pair[l]              # [2 contexts][slots, 2F] sums (zeroed with the gradient pool)
                R, width = self.stat_slots, self.stat_slots * 2 * F
                sums = [acc[0], acc[1]]
                fold = self.pair_fold[:4 * F]
            else:
                R = ops.colreduce_rows(self.B, F)
                width = R * 2 * F
                sums = [self.pair_part[0:width], self.pair_part[width:2 * width]]
                fold = self.pair_fold[:4 * F]
            with lane_b():
                self._gen_linear(cb, l, cb.Z[l], b, colstat=sums[1] if self.fused_stats else None)
                if not self.fused_stats:
                    assert ops.colreduce_partial(0, cb.Z[l], sums[1]) == R
            self._gen_linear(ca, l, ca.Z[l], b, colstat=sums[0] if self.fused_stats else None)
            if not self.fused_stats:
                assert ops.colreduce_partial(0, ca.Z[l], sums[0]) == R
            if lanes:
                main.wait_stream(self.early)
            # both contexts' slots / row-chunk partials folded to [2][2F] finished sums: one small all-reduce
            src = self._bn_pair[l].view(-1) if self.fused_stats else self.pair_part
            if R > 1:
                ops.colreduce_fold(src, R, 2 * F, nsets=2, set_stride=width, out=fold, out_stride=2 * F)
                self._allreduce(fold)
                sums, R = [fold[0:2 * F], fold[2 * F:4 * F]], 1
            else:
                self._allreduce(src[:2 * width])
            for i, c in enumerate((ca, cb)):
                if i == 1 and lanes:
                    self.early.wait_stream(main)     # after ca's: the running statistics are updated in sequence order
                with (lane_b() if i == 1 else contextlib.nullcontext()):
                    ops.bn_relu_fwd_fused(c.Z[l], c.Hg[:, self.start[l + 1]:self.start[l]], sums[i], R, count,
                                          g.P('seq.%d.bn.weight' % l), g.P('seq.%d.bn.bias' % l), c.bn_mean[l],
                                          c.bn_invstd[l], run_mean=self.bn_run['seq.%d.bn.running_mean' % l],
                                          run_var=self.bn_run['seq.%d.bn.running_var' % l])
        self.bn_batches += 2
        with lane_b():
            self._so = so_b
            self._gen_linear(cb, self.L, cb.fake, g.P('seq.%d.bias' % self.L))
            self._activate_loss({}, cb)
            if self._pair_sort:          # the backward's hot-id sort at the lane's end, where nothing waits for it
                ops.hot_sort(cb.sel, self.C, cb.order, cb.sid, cb.n_items)
        self._so = so_a
        self._gen_linear(ca, self.L, ca.fake, g.P('seq.%d.bias' % self.L))
        self._activate({}, ca, raw=True)
        self._early_open = lanes

    def _join_early(self):
        if self._early_open:
            torch.cuda.current_stream(self.dev).wait_stream(self.early)
            self._early_open = False

    def _bn_running(self, c):
        """The running-statistics update a ``_gen_forward(c, running=False)`` left out, in its place in
        the sequence of forwards (momentum updates do not commute)."""
        count = self.bn_count
        for l in range(self.L):
            part, R = c.sums[l]
            ops.colreduce_finish(part, R, 2, self.gdim[l], 0, self.bn_scratch[l][0], self.bn_scratch[l][1],
                                 count=count, run_mean=self.bn_run['seq.%d.bn.running_mean' % l],
                                 run_var=self.bn_run['seq.%d.bn.running_var' % l])
        self.bn_batches += 1

    def _activate(self, inj, c, raw=False):
        """Activated generator output -> the discriminator-input rows of ``c``.  ``raw``: a table
        without discrete columns feeds the RAW generator output to the discriminator in the
        discriminator update (dp_cgan.py:624-626, Q5) - but the activated one in the generator update
        (:699-702)."""
        if self.has_cond or not raw:
            ops.apply_activate_fwd(c.fake, c.X[:, :self.Dd], self.spans, exp_noise=inj.get('exp'), rng=self.rng,
                                   site=ops.SITE_GUMBEL + self._so)
        else:
            ops.apply_activate_fwd(c.fake, self.dfake, self.spans, exp_noise=inj.get('exp'), rng=self.rng,
                                   site=ops.SITE_GUMBEL + self._so)      # computed and dropped, as the reference does
            c.X.copy_(c.fake)

    def _w1ct(self):
        """W1cT = W1[:, cond columns]^T, sub-row major [pac*C, d0]: refreshed after every discriminator Adam step
        (_disc_update) and here when the parameters were loaded from outside."""
        if self.disc.loaded:
            self._refresh_w1ct()
        return self.W1cT

    def _refresh_w1ct(self):
        W1, C, Dd = self.disc.P('seq.0.weight'), self.C, self.Dd
        ops.transpose_slices([(W1[:, s * self.row + Dd:(s + 1) * self.row], self.W1cT[s * C:(s + 1) * C])
                              for s in range(self.pac)])
        self.disc.loaded = False

    def _disc_first(self, x, zp, rows, H, S, keep, sel, **act):
        """First hidden layer of `rows` input rows x [rows, K0x] -> H, S: the K0-long product goes split-K into
        the all-zero zp (no memset node: the pass below, or the per-example Gram kernel, clears it again), then
        bias + LeakyReLU + Dropout - and, with the cond columns off the matrix cores (sparse_d), the gather of
        W1's hot cond columns - in one pass."""
        d = self.disc
        W1 = d.P('seq.0.weight')
        if self.sparse_d:
            self._mm(x, W1, zp, trans_b=True, beta=1.0, seg_b=1, seg=self.seg)
            act.update(gat=self._w1ct(), pac=self.pac, C=self.C)
            act.update(sel if isinstance(sel, dict) else dict(sel_a=sel))
        else:
            self._mm(x, W1, zp, trans_b=True, beta=1.0)
        ops.disc_first_layer_act(zp, d.P('seq.0.bias'), rows, H, S, keep=keep, rng=self.rng,
                                 site_keep=ops.SITE_DROP0 + self._so, leak=LEAK, **act)

    def _disc_forward(self, rows, inj, row0=0, y=None, rowscale=None, X3=None, sel=None):
        """Discriminator forward over rows [row0, row0+rows) of the stacked input (or of ``X3``, whose
        results then land in rows [row0, row0+rows) of the layer buffers; ``sel``: the hot cond ids of those
        rows' sub-rows, sparse_d); with ``rowscale`` the head kernel also seeds the backward chain:
        Dl[M-1] = rowscale * w_last * S[M-1]."""
        d, M = self.disc, self.M
        rs = slice(row0, row0 + rows)
        x = self.Xd3[rs] if X3 is None else X3
        keeps = inj.get('keep')
        first = 0
        if (X3 is not None or self.sparse_d) and rows <= 2 * self.n:
            self._disc_first(x, self.Zp[0:rows], rows, self.H[0][rs], self.S[0][rs], keeps[0] if keeps else None, sel,
                             stacked=False, clear=True)
            x = self.H[0][rs]
            first = 1
        w_head = d.P('seq.%d.weight' % (3 * M))
        seeded = rowscale is not None and first < M and self.seed_in_epilogue
        for i in range(first, M):
            # the LAST hidden layer's product also writes the seed of the backward chain, rowscale * w_head * slope: the
            # chain then follows it directly and the head kernel (y alone: only the loss scalars read it) leaves the
            # critical path for a branch
            seed = (self.Dl[i][rs], rowscale[0:rows], w_head.view(-1)) if seeded and i == M - 1 else None
            self._mm(x, d.P('seq.%d.weight' % (3 * i)), self.H[i][rs], trans_b=True, bias=d.P('seq.%d.bias' % (3 * i)),
                     epi=ops.EPI_LEAKY_DROPOUT, aux_out=self.S[i][rs], keep=keeps[i] if keeps else None,
                     rng=self.rng, site=ops.SITE_DROP0 + i + self._so, leak=LEAK, seed=seed)
            x = self.H[i][rs]
        y = self.y[rs] if y is None else y
        if rowscale is None:
            ops.disc_out_fwd(x, w_head, d.P('seq.%d.bias' % (3 * M)), y)
        elif seeded:
            self._defer_head(x, y, rows)
        else:
            ops.disc_out_fwd(x, w_head, d.P('seq.%d.bias' % (3 * M)), y, rowscale=rowscale,
                             slope=self.S[M - 1][rs], delta=self.Dl[M - 1][rs])

    def _disc_forward_stacked(self, inj):
        """Discriminator forward of the discriminator update over the 3n stacked rows [fake | real |
        interpolates] (dp_cgan.py:629-630, :47-53) WITHOUT forming the interpolates: the first Linear's
        products are taken for the fake and real rows only, and those of xhat = a*real + (1-a)*fake follow
        by linearity inside the activation pass; the other layers run over all 3n rows as usual."""
        d, M, n = self.disc, self.M, self.n
        # Zp is all zero here (allocated so; its last reader - the activation pass, or the per-example Gram
        # kernel in DP-SGD mode - clears what it reads): the K0-long product
        # runs split-K with its atomics adding straight into it, no memset node in front
        keeps = inj.get('keep')
        self._disc_first(self.Xd3[0:2 * n], self.Zp, n, self.H[0], self.S[0], keeps[0] if keeps else None,
                         dict(sel_a=self.cd.sel, sel_b=self.sel_real) if self.sparse_d else None,
                         alpha_inj=inj.get('alpha'), site_alpha=ops.SITE_ALPHA + self._so, clear=self.dp_mode != 'dpsgd')
        x = self.H[0]
        w_head = d.P('seq.%d.weight' % (3 * M))
        seeded = M > 1 and self.seed_in_epilogue
        for i in range(1, M):
            seed = (self.Dl[i], self.rowscale_d, w_head.view(-1)) if seeded and i == M - 1 else None      # (see _disc_forward)
            self._mm(x, d.P('seq.%d.weight' % (3 * i)), self.H[i], trans_b=True, bias=d.P('seq.%d.bias' % (3 * i)),
                     epi=ops.EPI_LEAKY_DROPOUT, aux_out=self.S[i], keep=keeps[i] if keeps else None,
                     rng=self.rng, site=ops.SITE_DROP0 + i + self._so, leak=LEAK, seed=seed)
            x = self.H[i]
        if seeded:
            self._defer_head(x, self.y, 3 * self.n)
        else:
            ops.disc_out_fwd(x, w_head, d.P('seq.%d.bias' % (3 * M)), self.y,
                             rowscale=self.rowscale_d, slope=self.S[M - 1], delta=self.Dl[M - 1])

    def _defer_head(self, x, y, rows):
        """Leave the head y = h w_head + b of ``rows`` rows to the next side branch (_flush_head).  In the discriminator
        update's stacked pass (3n rows) the penalty's v-chain later overwrites the interpolates' rows of h, and nothing
        but a diagnostic reads their scores: they are computed - at once, on the main stream - only when
        ``want_interp_scores`` is set (the golden-trace tests compare them), otherwise y[2n:3n] is left as it was."""
        n = self.n
        if rows == 3 * n:
            if self.want_interp_scores:
                M = self.M
                ops.disc_out_fwd(x[2 * n:3 * n], self.disc.P('seq.%d.weight' % (3 * M)), self.disc.P('seq.%d.bias' % (3 * M)),
                                 y[2 * n:3 * n])
            x, y = x[0:2 * n], y[0:2 * n]
        self._head_pending = (x, y)

    def _flush_head(self):
        """The discriminator head y = h w_head + b of the latest forward, left out of the main chain by the seed epilogue
        (_disc_forward): issued where the caller has a side branch open anyway - a fork of its own cost more than the
        kernel (0.695 -> 0.737 ms per step with its own fork / join pair)."""
        if self._head_pending is not None:
            x, y = self._head_pending
            self._head_pending = None
            M = self.M
            ops.disc_out_fwd(x, self.disc.P('seq.%d.weight' % (3 * M)), self.disc.P('seq.%d.bias' % (3 * M)), y)

    def score_real_rows(self, inj=None, rows=None):
        """Discriminator output [rows / pac, 1] on freshly sampled real rows with their cond vectors
        (what the reference's xai_discriminator evaluates for len(data_samples) rows, dp_cgan.py:896-928;
        Dropout on).  Rows are drawn a batch at a time; ``rows`` defaults to one batch."""
        inj = inj or {}
        rows = self.B if rows is None else int(rows)
        if rows % self.pac != 0:
            raise AssertionError('rows %% pac != 0')            # dp_cgan.py:67
        out = []
        need = rows // self.pac
        while need > 0:
            self._so = 0
            self._last = self.cd
            self._draw_inputs(inj, need_real='rows', c=self.cd)
            self._join()
            self._disc_forward(self.n, inj, row0=self.n, y=self.y_x, sel=self.sel_real if self.sparse_d else None)
            ops.state_advance(self.rng)
            out.append(self.y_x[:min(need, self.n)].detach().cpu())
            need -= self.n
        return torch.cat(out).view(-1, 1)

    def _disc_chain_down(self, rows, rowscale):
        """D_M = rowscale * w_last * S_M;  D_{i-1} = (D_i W_i) * S_{i-1}."""
        d, M = self.disc, self.M       # (the seed D_M was written by the head kernel of _disc_forward)
        for i in range(M - 1, 0, -1):
            self._mm(self.Dl[i][0:rows], d.P('seq.%d.weight' % (3 * i)), self.Dl[i - 1][0:rows], epi=ops.EPI_MUL,
                     aux_in=self.S[i - 1][0:rows])

    class _Fork(object):
        """``with eng._fork(k):`` runs the block on side stream k after everything issued so far on
        the main stream; ``eng._join()`` makes the main stream wait for all open side streams.

        Launch order (``eng.defer_forks``): the block's launches are RECORDED while the block runs and issued right
        after the main stream's next launch (``_after_call``), behind an event taken here.  Dependencies are the same;
        what changes is the order in which the graph capture sees the two successors of the fork point.  The HIP graph
        runtime cuts a graph into chains by a depth-first walk along each node's FIRST successor and gives every
        chain a hardware queue: with the branch launched first the branch inherits the main stream's chain and the
        main stream's next kernel starts a new one - on another queue, behind a cross-queue dependency (~10 us in
        the round-3 traces against ~0 between two kernels of one queue), at every fork of the step."""

        def __init__(self, eng, k, site=0):
            self.eng, self.k, self.site = eng, k, site

        def __enter__(self):
            e = self.eng
            self.rec = None
            if not e.concurrent or not e.side or e._inline:
                self.ctx = None
                return e.part
            side = e.side[self.k]
            main = torch.cuda.current_stream(e.dev)
            L = ops.lib()
            if (e.defer_forks >> self.site) & 1 and e._defer_ok and L.defer is None:
                self.ev = torch.cuda.Event()
                self.ev.record(main)
                self.main = main.cuda_stream
                self.rec = L.defer = []
            else:
                e._flush_pending()
                side.wait_stream(main)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
            if side not in e._open:
                e._open.append(side)
            return e.part_side[self.k]

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)
            if self.rec is not None:
                e, L = self.eng, ops.lib()
                L.defer = None
                if a[0] is None:
                    e._pending.append((self.main, e.side[self.k], self.ev, self.rec))
                    L.after = e._after_call
            return False

    def _after_call(self, stream):
        """A launch has been issued: if it went to the stream a pending branch forked from, the branch goes now."""
        h = getattr(stream, 'value', stream) or 0      # (c_void_p(0).value is None: the default stream)
        if any(p[0] == h for p in self._pending):
            self._flush_pending()

    def _flush_pending(self):
        if not self._pending:
            return
        L = ops.lib()
        L.after = None
        pending, self._pending = self._pending, []
        for _, side, ev, rec in pending:
            if callable(rec):           # the early lane (train_step)
                rec(ev)
                continue
            side.wait_event(ev)
            for name, args in rec:
                L.call(name, *args)

    def _fork(self, k, site=0):
        return StepEngine._Fork(self, k, site)

    def _probe(self, label):
        """Critical-path probe (tuning aid, off unless DPCGAN_DELAY=<label>:<cycles> is set): a spin node of that many
        timer cycles (40000 ~ 20 us) at the labelled point of the step, on whatever stream is current.  A step
        that grows by the delay has that point on its critical path; one that does not has slack there; one
        that gets FASTER has a branch starting too early beside a product the chain waits for
        (tools/critical_probe.sh)."""
        spec = os.environ.get('DPCGAN_DELAY', '')
        if spec and spec.split(':')[0] == label:
            torch.cuda._sleep(int(spec.split(':')[1]))

    def _join(self):
        if self._inline:         # issuing the early branch: it forks nothing, and the open forks belong to the main stream
            return
        self._flush_pending()
        main = torch.cuda.current_stream(self.dev)
        for side in self._open:
            main.wait_stream(side)
        self._open = []

    def _allreduce(self, t):
        if not self.dist.active:
            return t
        if self._ar_fault is not None:     # test hook (GraphStep.INJECT_CAPTURE_FAULT): the next collective raises once
            fault, self._ar_fault = self._ar_fault, None
            raise fault
        if self._ar_hook is not None:      # graph capture in progress, the collective runs between graph segments:
            self._join()                   # a graph cut needs every branch back on the main stream
            self._join_early()
            self._ar_hook(t)
            return t
        # In the graph (or eager) the collective is one more node of the main stream: what it reduces was produced on
        # the main stream (or joined into it by the caller), and the branches that are open - weight-gradient products,
        # the other lane's activation - keep running beside it.  (Joining them here put every weight-gradient product
        # of the generator's backward in front of the next BatchNorm all-reduce: 0.916 -> 0.839 ms forced onto one rank.)
        return self.dist.allreduce(t)

    def _sync_sums(self, part, R, width):
        """Data parallel: the R row-chunk partials of a BatchNorm column reduction (``part`` [R, width] fp64) become
        the sums over the GLOBAL batch - folded on the device to one vector, then ONE all-reduce of `width` doubles
        (not R x width).  Returns the chunk count to hand to the consumer (1 once folded)."""
        if not self.bn_dist:
            return R
        if R > 1:
            ops.colreduce_fold(part, R, width)
        self._allreduce(part[:width])
        return 1

    def _colsum(self, x, out, w=None, accumulate=False, part=None):
        part = self.part if part is None else part
        R = ops.colreduce_partial(1, x, part, w=w)
        ops.colreduce_finish(part, R, 1, x.shape[1], 1, out, accumulate=accumulate)

    # ------------------------------------------------------------------------------------------
    # discriminator update
    # ------------------------------------------------------------------------------------------
    def d_step(self, inj=None, update=True, sub=None):
        """One discriminator update.  Stand-alone (``sub`` None) it ends by advancing the RNG / Adam
        step counters itself; inside ``train_step`` it is update number ``sub`` of the step, draws
        from call sites offset by 64*sub and leaves the single end-of-step advance to the caller."""
        inj = inj or {}
        self._defer_ok = not inj
        self._so = 0 if sub is None else 64 * sub
        self._probe('d_inputs')
        self._draw_inputs(inj, need_real=True, c=self.cd)
        self._gen_forward(self.cd)
        self._probe('g_forward_done')
        self._activate(inj, self.cd, raw=True)
        self._d_back(inj, update, sub)

    def _d_back(self, inj, update, sub):
        """Everything of a discriminator update after its generator forward + activation."""
        self._defer_ok = not inj
        self._so = 0 if sub is None else 64 * sub
        self._k = 0 if sub is None else sub
        n, B, M, d = self.n, self.B, self.M, self.disc
        self._last = self.cd
        self._join()
        if self._early_hook is not None and self.early_at == 'd_forward':      # see train_step
            self._early_hook()
        W1 = d.P('seq.0.weight')
        gp_rows = slice(2 * n, 3 * n)
        gram = self.gp_gram
        coef = 2.0 * GP_LAMBDA / self.gn
        if gram:
            # Gram-form penalty.  With G1 = W1 W1^T (d0 x d0) and u = the penalty rows of Dl[0]:
            #   ||grad_xhat D||^2 = ||u W1||^2 = rowsum((u G1) * u)       v1 = f * (u G1) * S_1
            #   penalty part of dW1 = u^T v0 = (u^T diag(f) u) W1
            # so neither xhat, g0 = u W1 nor v0 (each [n, K0]) is formed, and the K0-wide products
            # shrink from 3n to 2n rows (forward, weight gradient) or disappear (g0, v1).
            if not self._g1_issued:
                with self._fork(0, 0):
                    self._mm(W1, W1, self.G1, trans_b=True, beta=1.0)
            self._g1_issued = False
            self._disc_forward_stacked(inj)
        else:
            ops.interpolate(self.Xd[B:2 * B], self.Xd[0:B], self.Xd[2 * B:3 * B], n, self.K0, alpha_inj=inj.get('alpha'),
                            rng=self.rng, site=ops.SITE_ALPHA + self._so)
            self._disc_forward(3 * n, inj, rowscale=self.rowscale_d)
        self._loss_dirty[0] = True
        self._probe('d_forward_done')
        if self._early_hook is not None and self.early_at not in ('d_chain_done', 'd_grads_done'):      # see train_step
            self._early_hook()
        # one downward chain for all 3n rows: first-order deltas (fake, real) and the penalty's u-chain
        self._disc_chain_down(3 * n, self.rowscale_d)
        # penalty v-chain first, so that every layer's (delta, input) pair exists before the DP clip
        # factors: v_i = (v_{i-1} W_i^T) * S_i over the penalty rows, stored where H_i's xhat rows were
        if gram:
            self._join()
            self._mm(self.Dl[0][gp_rows], self.G1, self.Tq)
            ops.gp_gram_rows(self.Tq, self.Dl[0][gp_rows], self.S[0][gp_rows], coef, self.pen_rows, self.gp_f,
                             self.H[0][gp_rows], q_out=self.gp_q)
        else:
            self._mm(self.Dl[0][gp_rows], W1, self.Xd3[gp_rows])                    # g0 = u1 W1 over the xhat rows
            ops.gp_norm(self.Xd3[gp_rows], coef, self.pen_rows)                     # -> v0 in place
            self._mm(self.Xd3[gp_rows], W1, self.H[0][gp_rows], trans_b=True, epi=ops.EPI_MUL, aux_in=self.S[0][gp_rows])
        for i in range(1, M):
            self._mm(self.H[i - 1][gp_rows], d.P('seq.%d.weight' % (3 * i)), self.H[i][gp_rows], trans_b=True,
                     epi=ops.EPI_MUL, aux_in=self.S[i][gp_rows])
        kscale, w_last, w_bias, pen_scale = None, self.rowscale_d, None, self.gp_f
        if self.dp_mode == 'dpsgd':
            # per-example gradient norms (ghost norm over the 3 rows of each pac group) of every layer and
            # the clip factors, one launch
            layers, prev = [], self.Xd3
            for i in range(M):
                layers.append((self.Dl[i], prev))
                prev = self.H[i]
            layers.append((self.rowscale_d.view(-1, 1), prev))
            if gram:
                ops.persample_clip_all(layers, n, self.dp_clip, float(self.gn), self.norm2, self.rowc,
                                       roww=self.rowscale_d, rowc_w=self.rowc_w, norm_out=self.norms, zpre=self.Zp,
                                       gp_f=self.gp_f, gp_q=self.gp_q, fc_out=self.gp_fc, clear_zpre=True,
                                       sel_a=self.cd.sel if self.sparse_d else None,
                                       sel_b=self.sel_real if self.sparse_d else None, pac=self.pac)
                pen_scale = self.gp_fc
            else:
                ops.persample_clip_all(layers, n, self.dp_clip, float(self.gn), self.norm2, self.rowc,
                                       roww=self.rowscale_d, rowc_w=self.rowc_w, norm_out=self.norms)
            kscale, w_last, w_bias = self.rowc, self.rowc_w, self.rowc[0:2 * n]
        # Weight gradients.  Every product ADDS into the gradient pool that the update's first node zeroed
        # (beta = 1: split-K needs no memset node); the first layer's product (K0 wide) runs on the main stream,
        # everything else beside it: the other layers' products, and ONE launch pair for all bias gradients and
        # the head layer's weight gradient (weighted column sums).
        self._probe('d_weight_grads')
        if self._early_hook is not None and self.early_at == 'd_chain_done':
            self._early_hook()
        if gram and self.group_dweights and not self.deterministic:
            # ONE launch for the independent weight-gradient products of the update (every layer's dy^T x and the
            # penalty's u^T diag(f c) u: all m-/n-contiguous operands, one kernel variant - ops.gemm_group), the bias
            # sums and the cond columns' sorted row sums on a branch beside it, then the one product that depends on
            # the group (P1 W1 into dW1).  Before: three graph branches of 1 + 2 + (M-1) launches whose fixed costs
            # and fork / join pairs the step paid per branch (phase 78 -> ~60 us in the T1M trace).
            with self._fork(1, 2) as part:
                probs = [(self.Dl[i][0:2 * n], w_bias, d.G('seq.%d.bias' % (3 * i))) for i in range(M)]
                probs.append((self.H[M - 1], w_last, d.G('seq.%d.weight' % (3 * M)).view(-1)))
                probs.append((self.ones_col, w_last[0:2 * n], d.G('seq.%d.bias' % (3 * M))))
                for i in range(0, len(probs), 8):
                    ops.colsum_multi(probs[i:i + 8], part, atomic=True)
                if self.sparse_d:
                    ops.hot_rows_sum(self.Dl[0][0:2 * n], self.order_d, self.sid_d, self.n_items_d, self.dW1cT,
                                     rowscale=kscale[0:2 * n] if kscale is not None else None)
                self._flush_head()
            u = self.Dl[0][gp_rows]
            pen_serial = self.pen_serial
            with self._fork(0, 3):
                # the penalty's share of dW1, (u^T diag(f c) u) W1: two small dependent products beside the group (the
                # d0 x d0 one over n rows has 16 tiles - cut along K so that it brings 128 blocks, not 16); it ADDS to
                # dW1 while the group's first product does: atomics on both sides
                self._mm(u, u, self.P1, trans_a=True, beta=1.0, kscale=pen_scale, tile=64, splitk=max(1, min(8, n // 128)))
                if not pen_serial:
                    self._mm(self.P1, W1, d.G('seq.0.weight'), beta=1.0, atomic=True)
            ks2 = kscale[0:2 * n] if kscale is not None else None
            seg_kw = dict(seg_c=1, seg=self.seg) if self.sparse_d else {}
            descs = [ops.gemm_desc(self.Dl[0][0:2 * n], self.Xd3[0:2 * n], d.G('seq.0.weight'), trans_a=True, beta=1.0,
                                   kscale=ks2, splitk=0, tile=64, atomic=not pen_serial, **seg_kw)]
            prev = self.H[0]
            for i in range(1, M):
                descs.append(ops.gemm_desc(self.Dl[i], prev, d.G('seq.%d.weight' % (3 * i)), trans_a=True, beta=1.0,
                                           kscale=kscale, splitk=0, tile=64))
                prev = self.H[i]
            ops.gemm_group(descs)
            self._join()
            if pen_serial:
                # wide first layers: (u^T diag(f c) u) W1 BEHIND the group, plain read-modify-write instead of atomics on
                # both sides - W1's gradient is K0 x d0 = 76 M elements at config 5 and the atomic epilogue of this
                # K = d0 product ran at 22 TF/s (1.76 ms beside the group's 1.41 ms)
                self._mm(self.P1, W1, d.G('seq.0.weight'), beta=1.0)
            self._d_grads_tail(update, inj, sub)
            return
        with self._fork(1, 2) as part:
            probs = [(self.Dl[i][0:2 * n], w_bias, d.G('seq.%d.bias' % (3 * i))) for i in range(M)]
            probs.append((self.H[M - 1], w_last, d.G('seq.%d.weight' % (3 * M)).view(-1)))
            probs.append((self.ones_col, w_last[0:2 * n], d.G('seq.%d.bias' % (3 * M))))
            for i in range(0, len(probs), 8):
                ops.colsum_multi(probs[i:i + 8], part, atomic=not self.deterministic)
            if self.sparse_d:
                # cond columns of dW1: sums of the (clip-scaled) delta rows by hot id, transposed into place below
                ops.hot_rows_sum(self.Dl[0][0:2 * n], self.order_d, self.sid_d, self.n_items_d, self.dW1cT,
                                 rowscale=kscale[0:2 * n] if kscale is not None else None)
            self._flush_head()
            # the other layers' products behind the small kernels of this branch: branch 0 carries the penalty's two
            # products, the main stream the K0-wide one - three lanes of about equal length (0.795 -> 0.786 ms per
            # step against all products on branch 0)
            prev = self.H[0]
            for i in range(1, M):
                self._mm(self.Dl[i], prev, d.G('seq.%d.weight' % (3 * i)), trans_a=True, beta=1.0, kscale=kscale)
                prev = self.H[i]
        if gram:
            # penalty part (u^T diag(f c) u) W1 beside the fake and real rows' product: both ADD into the zeroed
            # gradient (the penalty part with atomics too - the two run at the same time)
            u = self.Dl[0][gp_rows]
            with self._fork(0, 3):
                self._mm(u, u, self.P1, trans_a=True, beta=1.0, kscale=pen_scale)
                self._mm(self.P1, W1, d.G('seq.0.weight'), beta=1.0, atomic=not self.deterministic)
            ks2 = kscale[0:2 * n] if kscale is not None else None
            if self.deterministic:
                self._join()
            at = not self.deterministic      # (both products add to the same gradient at the same time: atomics on both sides)
            if self.sparse_d:       # data columns: product over the compact input into W1's layout
                self._mm(self.Dl[0][0:2 * n], self.Xd3[0:2 * n], d.G('seq.0.weight'), trans_a=True, beta=1.0, kscale=ks2,
                         seg_c=1, seg=self.seg, atomic=at)
            else:
                self._mm(self.Dl[0][0:2 * n], self.Xd3[0:2 * n], d.G('seq.0.weight'), trans_a=True, beta=1.0, kscale=ks2,
                         atomic=at)
        else:
            # loss + penalty weight gradient in one product (rows scaled by their clip factor in DP-SGD mode)
            self._mm(self.Dl[0], self.Xd3, d.G('seq.0.weight'), trans_a=True, beta=1.0, kscale=kscale)
        self._join()
        self._d_grads_tail(update, inj, sub)

    def _d_grads_tail(self, update, inj, sub):
        """What follows the weight gradients of a discriminator update: the cond columns' share, the Adam step."""
        d = self.disc
        self._flush_head()       # (no-op when a branch has taken it)
        # The cond columns' sorted row sums join the flat gradient inside the Adam launch (_disc_update) - unless no
        # update follows, or the flat gradient is all-reduced first: then one transpose launch adds them here
        fused = update and self._adam_slices_d()
        if self.sparse_d and (not fused or self.dist.active):
            C, Dd, gw = self.C, self.Dd, d.G('seq.0.weight')
            ops.transpose_slices([(self.dW1cT[s * C:(s + 1) * C], gw[:, s * self.row + Dd:(s + 1) * self.row])
                                  for s in range(self.pac)], accumulate=True)
        self._probe('d_grads_done')
        if self._early_hook is not None:          # (early_at == 'd_grads_done': beside the Adam launch only)
            self._early_hook()
        if update:
            self._disc_update(inj, fused)
        if sub is None:
            ops.state_advance(self.rng, rng_b=self.rng_shared if update else None, t_a=self.disc.t if update else None)

    def _adam_slices_d(self):
        """The discriminator's Adam launch also moves the cond slices of W1 (gradient addend in, W1cT out)."""
        return self.sparse_d and self.pac <= 16

    def _adam_slices_g(self):
        return self.sparse_cond and self.L + 1 <= 8

    def _disc_update(self, inj, fused=False):
        d = self.disc
        self._allreduce(d.g)
        seg_add, seg_off, noise_scale, noise_inj = None, None, 0.0, None
        if self.dp_mode == 'reference':
            # reference behaviour (dp_cgan.py:638-647): one scalar N(0,1)/B per parameter
            # tensor and hook firing; after t discriminator steps each tensor carries t+1 hook sets.
            if 'hook_noise' in inj:
                self.hook_add.copy_(inj['hook_noise'])
            else:
                ops.hook_noise(self.hook_add, self.hook_fires, d.t, 1.0 / self.gB, self.rng_shared,
                               site=ops.SITE_HOOK + self._so, t_add=1 + self._k)
            seg_add, seg_off = self.hook_add, d.seg_off
        elif self.dp_mode == 'dpsgd':
            # the gradient buffer holds (1/n) sum_i clip(g_i); one N(0, (sigma*C)^2) draw per
            # parameter from the stream all ranks share, scaled by the same 1/n
            noise_scale = self.dp_sigma * self.dp_clip / self.gn
            noise_inj = inj.get('dp_noise')
        kw = dict(clamp=self.weight_clip if self.private else 0.0, t_dev=d.t, t_host=1 + self._k, seg_off=seg_off,
                  seg_add=seg_add, noise_scale=noise_scale, noise_inj=noise_inj, rng=self.rng_shared,
                  site=ops.SITE_NOISE_D + self._so)
        if fused:
            # one launch: the cond slices of W1 (pac of them, periodic in the pac-row layout) take their transposed
            # gradient addend and leave their updated values in W1cT - two transpose launches less on the main stream
            W1, C, Dd, row = d.P('seq.0.weight'), self.C, self.Dd, self.row
            # (data parallel: the addend went into the flat gradient before its all-reduce, _d_back; the launch
            # still writes W1cT)
            ops.adam_step_slices(d.p, d.g, d.m, d.v, self.lr_d, self.wd_d, [(W1, row, Dd)],
                                 [(W1, s * row + Dd, C, None if self.dist.active else self.dW1cT[s * C:(s + 1) * C],
                                   self.W1cT[s * C:(s + 1) * C]) for s in range(self.pac)], **kw)
            d.loaded = False
            return
        ops.adam_step(d.p, d.g, d.m, d.v, self.lr_d, self.wd_d, **kw)
        if self.sparse_d:
            self._refresh_w1ct()

    # ------------------------------------------------------------------------------------------
    # generator update
    # ------------------------------------------------------------------------------------------
    def g_step(self, inj=None, update=True, sub=None):
        """One generator update (``sub``: see d_step)."""
        inj = inj or {}
        self._g_front(inj, sub)
        self._g_back(inj, update, sub)

    def _g_front(self, inj, sub, running=True):
        """The part of a generator update that reads nothing a discriminator update writes: zeroed
        gradient buffer, z / cond inputs, generator forward, output activation."""
        self._so = 0 if sub is None else 64 * sub
        self._defer_ok = not inj
        c = self.cg
        # (the pool fill - one for all weight gradients: their split-K products then accumulate with beta = 1 - rides
        # in the launch that draws the inputs)
        self._draw_inputs(inj, need_real=False, c=c, zero=self._gen_gx)
        self._sort_pending = False
        if self.sparse_cond:
            if self._inline:
                # on the early lane a fork is not a branch: the sort would sit in front of the lane's first product,
                # and the lane is as long as the discriminator update it runs beside - _g_back issues it instead
                self._sort_pending = True
            else:
                with self._fork(1, 4):      # beside the forward; the backward's weight gradients read it
                    ops.hot_sort(c.sel, self.C, c.order, c.sid, c.n_items)
        self._gen_forward(c, running=running)
        self._activate_loss(inj, c)

    def _activate_loss(self, inj, c):
        """Output activation of the generator update plus the conditional loss and its gradient
        (dp_cgan.py:697,711) in one pass over the raw generator output: both need only that output and the cond
        vector, so they belong to the front half of the generator update (which runs beside the discriminator
        update).  The loss gradient is WRITTEN to dfake here and the activation's backward adds to it."""
        if not self.has_cond:
            self._activate(inj, c)
            return
        self._ce_parts = ops.activate_cond_loss_fwd(c.fake, c.X[:, :self.Dd], self.spans, c.cond, c.mask, self.col2d,
                                                    self.col2c, self.nd, self.dfake, self.part_loss,
                                                    exp_noise=inj.get('exp'), rng=self.rng,
                                                    site=ops.SITE_GUMBEL + self._so, batch_total=self.gB)

    def _g_back(self, inj, update, sub):
        self._defer_ok = not inj
        self._so = 0 if sub is None else 64 * sub
        self._k = 0
        self._last = c = self.cg
        n, B, M, L, d, g = self.n, self.B, self.M, self.L, self.disc, self.gen
        if getattr(self, '_sort_pending', False):
            self._sort_pending = False
            with self._fork(1, 4):          # joined below, in front of the first weight-gradient branch
                ops.hot_sort(c.sel, self.C, c.order, c.sid, c.n_items)
        self._disc_forward(n, inj, y=self.y_g, rowscale=self.rowscale_g, X3=c.X3, sel=c.sel if self.sparse_d else None)
        self._loss_dirty[1] = True
        self._probe('g_d_forward_done')
        self._disc_chain_down(n, self.rowscale_g)
        # d loss / d D-input (sparse_d: over the data columns only - nothing reads the gradient of the cond columns)
        if self.sparse_d:
            self._mm(self.Dl[0][0:n], d.P('seq.0.weight'), self.dX.view(n, self.K0x), seg_b=2, seg=self.seg)
        else:
            self._mm(self.Dl[0][0:n], d.P('seq.0.weight'), self.dX.view(n, self.K0))
        ops.apply_activate_bwd(c.X[:, :self.Dd], self.dX[:, :self.Dd], self.dfake, self.spans, accumulate=self.has_cond)
        # generator backward: the input-gradient chain (dHg, BatchNorm backward, dX) on the main stream,
        # the weight / bias gradients of each layer beside it on the side streams
        hid = self.start[0]                                    # columns that hold hidden activations
        WL = g.P('seq.%d.weight' % L)

        def fused_bnb(l):
            return self.fused_stats and self.gdim[l] <= self.bnb_max_width

        def bn_back(l):
            """Descriptor keywords that make the product completing block l's output gradient (the first gdim[l]
            columns of its result) also take BatchNorm backward's two column sums (fused statistics only)."""
            if l < 0 or not fused_bnb(l):
                return {}
            blk = slice(self.start[l + 1], self.start[l])
            return dict(colstat=self._bn_b[l], stat_slots=self.stat_slots,
                        bn_back=(c.Hg[:, blk], c.Z[l], c.bn_mean[l], c.bn_invstd[l]), splitk=1)
        if hid > 0:
            self._mm(self.dfake, WL[:, :hid], self.dHg[:, :hid], **bn_back(L - 1))
        self._join()     # branches still open from the front half (the hot-id sort, deferred running statistics)
        # the output layer's weight gradient is forked AFTER the input-gradient product: beside it, the two
        # would share the CUs and the chain that everything else waits for (dHg -> BatchNorm backward -> ..)
        # would take twice as long; from here on it runs beside that chain's small kernels
        self._probe('g_dhg_done')
        # (measured and dropped in round 3: the Residual blocks' Linear-bias gradients as atomics out of the BatchNorm-
        # backward launches and the output layer's column sum on this branch, so that no column-sum launch is left in
        # front of the Adam step - 0.693 -> 0.701 ms: 256 row chunks adding to the same 256 addresses serialise in L2
        # and lengthen the two BatchNorm-backward launches of the critical chain by more than the launch they save)
        gw_group = self.group_gweights and self.sparse_cond and not self.deterministic
        with self._fork(0, 5):
            self._probe('g_out_dweight_branch')
            self._gen_dweight(c, L, self.dfake, what=2 if gw_group else 3)
            self._flush_head()
        count = self.bn_count
        for l in range(L - 1, -1, -1):
            blk = slice(self.start[l + 1], self.start[l])
            dy, yv = self.dHg[:, blk], c.Hg[:, blk]
            if fused_bnb(l):            # the sums came out of the epilogue of the product that completed dy
                sums, R = self._bn_b[l], self.stat_slots
            else:
                sums = self.part
                R = ops.colreduce_partial(2, dy, sums, y=yv, z=c.Z[l], mean=c.bn_mean[l], invstd=c.bn_invstd[l])
            R = self._sync_sums(sums, R, 2 * self.gdim[l])
            dbeta, dgamma = g.G('seq.%d.bn.bias' % l), g.G('seq.%d.bn.weight' % l)
            # BatchNorm gradients come from all-reduced column sums, i.e. are global already: written
            # times 1/world so that the sum over ranks of the flat gradient buffer leaves them unchanged
            # (per-rank statistics: local sums, summed by that all-reduce like every other gradient)
            ops.bn_relu_bwd_fused(dy, yv, c.Z[l], sums, R, count, c.bn_mean[l], c.bn_invstd[l],
                                  g.P('seq.%d.bn.weight' % l), dbeta, dgamma, self.dZ[l],
                                  grad_scale=1.0 / self.dist.world if self.bn_dist else 1.0)
            Wl = g.P('seq.%d.fc.weight' % l)
            w_hid = hid - self.start[l]                         # earlier hidden blocks this layer read
            if w_hid > 0:
                self._mm(self.dZ[l], Wl[:, :w_hid], self.dHg[:, self.start[l]:hid], beta=1.0, **bn_back(l - 1))
            if gw_group:
                # every layer's weight-gradient product in ONE grouped launch behind the chain (ops.gemm_group), the
                # row sums of each layer on a branch as soon as its delta exists, the bias sums beside the group
                # (measured: this last branch issued BEFORE the group, so that its small kernels take their CU slots
                # first - 0.655 -> 0.678 ms: the branch then inherits the main chain's queue; bias sums first: +-0; both on the
                # main stream in front of the group: 0.677)
                with self._fork((L - l) % 2, 6) as part:
                    self._gen_dweight(c, l, self.dZ[l], what=2)
                    if l == 0:
                        self._gen_bias_grads(part)
                if l == 0:
                    dys = [self.dfake] + [self.dZ[k] for k in range(L - 1, -1, -1)]
                    descs = []
                    for k, dyk in zip(range(L, -1, -1), dys):
                        gwk = g.G('seq.%d.fc.weight' % k if k < L else 'seq.%d.weight' % L)
                        descs.append(ops.gemm_desc(dyk, c.Hg[:, self.start[k]:self.Wtot - self.C], gwk[:, :gwk.shape[1] - self.C],
                                                   trans_a=True, beta=1.0, splitk=0, tile=64))
                    ops.gemm_group(descs)
                continue
            if l == 0:
                # the step's last layer: nothing is left for the main stream, so it takes the row sums and the
                # bias gradients of every layer (one launch pair) while the product runs beside it
                with self._fork((L - l) % 2, 6):
                    self._gen_dweight(c, l, self.dZ[l], what=1)
                self._gen_dweight(c, l, self.dZ[l], what=2)
            else:
                with self._fork((L - l) % 2, 6):                   # after the chain's product, as above
                    self._gen_dweight(c, l, self.dZ[l])
        if not gw_group:
            self._gen_bias_grads(self.part)
        self._join()
        self._probe('g_grads_done')
        fused = update and self._adam_slices_g()
        if self.sparse_cond and not self.deterministic and (not fused or self.dist.active):
            # the scattered cond-column gradients into their places in the flat gradient buffer (inside the Adam
            # launch when an update follows and nothing all-reduces the flat gradient first: _gen_update)
            ops.transpose_slices([(self.dWcT[l], self.gen.G('seq.%d.fc.weight' % l if l < L else 'seq.%d.weight' % L)[:, -self.C:])
                                  for l in range(L + 1)])
        if update:
            self._gen_update(fused)
        if sub is None:
            ops.state_advance(self.rng, t_a=self.gen.t if update else None)

    def _gen_update(self, fused=False):
        g = self.gen
        self._allreduce(g.g)
        if fused:
            Ws = [self._gen_weight(l) for l in range(self.L + 1)]
            grads = self.dWcT if not (self.deterministic or self.dist.active) else [None] * (self.L + 1)
            ops.adam_step_slices(g.p, g.g, g.m, g.v, self.lr_g, self.wd_g, [(W, W.shape[1], W.shape[1] - self.C) for W in Ws],
                                 [(W, W.shape[1] - self.C, self.C, grads[l], self.WcT[l]) for l, W in enumerate(Ws)],
                                 t_dev=g.t, t_host=0 if self._adv_done else 1)
            self._wct_stale = False
            return
        ops.adam_step(g.p, g.g, g.m, g.v, self.lr_g, self.wd_g, t_dev=g.t, t_host=0 if self._adv_done else 1)
        self._refresh_wct()

    def train_step(self, d_steps=1):
        """``d_steps`` discriminator updates and one generator update (one iteration of the
        reference's loop, dp_cgan.py:570-722), then ONE launch that advances both RNG step counters
        and both Adam step counters."""
        overlap = self.concurrent and self.early is not None and not self.bn_dist and d_steps >= 1
        if self.bn_dist and self.pair_forwards and d_steps >= 1:
            # data parallel: the two generator forwards of the step's tail share their collectives
            for k in range(d_steps - 1):
                self.d_step(sub=k)
            last = d_steps - 1
            self._so = 64 * last
            self._draw_inputs({}, need_real=True, c=self.cd)
            self._so = 64 * d_steps
            if self.concurrent and self.early is not None:
                # the generator update's inputs on its own lane (the early stream) as well
                self.early.wait_stream(torch.cuda.current_stream(self.dev))
                self._inline = True
                try:
                    with torch.cuda.stream(self.early):
                        self._draw_inputs({}, need_real=False, c=self.cg, zero=self._gen_gx)
                finally:
                    self._inline = False
                self._pair_sort = self.sparse_cond
            else:
                self._draw_inputs({}, need_real=False, c=self.cg, zero=self._gen_gx)
                self._pair_sort = False
                if self.sparse_cond:
                    with self._fork(1, 4):      # joined by the first collective; the generator backward reads it
                        ops.hot_sort(self.cg.sel, self.C, self.cg.order, self.cg.sid, self.cg.n_items)
            self._gen_forward_pair(self.cd, self.cg, 64 * last, 64 * d_steps)
            self._d_back({}, True, last)
            self._join_early()
            self._adv_pending = d_steps if self.advance_in_colsum else None
            self._g_back({}, True, d_steps)
            self._finish_advance(d_steps)
            return
        for k in range(d_steps):
            if overlap and k == d_steps - 1:
                # the generator update's forward runs beside the tail of the last discriminator update
                # (the penalty chain, the clip factors, the weight gradients: many small kernels, while
                # beside the big first-layer GEMMs it would only take CUs from them);
                # its BatchNorm running-statistics update waits for its turn (_bn_running below)
                def issue(ev=None):
                    # forked where the discriminator forward ends.  (How the graph runtime maps branches
                    # to its 4 hardware queues depends on the capture order: issued a few kernels later
                    # the branch lands behind the bias-gradient reductions and overlaps nothing; forked
                    # before the first-layer GEMM it takes CUs from it.  Measured: 1.124 ms here,
                    # 1.137 / 1.219 ms for those two.)
                    self._early_hook = None
                    if ev is None and (self.defer_forks >> 1) & 1:
                        # (launch order as for every fork: the lane is issued after the main stream's next launch,
                        # behind an event taken here - see _Fork)
                        main = torch.cuda.current_stream(self.dev)
                        ev = torch.cuda.Event()
                        ev.record(main)
                        self._pending.append((main.cuda_stream, None, ev, issue))
                        ops.lib().after = self._after_call
                        return
                    so = self._so
                    if ev is None:
                        self.early.wait_stream(torch.cuda.current_stream(self.dev))
                    else:
                        self.early.wait_event(ev)
                    self._inline = True
                    try:
                        with torch.cuda.stream(self.early):
                            self._probe('early_lane_start')
                            self._g_front({}, d_steps, running=False)
                            self._probe('early_lane_end')
                    finally:
                        self._inline = False
                    self._so = so
                    self._early_open = True     # (a graph cut - eager collectives between segments - joins the lane first)
                self._early_hook = issue
            self.d_step(sub=k)
        if overlap:
            self._flush_pending()
            self._join_early()
            with self._fork(1, 7):
                self._bn_running(self.cg)
            # (measured and dropped in round 3: the NEXT step's first launch - its discriminator update's inputs and
            # pool fill, drawn with the step counter + 1 - on this branch, beside the generator update: 0.6505 ->
            # 0.647 ms per step, but the discriminator's gradient buffer and inputs then no longer survive the step
            # that produced them, which callers and tests read)
            self._adv_pending = d_steps if self.advance_in_colsum else None
            self._g_back({}, True, d_steps)
        else:
            self._adv_pending = d_steps if self.advance_in_colsum else None
            self.g_step(sub=d_steps)
        self._finish_advance(d_steps)

    def _finish_advance(self, d_steps):
        """End of a training step: the counters' bump, unless the generator's bias-sum launch has carried it."""
        done, self._adv_done, self._adv_pending = self._adv_done, False, None
        if done:
            return
        # (measured and dropped in round 3: this bookkeeping inside the generator's Adam launch - every block takes a
        # ticket when done and the last one bumps the counters, so the one-thread launch goes away: 0.6846 -> 0.688 ms
        # per step.  The returning atomic at the end of each of ~600 short-lived blocks costs the launch more than the
        # 4 us node it removes.)
        ops.state_advance(self.rng, rng_b=self.rng_shared, t_a=self.disc.t, add_a=d_steps, t_b=self.gen.t, add_b=1)

    # ------------------------------------------------------------------------------------------
    def sample_forward(self, cond=None):
        """Generator forward + activation for DPCGANSynthesizer.sample (BatchNorm in training mode,
        as the reference never calls .eval(): dp_cgan.py:881, Q2).  Returns the activated batch."""
        self._so = 0
        self._last = c = self.cg
        zv, cv = self._zc(c)
        ops.fill(zv, ops.FILL_NORMAL, self.rng, ops.SITE_Z, 0.0, 1.0)
        if self.has_cond:
            cv.copy_(cond)
            if self.sparse_cond:
                ops.cond_to_sel(cv, c.sel)
        self._gen_gx.zero_()            # the forward's BatchNorm accumulators live in the generator's zero pool
        self._gen_forward(c)
        out = torch.empty(self.B, self.Dd, device=self.dev)
        ops.apply_activate_fwd(c.fake, out, self.spans, rng=self.rng)
        ops.state_advance(self.rng)
        return out


class GraphStep(object):
    """One training step (``d_steps`` discriminator updates + one generator update) replayed as HIP
    graphs.  The step issues ~120 small launches; captured, the host pays one graph launch per
    segment instead (MI355X_MICROARCH.md, row graph-replay-floor).  Every kernel reads its RNG step
    and Adam step counters from device memory, so a replay draws fresh numbers.

    Single GPU: one graph.  Data parallel over RCCL: one graph as well, the all-reduces (BatchNorm sums, the two
    flat gradient buffers) captured as nodes of it.  Other backends (gloo rehearsals), DPCGAN_GRAPH_COLLECTIVES=0,
    or a runtime that refuses to capture a collective: the collectives stay eager on the same stream and cut the
    step into graph segments, ``[graph, all-reduce, graph, ...]``.
    """

    INJECT_CAPTURE_FAULT = False     # test hook: the first collective of the in-graph capture raises (exercises the fallback)

    def __init__(self, eng, d_steps=1):
        self.eng, self.d_steps = eng, int(d_steps)
        self.items = None
        self._stream = None

    def _body(self):
        self.eng.train_step(self.d_steps)

    def capture(self):
        """Capture the step.  With RCCL (backend ``nccl``) the all-reduces are captured INTO the graph - one graph
        per step, its branches intact - unless DPCGAN_GRAPH_COLLECTIVES=0 or the capture fails (a runtime that
        refuses a collective on a capturing stream raises there): then, and with any other backend, the
        collectives stay eager between graph segments."""
        eng = self.eng
        in_graph = getattr(eng.dist, 'backend', None) == 'nccl' and os.environ.get('DPCGAN_GRAPH_COLLECTIVES') != '0'
        if in_graph:
            if GraphStep.INJECT_CAPTURE_FAULT:
                eng._ar_fault = RuntimeError('injected: collective refused on a capturing stream')
            try:
                self._capture(True)
                return
            except Exception as ex:             # noqa: BLE001 - whatever the runtime raised, the segment path still works
                import warnings
                warnings.warn('dp_cgans_amd: capturing the RCCL collectives into the step graph failed (%s); '
                              'falling back to eager collectives between graph segments' % (ex,))
                torch.cuda.synchronize(eng.dev)
        self._capture(False)

    def _capture(self, in_graph):
        eng = self.eng
        items = []
        self.collectives_in_graph = in_graph
        self._stream = torch.cuda.Stream(device=eng.dev)
        self._stream.wait_stream(torch.cuda.current_stream(eng.dev))
        batches = eng.bn_batches
        with torch.cuda.stream(self._stream):
            state = {'g': torch.cuda.CUDAGraph()}
            state['g'].capture_begin()

            def cut(t):
                state['g'].capture_end()
                items.append(('graph', state['g']))
                items.append(('allreduce', t))
                state['g'] = torch.cuda.CUDAGraph()
                state['g'].capture_begin()

            eng._ar_hook = None if in_graph else cut
            try:
                self._body()
            except Exception:
                # Leave capture mode before the error travels on.  The step was cut short with branches open (side
                # streams, the early lane): they are joined back into the capturing stream first, so that the capture
                # ENDS CLEANLY - an end with unjoined work invalidates it and can leave those streams unusable for the
                # second capture - and the partial graph is dropped.
                try:
                    eng._inline = False
                    eng._join()
                    eng._join_early()
                    for st in list(eng.side) + ([eng.early] if eng.early is not None else []):
                        torch.cuda.current_stream(eng.dev).wait_stream(st)
                    state['g'].capture_end()
                except Exception:               # noqa: BLE001
                    pass
                eng._open, eng._early_open, eng._early_hook, eng._ar_fault = [], False, None, None
                eng.bn_batches = batches
                raise
            finally:
                eng._ar_hook = None
            state['g'].capture_end()
            items.append(('graph', state['g']))
        torch.cuda.current_stream(eng.dev).wait_stream(self._stream)
        eng.bn_batches = batches              # nothing ran during capture
        self.items = items

    def run(self):
        """Execute one step.  The first call runs eagerly (loads the code objects, sets up the
        communicators) and the second captures; from then on every call is a replay."""
        eng = self.eng
        if self.items is None:
            if self._stream is None:
                self._stream = False
                self._body()
                return
            self.capture()
        for kind, obj in self.items:
            if kind == 'graph':
                obj.replay()
            else:
                eng.dist.allreduce(obj)
        eng.bn_batches += self.d_steps + 1
