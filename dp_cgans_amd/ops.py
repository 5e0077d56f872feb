"""Thin torch-tensor wrappers over the C-ABI (include/dpcgan_hip.h).

Tensors are only used for device memory and the current HIP stream; every computation below runs
in libdpcgan_hip.so.  All functions launch asynchronously on ``torch.cuda.current_stream()``.
"""
import ctypes

import torch

from ._lib import lib

EPI_NONE, EPI_LEAKY_DROPOUT, EPI_MUL, EPI_RELU = 0, 1, 2, 3
FILL_U32, FILL_NORMAL, FILL_EXP, FILL_UNIFORM, FILL_UNIFORM64, FILL_KEEP = 0, 1, 2, 3, 4, 5

# call-site ids of the Philox streams (one independent stream per site and step)
SITE_Z, SITE_GUMBEL, SITE_DROP0, SITE_ALPHA, SITE_CONDVEC, SITE_ROWPICK, SITE_PERM = 1, 2, 16, 3, 4, 5, 6
SITE_NOISE_D, SITE_HOOK, SITE_ENCODE = 7, 8, 9


_dev_index = None        # GPU of the tensors of the call being assembled (see _stream)


def _p(t):
    global _dev_index
    if t is None:
        return None
    if t.is_cuda:
        _dev_index = t.device.index
    return ctypes.c_void_p(t.data_ptr())


def _stream_of(t):
    global _dev_index
    if t.is_cuda:
        _dev_index = t.device.index
    return _stream()


def _stream():
    """The launch stream: the current stream of the GPU that holds the call's tensors.  Every wrapper
    below converts its tensors (``_p``) before it asks for the stream, so a call whose tensors live on
    another GPU than the current one is refused here instead of being launched on the wrong device's
    stream with pointers into another device's memory."""
    idx = _dev_index
    if idx is not None and idx != torch.cuda.current_device():
        raise RuntimeError('dp_cgans_amd.ops: tensors on cuda:%d but the current device is cuda:%d; run under '
                           '`with torch.cuda.device(%d):`' % (idx, torch.cuda.current_device(), idx))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError('%s must live on the GPU' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))


def _ld(t):
    """Leading dimension of a 2-D row-major view (unit stride along columns required)."""
    if t.dim() == 1:
        return t.shape[0]
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError('expected a 2-D tensor with unit column stride, got strides %s' % (t.stride(),))
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


DETERMINISTIC = False


def set_deterministic(on=True):
    """Process-wide DEFAULT for bit-reproducible steps (no split-K atomics in the GEMMs, no scatter atomics), at
    some cost in speed.  It is a Python-side default only: the library keeps no state, every GEMM call carries
    its own ``splitk`` (a StepEngine reads the default once, when it is built)."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)


class GemmDesc(ctypes.Structure):
    """dpc_gemm_desc of include/dpcgan_hip.h."""
    _fields_ = [('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
                ('alpha', ctypes.c_float), ('beta', ctypes.c_float),
                ('A', ctypes.c_void_p), ('a_rs', ctypes.c_int64), ('a_cs', ctypes.c_int64),
                ('B', ctypes.c_void_p), ('b_rs', ctypes.c_int64), ('b_cs', ctypes.c_int64),
                ('C', ctypes.c_void_p), ('ldc', ctypes.c_int64),
                ('bias', ctypes.c_void_p), ('epi', ctypes.c_int32), ('leak', ctypes.c_float),
                ('aux_out', ctypes.c_void_p), ('ld_aux_out', ctypes.c_int64),
                ('aux_in', ctypes.c_void_p), ('ld_aux_in', ctypes.c_int64),
                ('keep', ctypes.c_void_p), ('ld_keep', ctypes.c_int64),
                ('rng', ctypes.c_void_p), ('site', ctypes.c_uint32),
                ('splitk', ctypes.c_int32), ('tile', ctypes.c_int32),
                ('kscale', ctypes.c_void_p),
                ('gat', ctypes.c_void_p), ('ldg', ctypes.c_int64), ('gidx', ctypes.c_void_p),
                ('seg_b', ctypes.c_int32), ('seg_c', ctypes.c_int32), ('seg_len', ctypes.c_int32),
                ('seg_pad', ctypes.c_int32), ('seg_stride', ctypes.c_int64), ('colstat', ctypes.c_void_p),
                ('atomic', ctypes.c_int32), ('bn_y', ctypes.c_void_p), ('ld_bn_y', ctypes.c_int64),
                ('bn_z', ctypes.c_void_p), ('ld_bn_z', ctypes.c_int64), ('bn_mean', ctypes.c_void_p),
                ('bn_invstd', ctypes.c_void_p), ('stat_n', ctypes.c_int32), ('stat_slots', ctypes.c_int32),
                ('seed_out', ctypes.c_void_p), ('ld_seed', ctypes.c_int64), ('seed_row', ctypes.c_void_p),
                ('seed_col', ctypes.c_void_p)]


def _addr(t):
    p = _p(t)
    return None if p is None else p.value


def new_rng(seed, device):
    """Device-resident {seed, step} pair."""
    return torch.tensor([seed & 0x7fffffffffffffff, 0], dtype=torch.int64, device=device)


def state_advance(rng=None, t_a=None, t_b=None, rng_b=None, add_a=1, add_b=1):
    lib().call('dpc_state_advance', _p(rng), _p(rng_b), _p(t_a), int(add_a), _p(t_b), int(add_b), _stream())


def fill(out, kind, rng, site, a=0.0, b=1.0):
    """out (2-D view or 1-D) <- Philox draws of (site, step); element (r, c) uses draw r*cols + c."""
    o2 = out if out.dim() == 2 else out.view(1, -1)
    lib().call('dpc_fill', _p(o2), kind, _ld(o2), o2.shape[0], o2.shape[1], _p(rng), site, a, b, _stream())
    return out


def _mat_strides(t, trans):
    """(row stride, col stride) of op(t) where t is a 2-D view with one unit stride."""
    rs, cs = t.stride(0), t.stride(1)
    if t.shape[0] == 1 and cs == 1:
        rs = max(rs, t.shape[1])
    if t.shape[1] == 1 and rs == 1:
        cs = max(cs, t.shape[0])
    return (cs, rs) if trans else (rs, cs)


def gemm_desc(A, B, C, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, bias=None, epi=EPI_NONE, aux_out=None,
              aux_in=None, keep=None, rng=None, site=0, leak=0.2, splitk=None, kscale=None, tile=0, gat=None, gidx=None,
              seg_b=0, seg_c=0, seg=None, colstat=None, atomic=False, bn_back=None, stat_slots=1, seed=None):
    """dpc_gemm_desc for C = epi(alpha * op(A) @ op(B) + bias [+ gat rows]) + beta * C.
    ``seg`` = (seg_len, seg_pad, seg_stride) with ``seg_b`` (1: along K of a k-contiguous B, 2: along N of an
    n-contiguous B) and / or ``seg_c`` (along N of C): B's (C's) mapped axis is given in COMPACT length by A / C
    (B) - the tensor passed for the mapped operand is the stored one, only its base pointer and leading
    dimension are used."""
    for t, n in ((A, 'A'), (B, 'B'), (C, 'C'), (bias, 'bias'), (aux_out, 'aux_out'), (aux_in, 'aux_in'), (gat, 'gat')):
        _chk(t, torch.float32, n)
    _chk(keep, torch.uint8, 'keep')
    _chk(kscale, torch.float32, 'kscale')
    _chk(gidx, torch.int32, 'gidx')
    a_rs, a_cs = _mat_strides(A, trans_a)
    b_rs, b_cs = _mat_strides(B, trans_b)
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if seg_b == 1:          # B's K axis is the stored (longer) one
        K2 = K
    elif seg_b == 2:        # B's N axis is the stored one; C is compact
        N = C.shape[1]
    if seg_c:               # C's N axis is the stored one; B is compact
        if K != K2 or C.shape[0] != M:
            raise ValueError('gemm shape mismatch (segment-mapped C)')
    elif K != K2 or C.shape[0] != M or C.shape[1] != N:
        raise ValueError('gemm shape mismatch: op(A) %dx%d, op(B) %dx%d, C %s' % (M, K, K2, N, tuple(C.shape)))
    if gat is not None and (gat.shape[1] != N or gidx.shape[0] != M or gidx.shape[1] != 2 or not gidx.is_contiguous()):
        raise ValueError('gemm: gat must be [ids, N], gidx a contiguous [M, 2]')
    if splitk is None:
        splitk = 1 if DETERMINISTIC else 0
    d = GemmDesc()
    d.M, d.N, d.K, d.alpha, d.beta = M, N, K, alpha, beta
    d.A, d.a_rs, d.a_cs = _addr(A), a_rs, a_cs
    d.B, d.b_rs, d.b_cs = _addr(B), b_rs, b_cs
    d.C, d.ldc = _addr(C), _ld(C)
    d.bias, d.epi, d.leak = _addr(bias), epi, leak
    d.aux_out, d.ld_aux_out = _addr(aux_out), _ld(aux_out) if aux_out is not None else 0
    d.aux_in, d.ld_aux_in = _addr(aux_in), _ld(aux_in) if aux_in is not None else 0
    d.keep, d.ld_keep = _addr(keep), _ld(keep) if keep is not None else 0
    d.rng, d.site, d.splitk, d.tile, d.kscale = _addr(rng), site, int(splitk), int(tile), _addr(kscale)
    d.gat, d.ldg, d.gidx = _addr(gat), _ld(gat) if gat is not None else 0, _addr(gidx)
    d.atomic = 1 if atomic else 0
    if seed is not None:        # (seed_out [M, N], rowscale [M], w_head [N]): the backward chain's seed beside the slope
        so, sr, sc = seed
        for t, nm in ((so, 'seed_out'), (sr, 'seed_row'), (sc, 'seed_col')):
            _chk(t, torch.float32, nm)
        if so.shape[0] != M or so.shape[1] != N or sr.numel() != M or sc.numel() != N or not sr.is_contiguous() or not sc.is_contiguous():
            raise ValueError('gemm: seed = (seed_out [M, N], rowscale [M], w_head [N])')
        d.seed_out, d.ld_seed, d.seed_row, d.seed_col = _addr(so), _ld(so), _addr(sr), _addr(sc)
    if colstat is not None:
        _chk(colstat, torch.float64, 'colstat')
        ncs = N
        if bn_back is not None:         # (y, z, mean, invstd): BatchNorm-backward sums over the first y.shape[1] columns
            by, bz, bmean, binv = bn_back
            for t, nm in ((by, 'bn_y'), (bz, 'bn_z'), (bmean, 'bn_mean'), (binv, 'bn_invstd')):
                _chk(t, torch.float32, nm)
            ncs = by.shape[1]
            if by.shape[0] != M or bz.shape != by.shape or bmean.numel() != ncs or binv.numel() != ncs or ncs > N:
                raise ValueError('gemm: bn_back = (y [M, F], z [M, F], mean [F], invstd [F]) with F <= N')
            d.bn_y, d.ld_bn_y, d.bn_z, d.ld_bn_z = _addr(by), _ld(by), _addr(bz), _ld(bz)
            d.bn_mean, d.bn_invstd, d.stat_n = _addr(bmean), _addr(binv), ncs
        if colstat.numel() < 2 * ncs * max(1, int(stat_slots)) or not colstat.is_contiguous():
            raise ValueError('gemm: colstat must be a contiguous fp64 [stat_slots, 2, N]')
        d.colstat = _addr(colstat)
        d.stat_slots = int(stat_slots)
        d.splitk = 1
    if seg is not None:
        d.seg_b, d.seg_c = int(seg_b), int(seg_c)
        d.seg_len, d.seg_pad, d.seg_stride = int(seg[0]), int(seg[1]), int(seg[2])
    d._flops = 2.0 * M * N * K
    # bytes the product must move once: both operands, C (read as well when it is accumulated into), the slope /
    # multiplier matrices of the fused epilogues, the gathered rows
    d._bytes = 4.0 * (M * K + K * N + M * N * (1 + (beta != 0.0) + (aux_out is not None) + (aux_in is not None) + (seed is not None)
                                                + 2 * (gat is not None)))
    d._shape = (M, N, K, bool(trans_a), bool(trans_b), kscale is not None)
    return d


def gemm_run(descs):
    """Issue the products of ``descs`` (list of gemm_desc results) in order with ONE library call."""
    n = len(descs)
    arr = (GemmDesc * n)(*descs)
    L = lib()
    L.call('dpc_gemm_f32_ex', ctypes.byref(arr), n, _stream())
    if L.log is not None:       # (bench.py's launch log: what this launch computes, for the roofline figures)
        L.log[-1] = L.log[-1] + (dict(flops=sum(d._flops for d in descs), bytes=sum(d._bytes for d in descs),
                                       shapes=[d._shape for d in descs]),)


def gemm_group(descs, fill=768):
    """INDEPENDENT products (list of gemm_desc results) in as few launches as their kernels allow (dpc_gemm_f32_group)."""
    n = len(descs)
    # Split-K for the GROUP, not per product: the library's cost model sizes each product's K split as if it had the
    # chip to itself, and the sum then overshoots one round of resident blocks (3 per CU: 768) - the generator's three
    # weight gradients came to 1240 blocks = 1.6 rounds (69.8 us; 59.7 us with 92 tiles x 8 = 736 blocks).  Products
    # that asked for the automatic split get fill // tiles when that is at least 2; larger groups keep the model's.
    tiles = sum(((d.M + 63) // 64) * ((d.N + 63) // 64) for d in descs)
    if tiles * 2 <= fill:
        for d in descs:
            if d.splitk == 0:
                d.splitk = max(1, min(fill // tiles, d.K // 64))
    arr = (GemmDesc * n)(*descs)
    L = lib()
    L.call('dpc_gemm_f32_group', ctypes.byref(arr), n, _stream())
    if L.log is not None:
        L.log[-1] = L.log[-1] + (dict(flops=sum(d._flops for d in descs), bytes=sum(d._bytes for d in descs),
                                       shapes=[d._shape for d in descs]),)


def gemm(A, B, C, **kw):
    """C = epi(alpha * op(A) @ op(B) + bias) + beta * C on the matrix cores (fp32 MFMA); keywords of gemm_desc."""
    gemm_run([gemm_desc(A, B, C, **kw)])
    return C


def gemm_gather(A, B, C, gat, gidx, trans_b=True, bias=None, epi=EPI_NONE, splitk=None, colstat=None, stat_slots=1):
    """C = epi(A @ op(B) + bias + gat[gidx[:,0]] + gat[gidx[:,1]]) (ids < 0 skipped): Linear over the dense
    columns of an input whose trailing cond columns are one-/two-hot; gat = W[:, cond columns]^T."""
    gemm_run([gemm_desc(A, B, C, trans_b=trans_b, bias=bias, epi=epi, gat=gat, gidx=gidx, splitk=splitk, colstat=colstat,
                        stat_slots=stat_slots)])
    return C


def cond_to_sel(cond, sel, bad=None):
    _chk(cond, torch.float32, 'cond')
    _chk(sel, torch.int32, 'sel')
    _chk(bad, torch.int32, 'bad')
    if sel.shape[0] != cond.shape[0] or sel.shape[1] != 2 or not sel.is_contiguous():
        raise ValueError('cond_to_sel: sel must be a contiguous [B, 2] tensor')
    lib().call('dpc_cond_to_sel', _p(cond), _ld(cond), cond.shape[0], cond.shape[1], _p(sel), _p(bad), _stream())
    return sel


def transpose_slices(pairs, accumulate=False):
    """pairs: [(src [r, c] view, dst [c, r] view)]; dst (+)= src^T, 16 pairs per launch."""
    for s_, d_ in pairs:
        _chk(s_, torch.float32, 'src')
        _chk(d_, torch.float32, 'dst')
        if d_.shape[0] != s_.shape[1] or d_.shape[1] != s_.shape[0]:
            raise ValueError('transpose_slices: dst must be the transposed shape of src')
    for i in range(0, len(pairs), 16):
        grp = pairs[i:i + 16]
        L = len(grp)
        P = ctypes.c_void_p * L
        I64 = ctypes.c_int64 * L
        I32 = ctypes.c_int * L
        lib().call('dpc_transpose_slices', L, P(*[s_.data_ptr() for s_, _ in grp]), I64(*[_ld(s_) for s_, _ in grp]),
                   I32(*[s_.shape[0] for s_, _ in grp]), I32(*[s_.shape[1] for s_, _ in grp]),
                   P(*[d_.data_ptr() for _, d_ in grp]), I64(*[_ld(d_) for _, d_ in grp]), int(accumulate), _stream_of(grp[0][0]))


def hot_sort(sel, C, order, sid, n_items, sel_b=None, pac=1):
    """Counting sort of the (row, hot id) pairs of sel [B, 2] (and sel_b) by id -> order, sid, n_items [1] (all
    int32); pac > 1: rows are groups of pac sub-rows, ids (b % pac) * C + sel (see dpc_hot_sort)."""
    for t, n in ((sel, 'sel'), (sel_b, 'sel_b'), (order, 'order'), (sid, 'sid'), (n_items, 'n_items')):
        _chk(t, torch.int32, n)
    B = sel.shape[0]
    need = 2 * B * (2 if sel_b is not None else 1)
    if not sel.is_contiguous() or order.numel() < need or sid.numel() < need:
        raise ValueError('hot_sort: sel must be contiguous [B, 2], order / sid hold 2B entries per id table')
    if sel_b is not None and (sel_b.shape != sel.shape or not sel_b.is_contiguous()):
        raise ValueError('hot_sort: sel_b must match sel')
    lib().call('dpc_hot_sort', _p(sel), _p(sel_b), B, int(pac), int(C), _p(order), _p(sid), _p(n_items), _stream())


def hot_rows_sum(dy, order, sid, n_items, acc, rowscale=None, max_items=None):
    """acc[id] += sum of the rows of dy (times rowscale[row]) listed for id (sorted pairs of hot_sort)."""
    _chk(dy, torch.float32, 'dy')
    _chk(acc, torch.float32, 'acc')
    _chk(rowscale, torch.float32, 'rowscale')
    if acc.shape[1] != dy.shape[1]:
        raise ValueError('hot_rows_sum: shape mismatch')
    lib().call('dpc_hot_rows_sum', _p(dy), _ld(dy), dy.shape[1], _p(order), _p(sid), _p(n_items),
               int(max_items if max_items is not None else order.numel()), _p(acc), _ld(acc), _p(rowscale), _stream())


def apply_activate_fwd(x, y, spans, exp_noise=None, rng=None, site=SITE_GUMBEL, tau=0.2):
    B, Dd = x.shape
    lib().call('dpc_apply_activate_fwd', _p(x), _ld(x), _p(y), _ld(y), _p(spans), spans.shape[0], B, Dd, _p(exp_noise),
               _ld(exp_noise) if exp_noise is not None else 0, _p(rng), site, tau, _stream())
    return y


def apply_activate_bwd(y, dy, dx, spans, tau=0.2, accumulate=False):
    lib().call('dpc_apply_activate_bwd', _p(y), _ld(y), _p(dy), _ld(dy), _p(dx), _ld(dx), _p(spans), spans.shape[0],
               dx.shape[0], dx.shape[1], tau, int(accumulate), _stream())
    return dx


def activate_cond_loss_fwd(x, y, spans, cond, mask, col2d, col2c, nd, dfake, partials, exp_noise=None, rng=None,
                           site=SITE_GUMBEL, tau=0.2, batch_total=0):
    """apply_activate_fwd + cond_loss (loss_out None, accumulate False) in one pass over x; returns how many
    block partials were written."""
    _chk(mask, torch.int32, 'mask')
    _chk(partials, torch.float64, 'partials')
    B, Dd = x.shape
    n_used = ctypes.c_int(0)
    lib().call('dpc_activate_cond_loss_fwd', _p(x), _ld(x), _p(y), _ld(y), _p(spans), spans.shape[0], B, Dd,
               _p(exp_noise), _ld(exp_noise) if exp_noise is not None else 0, _p(rng), site, tau, _p(cond), _ld(cond),
               _p(mask), _ld(mask), _p(col2d), _p(col2c), nd, int(batch_total), _p(dfake),
               _ld(dfake) if dfake is not None else 0, _p(partials), partials.numel(), ctypes.byref(n_used), _stream())
    return n_used.value


def cond_loss(fake, cond, mask, col2d, col2c, nd, loss_out, dfake, partials, accumulate=False, batch_total=0):
    """loss_out None: only the block partials are written; returns how many (sum them * 2/(B*nd) later)."""
    _chk(mask, torch.int32, 'mask')
    _chk(partials, torch.float64, 'partials')
    B, Dd = fake.shape
    n_used = ctypes.c_int(0)
    lib().call('dpc_cond_loss', _p(fake), _ld(fake), _p(cond), _ld(cond), _p(mask), _ld(mask), _p(col2d), _p(col2c), B,
               Dd, nd, _p(loss_out), _p(dfake), _ld(dfake) if dfake is not None else 0, int(accumulate), _p(partials),
               partials.numel(), int(batch_total), ctypes.byref(n_used), _stream())
    return n_used.value


def build_bitmaps(table, cat_col, bitmaps):
    N = table.shape[0]
    C, W = bitmaps.shape
    lib().call('dpc_build_bitmaps', _p(table), _ld(table), N, _p(cat_col), C, _p(bitmaps), W, _stream())
    return bitmaps


def cooccurrence(bitmaps, cnt, cntblk=None):
    C, W = bitmaps.shape
    _chk(cntblk, torch.int32, 'cntblk')
    S = cntblk.shape[2] if cntblk is not None else 1
    lib().call('dpc_cooccurrence', _p(bitmaps), W, C, _p(cnt), _p(cntblk), S, _stream())
    return cnt


def condvec_pair_sample(cnt, cond_st, dims, N, B, cond_dsts, mask, pair_id, val_idx, sel=None, gcount=None,
                        pairs_inj=None, r_inj=None, rng=None, site=SITE_CONDVEC):
    _chk(pairs_inj, torch.int32, 'pairs_inj')
    _chk(r_inj, torch.float64, 'r_inj')
    nd, C = dims.shape[0], cnt.shape[0]
    d = list(cond_dsts) + [None] * (3 - len(cond_dsts))
    args = []
    for t in d:
        args += [_p(t), _ld(t) if t is not None else 0]
    lib().call('dpc_condvec_pair_sample', _p(cnt), _p(cond_st), _p(dims), nd, C, N, _p(pairs_inj), _p(r_inj), _p(rng),
               site, B, *args, _p(mask), _ld(mask) if mask is not None else 0, _p(pair_id), _p(val_idx), _p(sel),
               _p(gcount), _stream())


def step_inputs(cnt, cond_st, dims, N, B, cond_dsts, mask, pair_id, val_idx, sel, gcount, rng, site, z=None,
                site_z=SITE_Z, zero=None):
    """condvec_pair_sample (Philox-driven) + the latent normal fill of ``z`` + the zero fill of the byte buffer
    ``zero`` in one launch (dpc_step_inputs)."""
    nd, C = dims.shape[0], cnt.shape[0]
    d = list(cond_dsts) + [None] * (3 - len(cond_dsts))
    args = []
    for t in d:
        args += [_p(t), _ld(t) if t is not None else 0]
    _chk(z, torch.float32, 'z')
    if zero is not None and (zero.dtype != torch.uint8 or not zero.is_contiguous() or zero.numel() % 16 != 0):
        raise ValueError('step_inputs: zero must be a contiguous uint8 buffer, a multiple of 16 bytes')
    lib().call('dpc_step_inputs', _p(cnt), _p(cond_st), _p(dims), nd, C, N, _p(rng), site, B, *args, _p(mask),
               _ld(mask) if mask is not None else 0, _p(pair_id), _p(val_idx), _p(sel), _p(gcount), _p(z),
               _ld(z) if z is not None else 0, z.shape[0] if z is not None else 0, z.shape[1] if z is not None else 0,
               site_z, _p(zero), zero.numel() if zero is not None else 0, _stream())


def row_select(bitmaps, sel, gcount, row_id, perm=None, k_inj=None, rng=None, site=SITE_ROWPICK, cntblk=None):
    _chk(perm, torch.int64, 'perm')
    _chk(k_inj, torch.int64, 'k_inj')
    _chk(cntblk, torch.int32, 'cntblk')
    C, S = (cntblk.shape[0], cntblk.shape[2]) if cntblk is not None else (0, 1)
    lib().call('dpc_row_select', _p(bitmaps), bitmaps.shape[1], _p(sel), _p(gcount), _p(perm), _p(k_inj), _p(rng), site,
               row_id.shape[0], _p(row_id), _p(cntblk), C, S, _stream())
    return row_id


def row_uniform(N, row_id, u_inj=None, rng=None, site=SITE_ROWPICK):
    _chk(u_inj, torch.float64, 'u_inj')
    lib().call('dpc_row_uniform', N, _p(u_inj), _p(rng), site, row_id.shape[0], _p(row_id), _stream())
    return row_id


def row_gather(table, row_id, out, Dd, sel=None, perm=None, C=0, sel_out=None):
    _chk(sel_out, torch.int32, 'sel_out')
    lib().call('dpc_row_gather', _p(table), _ld(table), _p(row_id), row_id.shape[0], Dd, _p(out), _ld(out), _p(sel),
               _p(perm), C, _p(sel_out), _stream())
    return out


def random_perm(rng, perm, keys_ws, site=SITE_PERM):
    lib().call('dpc_random_perm', _p(rng), site, perm.shape[0], _p(perm), _p(keys_ws), _stream())
    return perm


def encode_continuous(x, tab, valid_idx, out64=None, out32=None, col_off=0, u_inj=None, rng=None, site=SITE_ENCODE,
                      comp_out=None):
    _chk(x, torch.float64, 'x')
    _chk(tab, torch.float64, 'tab')
    _chk(valid_idx, torch.int32, 'valid_idx')
    out = out64 if out64 is not None else out32
    lib().call('dpc_encode_continuous', _p(x), x.stride(0), x.shape[0], _p(tab), tab.shape[1], _p(valid_idx),
               valid_idx.shape[0], _p(u_inj), _p(rng), site, _p(out64), _p(out32), _ld(out), col_off, _p(comp_out),
               _stream())


def gmm_estep(x, fit_col, params, prior, info, part, init, resp_inj=None, rng=None, site=0x4000):
    """x fp64 [ncol, N] (column-major table), params [F, 8, K], prior [F, 2], info [F, 4], part [F, R, 3K+1]."""
    for t, n in ((x, 'x'), (params, 'params'), (prior, 'prior'), (info, 'info'), (part, 'part'), (resp_inj, 'resp_inj')):
        _chk(t, torch.float64, n)
    _chk(fit_col, torch.int32, 'fit_col')
    F, _, K = params.shape
    assert x.stride(1) == 1 and part.shape[0] == F and part.shape[2] == 3 * K + 1 and part.is_contiguous()
    assert params.is_contiguous() and prior.shape == (F, 2) and info.shape == (F, 4) and fit_col.shape[0] == F
    if resp_inj is not None:
        assert resp_inj.shape == (F, x.shape[1], K) and resp_inj.is_contiguous()
    lib().call('dpc_gmm_estep', _p(x), x.stride(0), x.shape[1], _p(fit_col), F, _p(params), K, _p(prior), _p(info),
               1 if init else 0, _p(resp_inj), _p(rng), site, _p(part), part.shape[1], _stream())


def gmm_mstep(part, prior, params, info, init, weight_concentration_prior=1e-3, mean_precision_prior=1.0,
              dof_prior=1.0, reg_covar=1e-6, tol=1e-3, max_iter=2000):
    F, _, K = params.shape
    lib().call('dpc_gmm_mstep', _p(part), part.shape[1], F, K, _p(prior), _p(params), _p(info), 1 if init else 0,
               weight_concentration_prior, mean_precision_prior, dof_prior, reg_covar, tol, max_iter, _stream())


def encode_onehot(codes, dim, out64=None, out32=None, col_off=0):
    _chk(codes, torch.int32, 'codes')
    out = out64 if out64 is not None else out32
    lib().call('dpc_encode_onehot', _p(codes), codes.shape[0], dim, _p(out64), _p(out32), _ld(out), col_off, _stream())


def colreduce_partial(mode, x, part, w=None, y=None, z=None, mean=None, invstd=None, max_R=64):
    R = ctypes.c_int(0)
    rows, F = x.shape
    lib().call('dpc_colreduce_partial', mode, _p(x), _ld(x), rows, F, _p(w), _p(y), _ld(y) if y is not None else 0,
               _p(z), _ld(z) if z is not None else 0, _p(mean), _p(invstd), _p(part), max_R, ctypes.byref(R), _stream())
    return R.value


def colreduce_rows(rows, F, max_R=64):
    """Row chunks dpc_colreduce_partial uses for a [rows, F] matrix (pick_R of csrc/nets.hip)."""
    R = min(-(-512 // -(-F // 64)), -(-rows // 64), max_R)
    return max(R, 1)


def colreduce_finish(part, R, nout, F, kind, out0, out1=None, count=1.0, eps=1e-5, momentum=0.1, run_mean=None,
                     run_var=None, accumulate=False):
    lib().call('dpc_colreduce_finish', _p(part), R, nout, F, kind, float(count), eps, momentum, _p(out0), _p(out1),
               _p(run_mean), _p(run_var), int(accumulate), _stream())


def colreduce_fold(part, R, width, nsets=1, set_stride=0, out=None, out_stride=0):
    """Fold R row-chunk partials [R, width] (per set) into one fp64 vector each, in place unless ``out`` is given."""
    _chk(part, torch.float64, 'part')
    _chk(out, torch.float64, 'out')
    lib().call('dpc_colreduce_fold', _p(part), R, width, nsets, set_stride, _p(part if out is None else out),
               out_stride if out is not None else set_stride, _stream())


class StepAdvance(ctypes.Structure):
    """dpc_step_advance (include/dpcgan_hip.h): state_advance's bookkeeping as an argument of another launch."""
    _fields_ = [('rng_a', ctypes.c_void_p), ('rng_b', ctypes.c_void_p), ('t_a', ctypes.c_void_p), ('t_b', ctypes.c_void_p),
                ('add_a', ctypes.c_int), ('add_b', ctypes.c_int)]


def step_advance(rng=None, t_a=None, t_b=None, rng_b=None, add_a=1, add_b=1):
    """The arguments of state_advance as a StepAdvance for colsum_multi(..., advance=)."""
    a = StepAdvance(_addr(rng), _addr(rng_b), _addr(t_a), _addr(t_b), int(add_a), int(add_b))
    a._keep = (rng, rng_b, t_a, t_b)
    return a


def colsum_multi(problems, part, accumulate=False, atomic=False, advance=None):
    """problems: [(x [rows, F], w [rows] or None, out [F])], up to 8: out = sum_r w[r] x[r, :], two launches in all;
    ``atomic``: one launch, the row chunks add into (pre-zeroed / accumulated) out with fp32 atomics; ``advance`` =
    step_advance(...): the launch's first block also does the end-of-step bookkeeping (dpc_colsum_multi_adv)."""
    n = len(problems)
    for x, w, out in problems:
        _chk(x, torch.float32, 'x')
        _chk(w, torch.float32, 'w')
        _chk(out, torch.float32, 'out')
        if out.numel() != x.shape[1] or (w is not None and w.numel() != x.shape[0]):
            raise ValueError('colsum_multi: out needs F entries, w one per row')
    _chk(part, torch.float64, 'part')
    P = ctypes.c_void_p * n
    I64 = ctypes.c_int64 * n
    I32 = ctypes.c_int * n
    tail = (_stream(),) if advance is None else (ctypes.byref(advance), _stream())
    lib().call('dpc_colsum_multi' if advance is None else 'dpc_colsum_multi_adv',
               n, P(*[_addr(x) for x, _, _ in problems]), I64(*[_ld(x) for x, _, _ in problems]),
               I64(*[x.shape[0] for x, _, _ in problems]), I32(*[x.shape[1] for x, _, _ in problems]),
               P(*[_addr(w) for _, w, _ in problems]), P(*[_addr(o) for _, _, o in problems]), 2 if atomic else int(accumulate), _p(part),
               part.numel() if part is not None else 0, *tail)


def colsum_noalloc(x, out, w=None, accumulate=False):
    """out[f] (+)= sum_r w[r] x[r, f] through the library's static scratch (no workspace, no allocation)."""
    _chk(x, torch.float32, 'x')
    _chk(out, torch.float32, 'out')
    _chk(w, torch.float32, 'w')
    rows, F = x.shape
    if out.numel() != F or (w is not None and w.numel() != rows):
        raise ValueError('colsum_noalloc: out needs F entries, w one per row')
    lib().call('dpc_colsum_noalloc', _p(x), _ld(x), rows, F, _p(w), _p(out), int(accumulate), _stream())
    return out


def bn_relu_fwd(z, y, mean, invstd, gamma, beta):
    lib().call('dpc_bn_relu_fwd', _p(z), _ld(z), _p(y), _ld(y), z.shape[0], z.shape[1], _p(mean), _p(invstd), _p(gamma),
               _p(beta), _stream())


def bn_relu_bwd(dy, y, z, mean, invstd, gamma, dbeta, dgamma, count, dz):
    lib().call('dpc_bn_relu_bwd', _p(dy), _ld(dy), _p(y), _ld(y), _p(z), _ld(z), z.shape[0], z.shape[1], _p(mean),
               _p(invstd), _p(gamma), _p(dbeta), _p(dgamma), float(count), _p(dz), _ld(dz), _stream())


def bn_relu_fwd_fused(z, y, part, R, count, gamma, beta, mean_out, invstd_out, run_mean=None, run_var=None, eps=1e-5,
                      momentum=0.1):
    """colreduce_finish (kind 0) + bn_relu_fwd in one launch."""
    _chk(part, torch.float64, 'part')
    lib().call('dpc_bn_relu_fwd_fused', _p(z), _ld(z), _p(y), _ld(y), z.shape[0], z.shape[1], _p(part), R, float(count),
               eps, momentum, _p(gamma), _p(beta), _p(mean_out), _p(invstd_out), _p(run_mean), _p(run_var), _stream())


def bn_relu_bwd_fused(dy, y, z, part, R, count, mean, invstd, gamma, dbeta, dgamma, dz, grad_scale=1.0):
    """colreduce_finish (kind 2) + bn_relu_bwd in one launch; dbeta / dgamma are written times grad_scale."""
    _chk(part, torch.float64, 'part')
    lib().call('dpc_bn_relu_bwd_fused', _p(dy), _ld(dy), _p(y), _ld(y), _p(z), _ld(z), z.shape[0], z.shape[1], _p(part), R,
               float(count), _p(mean), _p(invstd), _p(gamma), _p(dbeta), _p(dgamma), _p(dz), _ld(dz), float(grad_scale),
               _stream())


def disc_out_fwd(h, w, b, y, rowscale=None, slope=None, delta=None):
    """y = h.w + b; with delta also delta = rowscale[:,None] * w * slope (seed of the backward chain)."""
    lib().call('dpc_disc_out_fwd', _p(h), _ld(h), _p(w), _p(b), h.shape[0], h.shape[1], _p(y), _p(rowscale), _p(slope),
               _ld(slope) if slope is not None else 0, _p(delta), _ld(delta) if delta is not None else 0, _stream())


def disc_seed(rowscale, w, slope, delta):
    lib().call('dpc_disc_seed', _p(rowscale), _p(w), _p(slope), _ld(slope), slope.shape[0], slope.shape[1], _p(delta),
               _ld(delta), _stream())


def interpolate(real, fake, out, n, group_len, alpha_inj=None, rng=None, site=SITE_ALPHA, alpha_out=None):
    lib().call('dpc_interpolate', _p(real), _p(fake), _p(out), n, group_len, _p(alpha_inj), _p(rng), site,
               _p(alpha_out), _stream())


def gp_norm(g0, coef, pen_rows, nrm_out=None):
    lib().call('dpc_gp_norm', _p(g0), _ld(g0), g0.shape[0], g0.shape[1], coef, _p(pen_rows), _p(nrm_out), _stream())


def disc_first_layer_act(zpre, bias, n, H, S, alpha_inj=None, keep=None, rng=None, site_alpha=SITE_ALPHA,
                         site_keep=SITE_DROP0, leak=0.2, alpha_out=None, stacked=True, clear=False, gat=None,
                         sel_a=None, sel_b=None, pac=1, C=0):
    """H, S [3n, d] of the stacked rows [fake | real | interpolates] from zpre [2n, d] = x W1^T; not
    ``stacked``: H, S [n, d] from zpre [n, d].  ``clear``: zpre is zeroed once read."""
    _chk(keep, torch.uint8, 'keep')
    d = zpre.shape[1]
    zi, zo = (2, 3) if stacked else (1, 1)
    if zpre.shape[0] != zi * n or H.shape[0] != zo * n or S.shape[0] != zo * n or H.shape[1] != d or S.shape[1] != d:
        raise ValueError('disc_first_layer_act: zpre needs %dn rows, H and S %dn rows of the same width' % (zi, zo))
    if keep is not None and (keep.shape[0] != zo * n or keep.shape[1] != d):
        raise ValueError('disc_first_layer_act: keep must be [%dn, d]' % zo)
    if gat is not None:
        _chk(gat, torch.float32, 'gat')
        _chk(sel_a, torch.int32, 'sel_a')
        _chk(sel_b, torch.int32, 'sel_b')
        if gat.shape[0] != pac * C or gat.shape[1] != d or sel_a.numel() != 2 * n * pac or \
                (stacked and (sel_b is None or sel_b.numel() != 2 * n * pac)):
            raise ValueError('disc_first_layer_act: gat must be [pac*C, d], the id tables [n*pac, 2]')
    lib().call('dpc_disc_first_layer_act', _p(zpre), _ld(zpre), _p(bias), n, d, _p(alpha_inj), _p(keep),
               _ld(keep) if keep is not None else 0, _p(rng), site_alpha, site_keep, leak, _p(H), _ld(H), _p(S), _ld(S),
               _p(alpha_out), int(bool(stacked)), int(bool(clear)), _p(gat), _ld(gat) if gat is not None else 0,
               _p(sel_a), _p(sel_b), int(pac), int(C), _stream())


def gp_gram_rows(T, u, slope, coef, pen_rows, f_out, v1, q_out=None, nrm_out=None):
    _chk(pen_rows, torch.float64, 'pen_rows')
    _chk(q_out, torch.float64, 'q_out')
    n, d = T.shape
    if u.shape != T.shape or slope.shape != T.shape or v1.shape != T.shape:
        raise ValueError('gp_gram_rows: T, u, slope, v1 must have one shape')
    lib().call('dpc_gp_gram_rows', _p(T), _ld(T), _p(u), _ld(u), _p(slope), _ld(slope), n, d, coef, _p(pen_rows),
               _p(f_out), _p(q_out), _p(nrm_out), _p(v1), _ld(v1), _stream())


def sum_f64(v, scale, out, accumulate=False):
    lib().call('dpc_sum_f64', _p(v), v.numel(), float(scale), _p(out), int(accumulate), _stream())


def mean_pair(y, n, sa, sb, out, accumulate=False):
    lib().call('dpc_mean_pair', _p(y), n, sa, sb, _p(out), int(accumulate), _stream())


def adam_step(p, g, m, v, lr, wd, b1=0.5, b2=0.9, eps=1e-8, clamp=0.0, gscale=1.0, t_dev=None, t_host=1,
              seg_off=None, seg_add=None, noise_scale=0.0, noise_inj=None, rng=None, site=SITE_NOISE_D):
    nseg = seg_add.shape[0] if seg_add is not None else 0
    lib().call('dpc_adam_step', _p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, wd, clamp, gscale, _p(t_dev),
               t_host, _p(seg_off), _p(seg_add), nseg, noise_scale, _p(noise_inj), _p(rng), site, _stream())


def adam_step_slices(p, g, m, v, lr, wd, mats, slices, b1=0.5, b2=0.9, eps=1e-8, clamp=0.0, gscale=1.0, t_dev=None,
                     t_host=1, seg_off=None, seg_add=None, noise_scale=0.0, noise_inj=None, rng=None, site=SITE_NOISE_D):
    """adam_step over a flat buffer with cond slices (dpc_adam_step_slices).  mats: [(W, period, col0)] with W a
    [rows, ld] view INTO p whose columns c with c % period >= col0 are cond columns; slices: [(W, col0, ncol, gT, pT)]
    - columns [col0, col0 + ncol) of W, gT / pT the transposed gradient addend / parameter copy [ncol, >= rows]
    (either may be None)."""
    def off(W):
        o = (W.data_ptr() - p.data_ptr()) // 4
        if o < 0 or o + W.shape[0] * W.shape[1] > p.numel() or not W.is_contiguous():
            raise ValueError('adam_step_slices: matrices must be contiguous views into p')
        return o
    nm, ns = len(mats), len(slices)
    I64m, I32m = ctypes.c_int64 * max(nm, 1), ctypes.c_int * max(nm, 1)
    I64s, I32s, Ps = ctypes.c_int64 * max(ns, 1), ctypes.c_int * max(ns, 1), ctypes.c_void_p * max(ns, 1)
    for _, _, _, gT, pT in slices:
        _chk(gT, torch.float32, 'gT')
        _chk(pT, torch.float32, 'pT')
    nseg = seg_add.shape[0] if seg_add is not None else 0
    lib().call('dpc_adam_step_slices', _p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, wd, clamp, gscale, _p(t_dev),
               t_host, _p(seg_off), _p(seg_add), nseg, noise_scale, _p(noise_inj), _p(rng), site,
               nm, I64m(*[off(W) for W, _, _ in mats]), I32m(*[W.shape[0] for W, _, _ in mats]),
               I32m(*[W.shape[1] for W, _, _ in mats]), I32m(*[int(pe) for _, pe, _ in mats]), I32m(*[int(c0) for _, _, c0 in mats]),
               ns, I64s(*[off(W) for W, _, _, _, _ in slices]), I32s(*[W.shape[0] for W, _, _, _, _ in slices]),
               I32s(*[W.shape[1] for W, _, _, _, _ in slices]), I32s(*[int(c0) for _, c0, _, _, _ in slices]),
               I32s(*[int(nc) for _, _, nc, _, _ in slices]),
               Ps(*[(gT.data_ptr() if gT is not None else None) for _, _, _, gT, _ in slices]),
               I64s(*[(_ld(gT) if gT is not None else 0) for _, _, _, gT, _ in slices]),
               Ps(*[(pT.data_ptr() if pT is not None else None) for _, _, _, _, pT in slices]),
               I64s(*[(_ld(pT) if pT is not None else 0) for _, _, _, _, pT in slices]), _stream())


def noise_add_clamp(x, scale, noise_inj=None, rng=None, site=SITE_NOISE_D, clamp=0.0):
    lib().call('dpc_noise_add_clamp', _p(x), x.numel(), scale, _p(noise_inj), _p(rng), site, clamp, _stream())
    return x


def persample_gram(delta, act, n, norm2, with_bias=True, accumulate=False):
    """norm2[i] (+)= squared gradient norm of example i (rows i, n+i, 2n+i) for one Linear layer."""
    _chk(norm2, torch.float64, 'norm2')
    if delta.shape[0] != 3 * n or act.shape[0] != 3 * n or norm2.shape[0] != n:
        raise ValueError('persample_gram: delta/act need 3n rows, norm2 n entries')
    lib().call('dpc_persample_gram', _p(delta), _ld(delta), delta.shape[1], _p(act), _ld(act), act.shape[1], n,
               int(with_bias), int(accumulate), _p(norm2), _stream())


def clip_factors(norm2, clip, gscale, rowc, roww=None, rowc_w=None, norm_out=None):
    n = norm2.shape[0]
    if rowc.shape[0] != 3 * n:
        raise ValueError('clip_factors: rowc needs 3n entries')
    lib().call('dpc_clip_factors', _p(norm2), n, float(clip), float(gscale), _p(rowc), _p(roww), _p(rowc_w),
               _p(norm_out), _stream())


def hook_noise(seg_add, fires, t_dev, inv_B, rng, site=SITE_HOOK, t_add=1):
    lib().call('dpc_hook_noise', _p(seg_add), seg_add.shape[0], _p(fires), _p(t_dev), int(t_add), float(inv_B), _p(rng),
               site, _stream())


def persample_clip_all(layers, n, clip, gscale, norm2, rowc, roww=None, rowc_w=None, norm_out=None, zpre=None,
                       gp_f=None, gp_q=None, fc_out=None, clear_zpre=False, sel_a=None, sel_b=None, pac=1):
    """layers: [(delta [3n,d], act [3n,K])]; all Gram contractions and the clip factors in one launch.
    With ``zpre`` the penalty row of layer 0's act is implicit (act[0] needs only its first 2n rows):
    see dpc_persample_clip_all_gram."""
    _chk(norm2, torch.float64, 'norm2')
    _chk(gp_q, torch.float64, 'gp_q')
    L = len(layers)
    for l, (dl, ac) in enumerate(layers):
        _chk(dl, torch.float32, 'delta')
        _chk(ac, torch.float32, 'act')
        need = 2 * n if (l == 0 and zpre is not None) else 3 * n
        if dl.shape[0] != 3 * n or ac.shape[0] < need:
            raise ValueError('persample_clip_all: delta/act need 3n rows')
    P = ctypes.c_void_p * L
    I64 = ctypes.c_int64 * L
    I32 = ctypes.c_int * L
    args = [L, P(*[dl.data_ptr() for dl, _ in layers]), I64(*[_ld(dl) for dl, _ in layers]),
            I32(*[dl.shape[1] for dl, _ in layers]), P(*[ac.data_ptr() for _, ac in layers]),
            I64(*[_ld(ac) for _, ac in layers]), I32(*[ac.shape[1] for _, ac in layers]), n, float(clip),
            float(gscale), _p(norm2), _p(rowc), _p(roww), _p(rowc_w), _p(norm_out)]
    if zpre is None:
        lib().call('dpc_persample_clip_all', *args, _stream())
    else:
        if zpre.shape[0] != 2 * n or zpre.shape[1] != layers[0][0].shape[1]:
            raise ValueError('persample_clip_all: zpre must be [2n, d0]')
        _chk(sel_a, torch.int32, 'sel_a')
        _chk(sel_b, torch.int32, 'sel_b')
        lib().call('dpc_persample_clip_all_gram', *args, _p(zpre), _ld(zpre), _p(gp_f), _p(gp_q), _p(fc_out),
                   int(bool(clear_zpre)), _p(sel_a), _p(sel_b), int(pac), _stream())


def condvec_original(table, cond_st, dims, matrix_st, cond, rows_inj=None, cols_inj=None, rng=None, site=SITE_CONDVEC):
    _chk(rows_inj, torch.int64, 'rows_inj')
    _chk(cols_inj, torch.int64, 'cols_inj')
    _chk(matrix_st, torch.int32, 'matrix_st')
    lib().call('dpc_condvec_original', _p(table), _ld(table), table.shape[0], _p(cond_st), _p(dims), _p(matrix_st),
               dims.shape[0], cond.shape[1], _p(rows_inj), _p(cols_inj), _p(rng), site, cond.shape[0], _p(cond), _ld(cond),
               _stream())
    return cond


def inverse_transform(data, desc, params, Kc, out):
    _chk(desc, torch.int32, 'desc')
    _chk(params, torch.float64, 'params')
    _chk(out, torch.float64, 'out')
    lib().call('dpc_inverse_transform', _p(data), _ld(data), data.shape[0], _p(desc), desc.shape[0], _p(params), int(Kc),
               _p(out), _ld(out), _stream())
    return out
