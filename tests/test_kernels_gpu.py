"""Per-kernel parity: HIP kernels (through the C-ABI) vs the CPU oracle / golden fixtures.
Tolerance for fp32 results: 1e-5 relative to the tensor's scale (north_star); integer / index
results are bit-exact."""
import numpy as np
import pytest
import torch

from golden_util import Trace, load

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def dev():
    return torch.device('cuda:0')


def close(a, b, rtol=RTOL, what='', scale=None):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    sc = max(np.abs(b).max(), 1e-30) if scale is None else scale
    err = np.abs(a - b).max() / sc
    assert err <= rtol, '%s: max err / scale = %.3e' % (what, err)


@pytest.fixture(scope='module')
def ops():
    from dp_cgans_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------
# GEMM
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(50, 64, 940), (500, 256, 166), (300, 56, 678), (1, 1, 1), (64, 64, 16),
                                   (129, 67, 33), (1024, 256, 4888), (3072, 335, 916), (257, 4888, 256)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False)])
def test_gemm_shapes(ops, M, N, K, ta, tb):
    if M * N * K > 3e9:
        pytest.skip('big')
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    C = torch.full((M, N), float('nan'), device=dev())
    ops.gemm(A.to(dev()), B.to(dev()), C, trans_a=ta, trans_b=tb)
    close(C, ref, what='gemm %dx%dx%d' % (M, N, K))


def test_gemm_strided_views_bias_beta_and_unaligned(ops):
    g = torch.Generator().manual_seed(5)
    # operands that are column slices of wider buffers (the concat-free generator layout), odd ld
    X = torch.randn(130, 333, generator=g).to(dev())
    W = torch.randn(77, 201, generator=g).to(dev())
    bias = torch.randn(77, generator=g).to(dev())
    Cbuf = torch.randn(130, 99, generator=g).to(dev())
    C0 = Cbuf.clone()
    xin = X[:, 132:333]           # [130, 201], ld 333, start not 16-byte aligned
    out = Cbuf[:, 5:82]
    ops.gemm(xin, W, out, trans_b=True, bias=bias, alpha=0.5, beta=2.0)
    ref = 0.5 * (xin.double().cpu() @ W.double().cpu().t()) + bias.double().cpu() + 2.0 * C0[:, 5:82].double().cpu()
    close(out, ref, what='strided gemm')
    assert torch.equal(Cbuf[:, :5], C0[:, :5]) and torch.equal(Cbuf[:, 82:], C0[:, 82:])


@pytest.mark.parametrize('splitk', [1, 4, 0])
def test_gemm_weight_gradient_shape_splitk(ops, splitk):
    g = torch.Generator().manual_seed(9)
    dY = torch.randn(8192, 256, generator=g).to(dev())
    X = torch.randn(8192, 404, generator=g).to(dev())
    dW = torch.zeros(256, 404, device=dev())
    ops.gemm(dY, X, dW, trans_a=True, splitk=splitk)
    close(dW, dY.double().cpu().t() @ X.double().cpu(), what='dW splitk=%d' % splitk)
    acc = dW.clone()
    ops.gemm(dY, X, dW, trans_a=True, beta=1.0, splitk=splitk)
    close(dW, 2 * acc.double().cpu(), what='dW accumulate')


def test_gemm_epilogues(ops):
    g = torch.Generator().manual_seed(11)
    X = torch.randn(96, 300, generator=g)
    W = torch.randn(70, 300, generator=g)
    b = torch.randn(70, generator=g)
    keep = (torch.rand(96, 70, generator=g) > 0.5).to(torch.uint8)
    pre = X.double() @ W.double().t() + b.double()
    slope = torch.where(pre > 0, 1.0, 0.2) * keep.double() * 2
    H = torch.empty(96, 70, device=dev())
    S = torch.empty(96, 70, device=dev())
    ops.gemm(X.to(dev()), W.to(dev()), H, trans_b=True, bias=b.to(dev()), epi=ops.EPI_LEAKY_DROPOUT, aux_out=S,
             keep=keep.to(dev()))
    close(S, slope, what='slope')
    close(H, pre * slope, what='leaky dropout')
    # Philox mask: the epilogue must use exactly the bits dpc_fill(kind=KEEP) exposes
    rng = ops.new_rng(1234, dev())
    kb = torch.empty(96, 70, dtype=torch.uint8, device=dev())
    ops.fill(kb, ops.FILL_KEEP, rng, 17)
    ops.gemm(X.to(dev()), W.to(dev()), H, trans_b=True, bias=b.to(dev()), epi=ops.EPI_LEAKY_DROPOUT, aux_out=S,
             rng=rng, site=17)
    slope2 = torch.where(pre > 0, 1.0, 0.2) * kb.cpu().double() * 2
    close(S, slope2, what='slope philox')
    assert 0.4 < kb.float().mean().item() < 0.6
    # multiply epilogue
    Mul = torch.randn(96, 70, generator=g)
    ops.gemm(X.to(dev()), W.to(dev()), H, trans_b=True, epi=ops.EPI_MUL, aux_in=Mul.to(dev()))
    close(H, (X.double() @ W.double().t()) * Mul.double(), what='mul epilogue')


@pytest.mark.parametrize('M,N,K', [(1024, 256, 256), (3072, 256, 256), (256, 256, 1024), (130, 77, 201), (33, 31, 64), (500, 96, 1000)])
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False)])
def test_gemm_small_product_kernel_whose_waves_split_k(ops, M, N, K, ta, tb):
    """tile=32 (gemm_ksplit_kernel: 32x32 tiles, the block's four waves split K): every layout, the shapes of
    the discriminator's K = 256 chain links, ragged edges, each epilogue, beta, and the clip scale of the
    weight-gradient layout - against fp64."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    d_ = dev()
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    prod = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    Ad, Bd = A.to(d_), B.to(d_)
    C = torch.full((M, N), float('nan'), device=d_)
    ops.gemm(Ad, Bd, C, trans_a=ta, trans_b=tb, tile=32)
    close(C, prod, what='plain')
    C64 = torch.empty(M, N, device=d_)
    ops.gemm(Ad, Bd, C64, trans_a=ta, trans_b=tb, tile=64, splitk=1)
    close(C, C64, what='vs the 64x64 tile', rtol=2e-6)
    ops.gemm(Ad, Bd, C64, trans_a=ta, trans_b=tb, tile=324)
    close(C64, prod, what='groups of 4 K-tiles')
    bias = torch.randn(N, generator=g)
    base = torch.randn(M, N, generator=g)
    C.copy_(base)
    ops.gemm(Ad, Bd, C, trans_a=ta, trans_b=tb, tile=32, bias=bias.to(d_), alpha=0.5, beta=2.0)
    close(C, 0.5 * prod + bias.double() + 2.0 * base.double(), what='bias, alpha, beta')
    mul = torch.randn(M, N, generator=g)
    ops.gemm(Ad, Bd, C, trans_a=ta, trans_b=tb, tile=32, epi=ops.EPI_MUL, aux_in=mul.to(d_))
    close(C, prod * mul.double(), what='multiply epilogue')
    ops.gemm(Ad, Bd, C, trans_a=ta, trans_b=tb, tile=32, epi=ops.EPI_RELU, bias=bias.to(d_))
    close(C, torch.relu(prod + bias.double()), what='ReLU epilogue')
    rng = ops.new_rng(99, d_)
    kb = torch.empty(M, N, dtype=torch.uint8, device=d_)
    ops.fill(kb, ops.FILL_KEEP, rng, 21)
    S = torch.empty(M, N, device=d_)
    ops.gemm(Ad, Bd, C, trans_a=ta, trans_b=tb, tile=32, bias=bias.to(d_), epi=ops.EPI_LEAKY_DROPOUT, aux_out=S, rng=rng, site=21)
    pre = prod + bias.double()
    slope = torch.where(pre > 0, 1.0, 0.2) * kb.cpu().double() * 2
    ok = pre.abs() > 1e-5 * float(pre.abs().max())                # (a pre-activation within rounding of 0 may take either slope)
    close(torch.where(ok.to(d_), S, 0.0), torch.where(ok, slope, 0.0), what='slopes from the Philox keep bits')
    close(torch.where(ok.to(d_), C, 0.0), torch.where(ok, pre * slope, 0.0), what='LeakyReLU + Dropout epilogue')
    if ta:
        ks = torch.rand(K, generator=g)
        ops.gemm(Ad, Bd, C, trans_a=True, trans_b=tb, tile=32, kscale=ks.to(d_))
        close(C, (A * ks[:, None]).t().double() @ (B.t() if tb else B).double(), what='clip scale')


@pytest.mark.parametrize('n,pac,Dd,C,d0', [(64, 8, 335, 276, 256), (10, 10, 56, 38, 128), (37, 3, 40, 30, 21), (1024, 8, 335, 276, 256)])
def test_gemm_segment_maps_run_over_the_data_columns_of_the_pac_row_layout(ops, n, pac, Dd, C, d0):
    """The three products of the discriminator's first layer over the DATA columns of its pac-row input
    [pac][Dd data | C cond] with W1 left in the reference's layout (dp_cgan.py:66-68, 622-623): forward
    x W1^T (K segments on B), input gradient dy W1 (N segments on B), weight gradient dy^T x (N segments
    on C, accumulated) - each against the dense fp64 product over the data columns."""
    g = torch.Generator().manual_seed(n + Dd)
    d_ = dev()
    row, P = Dd + C, (Dd + 31) // 32 * 32
    K0, Kc = pac * row, pac * P
    W1 = (torch.randn(d0, K0, generator=g) / K0 ** 0.5)
    W1d = torch.randn(d0, (K0 + 3) // 4 * 4, generator=g)[:, :K0].to(d_)           # rows padded to 16 bytes, as the engine's are
    W1d.copy_(W1)
    xd = torch.randn(n, pac, Dd, generator=g)
    xc = torch.zeros(n, pac, P)
    xc[:, :, :Dd] = xd                                                                # compact rows: zero padding
    Xc = xc.reshape(n, Kc).to(d_)
    W1data = W1.reshape(d0, pac, row)[:, :, :Dd]                                     # [d0, pac, Dd]
    seg = (Dd, P, row)
    # forward
    ref = torch.einsum('rsj,osj->ro', xd.double(), W1data.double())
    for splitk in (1, 0):
        out = torch.zeros(n, d0, device=d_)
        ops.gemm(Xc, W1d, out, trans_b=True, beta=1.0, seg_b=1, seg=seg, splitk=splitk)
        close(out, ref, what='forward over data columns (splitk %d)' % splitk)
    # input gradient (compact output; the padding columns hold products with cond weights and are never read)
    dy = torch.randn(n, d0, generator=g)
    dX = torch.full((n, Kc), float('nan'), device=d_)
    ops.gemm(dy.to(d_), W1d, dX, seg_b=2, seg=seg)
    ref = torch.einsum('ro,osj->rsj', dy.double(), W1data.double())
    close(dX.view(n, pac, P)[:, :, :Dd], ref, what='input gradient over data columns')
    # weight gradient accumulated into W1's layout: data columns get dy^T x added, cond columns stay
    base = torch.randn(d0, K0, generator=g)
    ks = torch.rand(n, generator=g)
    for splitk in (1, 0):
        dW = torch.empty_like(W1d)
        dW.copy_(base)
        ops.gemm(dy.to(d_), Xc, dW, trans_a=True, beta=1.0, seg_c=1, seg=seg, kscale=ks.to(d_), splitk=splitk)
        ref = base.double().reshape(d0, pac, row).clone()
        ref[:, :, :Dd] += torch.einsum('ro,rsj->osj', (dy * ks[:, None]).double(), xd.double())
        close(dW, ref.reshape(d0, K0), what='weight gradient into the data columns (splitk %d)' % splitk)
        assert torch.equal(dW.view(d0, pac, row)[:, :, Dd:].cpu(), base.view(d0, pac, row)[:, :, Dd:])
    with pytest.raises(RuntimeError, match='seg_pad'):
        ops.gemm(Xc, W1d, out, trans_b=True, beta=1.0, seg_b=1, seg=(Dd, P + 4, row))
    with pytest.raises(RuntimeError, match='runs past'):       # a padded segment longer than what is left of W1's row
        ops.gemm(Xc[:, :P].contiguous(), torch.zeros(d0, P - 1, device=d_), out, trans_b=True, beta=1.0, seg_b=1, seg=(Dd, P, row))


@pytest.mark.parametrize('M,N,K,F', [(512, 256, 128, 256), (200, 100, 37, 100), (8192, 512, 335, 256), (130, 96, 64, 40)])
def test_gemm_column_statistics_in_the_epilogue(ops, M, N, K, F):
    """colstat: BatchNorm's forward sums (sum, sum of squares of the result's columns) and, with bn_back, the two
    backward sums over the first F columns of the accumulated result - against fp64 references and against the
    separate reduction kernel (dpc_colreduce_partial mode 2)."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    C = torch.empty(M, N, device=dev())
    stat = torch.zeros(2, N, dtype=torch.float64, device=dev())
    ops.gemm(A.to(dev()), B.to(dev()), C, trans_b=True, bias=bias.to(dev()), colstat=stat)
    ref = A.double() @ B.double().t() + bias.double()
    close(C, ref, what='product')
    close(stat[0], ref.sum(0), what='column sums', scale=float(ref.abs().sum(0).max()))
    close(stat[1], (ref ** 2).sum(0), what='column sums of squares')
    # slots: tile row t adds into slot t % 4; the slots sum to the same statistics
    st4 = torch.zeros(4, 2, N, dtype=torch.float64, device=dev())
    ops.gemm(A.to(dev()), B.to(dev()), C, trans_b=True, bias=bias.to(dev()), colstat=st4, stat_slots=4)
    close(st4.sum(0)[0], ref.sum(0), what='slotted column sums', scale=float(ref.abs().sum(0).max()))
    close(st4.sum(0)[1], (ref ** 2).sum(0), what='slotted column sums of squares')
    assert M <= 64 or float(st4[1].abs().max()) > 0
    # backward form: dy = C0 + A B^T (beta = 1); sums of dyh and dyh * xhat over the first F columns
    C0 = torch.randn(M, N, generator=g)
    y = torch.relu(torch.randn(M, F, generator=g))
    z = torch.randn(M, F, generator=g) * 2 + 0.5
    mean, invstd = z.mean(0), 1.0 / (z.var(0, unbiased=False) + 1e-5).sqrt()
    Cd = C0.clone().to(dev())
    bst = torch.zeros(2, F, dtype=torch.float64, device=dev())
    yd, zd = y.to(dev()), z.to(dev())
    ops.gemm(A.to(dev()), B.to(dev()), Cd, trans_b=True, beta=1.0, colstat=bst,
             bn_back=(yd, zd, mean.to(dev()), invstd.to(dev())))
    dy = C0.double() + A.double() @ B.double().t()
    close(Cd, dy, what='accumulated product')
    dyh = torch.where(y > 0, dy[:, :F], torch.zeros((), dtype=torch.float64))
    xh = (z.double() - mean.double()) * invstd.double()
    sc = float(dyh.abs().sum(0).max())
    close(bst[0], dyh.sum(0), what='sum dyh', scale=sc)
    close(bst[1], (dyh * xh).sum(0), what='sum dyh xhat', scale=float((dyh * xh).abs().sum(0).max()))
    part = torch.empty(64 * 2 * F, dtype=torch.float64, device=dev())
    R = ops.colreduce_partial(2, Cd[:, :F], part, y=yd, z=zd, mean=mean.to(dev()), invstd=invstd.to(dev()))
    ops.colreduce_fold(part, R, 2 * F)
    close(bst.view(-1), part[:2 * F], what='epilogue sums vs the reduction kernel', scale=sc, rtol=1e-6)


def test_colsum_multi_all_modes(ops):
    """dpc_colsum_multi: several weighted column sums in one launch pair (overwrite / accumulate, bit-identical to
    dpc_colreduce_partial(mode 1) + dpc_colreduce_finish(kind 1) per sum) and in ONE launch with fp32 atomics."""
    g = torch.Generator().manual_seed(8)
    shapes = [(3072, 256, True), (2048, 256, False), (3072, 1, True), (100, 37, False), (8192, 335, True)]
    xs = [torch.randn(r, F, generator=g).to(dev()) for r, F, _ in shapes]
    ws = [torch.randn(r, generator=g).to(dev()) if w else None for r, F, w in shapes]
    refs = [((x.double() * (w.double()[:, None] if w is not None else 1.0)).sum(0)).cpu() for x, w in zip(xs, ws)]
    part = torch.empty(64 * sum(F for _, F, _ in shapes), dtype=torch.float64, device=dev())
    outs = [torch.full((F,), float('nan'), device=dev()) for _, F, _ in shapes]
    ops.colsum_multi(list(zip(xs, ws, outs)), part)
    for o, x, w, r in zip(outs, xs, ws, refs):
        close(o, r, what='colsum_multi', scale=float((x.double().abs() * (w.double().abs()[:, None] if w is not None else 1.0)).sum(0).max()))
        single = torch.empty_like(o)
        R = ops.colreduce_partial(1, x, part, w=w)
        ops.colreduce_finish(part, R, 1, x.shape[1], 1, single)
        assert torch.equal(single, o)
    base = [o.clone() for o in outs]
    ops.colsum_multi(list(zip(xs, ws, outs)), part, accumulate=True)
    for o, b in zip(outs, base):
        assert torch.equal(o, b + b)
    at = [torch.zeros_like(o) for o in outs]
    ops.colsum_multi(list(zip(xs, ws, at)), None, atomic=True)
    for o, x, w, r in zip(at, xs, ws, refs):
        close(o, r, what='colsum_multi atomic', scale=float((x.double().abs() * (w.double().abs()[:, None] if w is not None else 1.0)).sum(0).max()))
    # the launch that also carries the end-of-step bookkeeping (dpc_colsum_multi_adv): same sums, counters as
    # dpc_state_advance leaves them; and the optimiser launch behind it counts from the bumped step (t_host = 0)
    rng_a, rng_b = ops.new_rng(5, dev()), ops.new_rng(6, dev())
    rng_c, rng_d = rng_a.clone(), rng_b.clone()
    ta, tb = torch.full((1,), 4, dtype=torch.int32, device=dev()), torch.full((1,), 9, dtype=torch.int32, device=dev())
    tc, td = ta.clone(), tb.clone()
    outs2 = [torch.full((F,), float('nan'), device=dev()) for _, F, _ in shapes]
    ops.colsum_multi(list(zip(xs, ws, outs2)), part, advance=ops.step_advance(rng_a, t_a=ta, t_b=tb, rng_b=rng_b, add_a=3, add_b=1))
    ops.state_advance(rng_c, t_a=tc, t_b=td, rng_b=rng_d, add_a=3, add_b=1)
    for o, b in zip(outs2, base):
        assert torch.equal(o, b)
    assert torch.equal(rng_a, rng_c) and torch.equal(rng_b, rng_d) and torch.equal(ta, tc) and torch.equal(tb, td)
    assert int(ta.item()) == 7 and int(tb.item()) == 10 and int(rng_a[1].item()) == 1
    n = 1000
    mk = lambda: torch.randn(n, generator=g).to(dev())
    p1, g1, m1, v1 = mk(), mk(), mk() * 0.01, mk().abs() * 0.01
    p2, m2, v2 = p1.clone(), m1.clone(), v1.clone()
    t_old = torch.full((1,), 9, dtype=torch.int32, device=dev())
    ops.adam_step(p1, g1, m1, v1, 2e-4, 1e-6, t_dev=t_old, t_host=1)          # step 10, counted the usual way
    ops.adam_step(p2, g1, m2, v2, 2e-4, 1e-6, t_dev=tb, t_host=0)             # the counter already says 10
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)


def test_gemm_rejects_bad_arguments(ops):
    A = torch.zeros(4, 4, device=dev())
    with pytest.raises(ValueError):
        ops.gemm(A, torch.zeros(5, 4, device=dev()), A)
    with pytest.raises(RuntimeError):
        ops.gemm(A, A, A, epi=ops.EPI_MUL)       # aux_in missing


# ------------------------------------------------------------------------------------------------
# RNG, noise add, Adam
# ------------------------------------------------------------------------------------------------
def test_philox_fills_and_noise_add(ops):
    rng = ops.new_rng(42, dev())
    n = 1 << 20
    z = torch.empty(n, device=dev())
    ops.fill(z, ops.FILL_NORMAL, rng, 3)
    z2 = torch.empty(n, device=dev())
    ops.fill(z2, ops.FILL_NORMAL, rng, 3)
    assert torch.equal(z, z2)                       # same (seed, step, site) -> same stream
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    ops.fill(z2, ops.FILL_NORMAL, rng, 4)
    assert not torch.equal(z, z2)                   # other site
    ops.state_advance(rng)
    ops.fill(z2, ops.FILL_NORMAL, rng, 3)
    assert not torch.equal(z, z2)                   # other step
    assert rng.cpu().tolist() == [42, 1]
    e = torch.empty(n, device=dev())
    ops.fill(e, ops.FILL_EXP, rng, 5)
    assert abs(e.mean().item() - 1) < 5e-3 and e.min().item() > 0
    # fp32 uniforms must stay strictly inside (0,1): -log(-log(u)) of the Gumbel draw is finite
    # (regression: a 24-bit mapping rounded its top bucket to 1.0, ~4 hits in 2^26 draws)
    uf = torch.empty(1 << 26, device=dev())
    ops.fill(uf, ops.FILL_UNIFORM, rng, 9)
    assert uf.min().item() > 0.0 and uf.max().item() < 1.0
    ef = torch.empty(1 << 26, device=dev())
    ops.fill(ef, ops.FILL_EXP, rng, 9)
    assert ef.min().item() > 0.0 and torch.isfinite(ef).all()
    del uf, ef
    u = torch.empty(n, dtype=torch.float64, device=dev())
    ops.fill(u, ops.FILL_UNIFORM64, rng, 6)
    assert 0 <= u.min().item() and u.max().item() < 1 and abs(u.mean().item() - 0.5) < 2e-3

    # noise_add_clamp with Philox == with the injected tensor of the same draws; and vs numpy
    x0 = torch.randn(n + 3, device=dev()) * 0.01
    zz = torch.empty(n + 3, device=dev())
    ops.fill(zz, ops.FILL_NORMAL, rng, 7)
    xa = ops.noise_add_clamp(x0.clone(), 0.002, rng=rng, site=7, clamp=0.01)
    xb = ops.noise_add_clamp(x0.clone(), 0.002, noise_inj=zz, clamp=0.01)
    assert torch.equal(xa, xb)
    ref = np.clip(x0.cpu().numpy() + np.float32(0.002) * zz.cpu().numpy(), -0.01, 0.01)
    close(xa, ref, what='noise_add_clamp')


@pytest.mark.parametrize('name', ['trace_np.npz', 'trace_priv.npz'])
def test_adam_matches_reference_trace(ops, name):
    """Replay the discriminator's optimiser steps of the reference trace on the flat-buffer kernel."""
    tr = Trace(name)
    steps = [s for s, k in enumerate(tr.kinds) if k == 'D']
    p0 = tr.params(steps[0], 'p_before')
    sizes = [p0[k].size for k in tr.dkeys]
    off = np.concatenate([[0], np.cumsum(sizes)])
    flat = lambda d: torch.from_numpy(np.concatenate([d[k].reshape(-1) for k in tr.dkeys])).to(dev())
    p = flat(p0)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    t_dev = torch.zeros(1, dtype=torch.int32, device=dev())
    for s in steps:
        g = flat(tr.params(s, 'g'))
        ops.adam_step(p, g, m, v, 2e-4, 1e-6, t_dev=t_dev)
        ops.state_advance(t_a=t_dev)
        ref = flat(tr.params(s, 'p_after'))
        close(p, ref, what='adam step %d' % s)
        if tr.private:
            p.clamp_(-0.01, 0.01)
    fin = torch.from_numpy(np.concatenate([tr.z['disc_final/' + k].reshape(-1) for k in tr.dkeys]))
    close(p, fin, what='final D params')
    # fused form: hook-noise segments + clamp inside the kernel == separate ops
    if tr.private:
        s = steps[0]
        p1 = flat(p0); m1 = torch.zeros_like(p1); v1 = torch.zeros_like(p1)
        seg_add = torch.randn(len(sizes), device=dev()) * 0.01
        seg_off = torch.from_numpy(off.astype(np.int64)).to(dev())
        g = flat(tr.params(s, 'g'))
        ops.adam_step(p1, g, m1, v1, 2e-4, 1e-6, t_host=1, seg_off=seg_off, seg_add=seg_add, clamp=0.01)
        g2 = g + torch.repeat_interleave(seg_add, torch.tensor(sizes, device=dev()))
        p2 = flat(p0); m2 = torch.zeros_like(p2); v2 = torch.zeros_like(p2)
        ops.adam_step(p2, g2, m2, v2, 2e-4, 1e-6, t_host=1)
        close(p1, p2.clamp(-0.01, 0.01), what='fused segment noise + clamp')


@pytest.mark.parametrize('noise', ['none', 'philox', 'injected'])
@pytest.mark.parametrize('rows_g, rows_d', [(70, 40), (32, 64), (5, 33)])
def test_adam_with_cond_slices_equals_transposes_around_the_plain_step(ops, noise, rows_g, rows_d):
    """dpc_adam_step_slices against dpc_transpose_slices (gradient addend in) + dpc_adam_step + dpc_transpose_slices
    (updated cond columns out): bit-identical parameters, moments, completed gradient and transposed copies - a
    generator-like matrix with one trailing cond slice and a discriminator-like one with pac = 2 periodic slices,
    beside tensors without slices, with hook segments, clamp and the three noise sources."""
    d_ = dev()
    g_ = torch.Generator().manual_seed(rows_g * 100 + rows_d + len(noise))
    shapes = [(7,), (rows_g, 45), (11,), (rows_d, 52), (3,)]
    offs, n = [], 0
    for sh in shapes:
        offs.append(n)
        n += (int(np.prod(sh)) + 3) // 4 * 4
    mk = lambda scale=1.0: (torch.randn(n, generator=g_) * scale).to(d_)
    p0, g0, m0, v0 = mk(0.02), mk(0.01), mk(0.001), mk(0.001).abs()
    view = lambda buf, k: buf[offs[k]:offs[k] + int(np.prod(shapes[k]))].view(shapes[k])
    # slices: (matrix index, first column, columns)
    sl = [(1, 30, 15), (3, 17, 9), (3, 43, 9)]
    gTs = [(torch.randn(nc, shapes[k][0] + 3, generator=g_) * 0.01).to(d_)[:, :shapes[k][0]] for k, c0, nc in sl]
    seg_off = torch.tensor(offs + [n], dtype=torch.int64, device=d_)
    seg_add = (torch.randn(len(shapes), generator=g_) * 0.005).to(d_)
    rng = ops.new_rng(77, d_)
    t_dev = torch.full((1,), 3, dtype=torch.int32, device=d_)
    kw = dict(clamp=0.05, t_dev=t_dev, t_host=2, seg_off=seg_off, seg_add=seg_add, gscale=0.5)
    if noise == 'philox':
        kw.update(noise_scale=0.003, rng=rng, site=ops.SITE_NOISE_D + 64)
    elif noise == 'injected':
        kw.update(noise_scale=0.003, noise_inj=mk())
    # reference: three launches
    p2, g2, m2, v2 = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    ops.transpose_slices([(gT, view(g2, k)[:, c0:c0 + nc]) for (k, c0, nc), gT in zip(sl, gTs)], accumulate=True)
    ops.adam_step(p2, g2, m2, v2, 2e-4, 1e-6, **kw)
    pT2 = [torch.full((nc, shapes[k][0] + 1), 9.0, device=d_)[:, :shapes[k][0]] for k, c0, nc in sl]
    ops.transpose_slices([(view(p2, k)[:, c0:c0 + nc], pT) for (k, c0, nc), pT in zip(sl, pT2)])
    # one launch
    p1, g1, m1, v1 = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    pT1 = [torch.full((nc, shapes[k][0] + 1), 9.0, device=d_)[:, :shapes[k][0]] for k, c0, nc in sl]
    mats = [(view(p1, 1), 45, 30), (view(p1, 3), 26, 17)]
    ops.adam_step_slices(p1, g1, m1, v1, 2e-4, 1e-6, mats,
                         [(view(p1, k), c0, nc, gT, pT) for (k, c0, nc), gT, pT in zip(sl, gTs, pT1)], **kw)
    for a, b, what in ((p1, p2, 'p'), (g1, g2, 'g'), (m1, m2, 'm'), (v1, v2, 'v')):
        assert torch.equal(a, b), what
    for a, b in zip(pT1, pT2):
        assert torch.equal(a, b)
    assert not torch.equal(p1, p0)
    # without an addend and without a copy the slices still take the plain update
    p3, g3, m3, v3 = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    ops.adam_step(p3, g3, m3, v3, 2e-4, 1e-6, **kw)
    p4, g4, m4, v4 = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    mats = [(view(p4, 1), 45, 30), (view(p4, 3), 26, 17)]
    ops.adam_step_slices(p4, g4, m4, v4, 2e-4, 1e-6, mats, [(view(p4, k), c0, nc, None, None) for k, c0, nc in sl], **kw)
    assert torch.equal(p3, p4) and torch.equal(m3, m4) and torch.equal(v3, v4) and torch.equal(g4, g0)
    # the flat pass skips every cond column of a listed matrix: a call whose slices leave one of them out (it would
    # never be updated), or name a column twice, is refused and touches nothing
    p5, g5, m5, v5 = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    mats = [(view(p5, 1), 45, 30), (view(p5, 3), 26, 17)]
    for bad in (sl[:2], [sl[0], sl[1], (3, 43, 8)], sl + [(3, 43, 9)], [sl[0], sl[1], (3, 42, 10)]):
        with pytest.raises(RuntimeError, match='adam_slices'):
            ops.adam_step_slices(p5, g5, m5, v5, 2e-4, 1e-6, mats, [(view(p5, k), c0, nc, None, None) for k, c0, nc in bad], **kw)
    torch.cuda.synchronize()
    assert torch.equal(p5, p0) and torch.equal(m5, m0) and torch.equal(v5, v0)


# ------------------------------------------------------------------------------------------------
# activation + cond loss
# ------------------------------------------------------------------------------------------------
def span_table(span_info):
    rows, st = [], 0
    for dim, sm, _ in span_info.tolist():
        rows.append((st, dim, sm))
        st += dim
    return torch.tensor(rows, dtype=torch.int32)


def col_maps(span_info):
    from oracle.nets import discrete_columns
    Dd = int(span_info[:, 0].sum())
    col2d = -np.ones(Dd, dtype=np.int32)
    col2c = np.zeros(Dd, dtype=np.int32)
    for d, (ds, cs, k) in enumerate(discrete_columns(span_info)):
        col2d[ds:ds + k] = d
        col2c[ds:ds + k] = np.arange(cs, cs + k)
    return torch.from_numpy(col2d), torch.from_numpy(col2c)


def test_apply_activate_matches_reference_trace(ops):
    from oracle import nets as onets
    tr = Trace('trace_np.npz')
    spans = span_table(tr.span_info).to(dev())
    for s in range(tr.n_steps):
        fake = torch.from_numpy(tr.get(s, 'fake'))
        expn = torch.from_numpy(tr.exp_full(s))
        y = torch.full((tr.B, fake.shape[1] + 7), float('nan'), device=dev())    # wider destination (ld > Dd)
        ops.apply_activate_fwd(fake.to(dev()), y[:, :fake.shape[1]], spans, exp_noise=expn.to(dev()))
        close(y[:, :fake.shape[1]], tr.get(s, 'fakeact'), what='fakeact step %d' % s)
        assert torch.isnan(y[:, fake.shape[1]:]).all()
    # backward vs autograd of the oracle
    fake = torch.from_numpy(tr.get(0, 'fake')).requires_grad_(True)
    expn = torch.from_numpy(tr.exp_full(0))
    yo = onets.apply_activate(fake, tr.span_info, expn)
    dy = torch.randn(yo.shape, generator=torch.Generator().manual_seed(1))
    (gref,) = torch.autograd.grad(yo, fake, dy)
    dx = torch.ones_like(dy).to(dev())
    ops.apply_activate_bwd(yo.detach().to(dev()), dy.to(dev()), dx, spans, accumulate=False)
    close(dx, gref, what='activate bwd')
    ops.apply_activate_bwd(yo.detach().to(dev()), dy.to(dev()), dx, spans, accumulate=True)
    close(dx, 2 * gref, what='activate bwd accumulate')
    # Philox path == injected path fed with the exposed draws
    rng = ops.new_rng(7, dev())
    B, Dd = fake.shape
    e = torch.empty(B, Dd, device=dev())
    ops.fill(e, ops.FILL_EXP, rng, ops.SITE_GUMBEL)
    ya = torch.empty(B, Dd, device=dev())
    yb = torch.empty(B, Dd, device=dev())
    ops.apply_activate_fwd(fake.detach().to(dev()), ya, spans, rng=rng)
    ops.apply_activate_fwd(fake.detach().to(dev()), yb, spans, exp_noise=e)
    assert torch.equal(ya, yb)
    close(ya, onets.apply_activate(fake.detach(), tr.span_info, e.cpu()), what='philox gumbel vs oracle')


def test_cond_loss_matches_reference_and_oracle(ops):
    from oracle import nets as onets
    tr = Trace('trace_np.npz')
    col2d, col2c = col_maps(tr.span_info)
    nd = len(onets.discrete_columns(tr.span_info))
    part = torch.empty(2048, dtype=torch.float64, device=dev())
    for s, kind in enumerate(tr.kinds):
        if kind != 'G':
            continue
        fake = torch.from_numpy(tr.get(s, 'fake')).requires_grad_(True)
        cond, mask = torch.from_numpy(tr.get(s, 'cond')), torch.from_numpy(tr.get(s, 'mask'))
        ref = onets.cond_loss_pair_loop(fake, cond, mask, tr.span_info)
        (gref,) = torch.autograd.grad(ref, fake)
        loss = torch.zeros(1, device=dev())
        dfake = torch.full(fake.shape, float('nan'), device=dev())
        ops.cond_loss(fake.detach().to(dev()), cond.to(dev()), mask.to(dev()), col2d.to(dev()), col2c.to(dev()), nd,
                      loss, dfake, part)
        close(loss, [float(tr.get(s, 'ce'))], what='ce vs reference')
        close(loss, [float(ref.detach())], what='ce vs oracle loop')
        close(dfake, gref, what='d ce / d fake')


def test_activation_and_cond_loss_in_one_pass_equal_the_two_kernels(ops):
    """dpc_activate_cond_loss_fwd = dpc_apply_activate_fwd + dpc_cond_loss on the generator update of the
    reference trace: identical activations and loss gradient, the loss to the reference's recorded value."""
    tr = Trace('trace_np.npz')
    from oracle import nets as onets
    spans = span_table(tr.span_info).to(dev())
    col2d, col2c = (t.to(dev()) for t in col_maps(tr.span_info))
    nd = len(onets.discrete_columns(tr.span_info))
    s = tr.kinds.index('G')
    fake = torch.from_numpy(tr.get(s, 'fake')).to(dev())
    expn = torch.from_numpy(tr.exp_full(s)).to(dev())
    cond, mask = torch.from_numpy(tr.get(s, 'cond')).to(dev()), torch.from_numpy(tr.get(s, 'mask')).to(dev())
    part = torch.zeros(2048, dtype=torch.float64, device=dev())
    y0, y1 = torch.empty_like(fake), torch.empty_like(fake)
    d0 = torch.full(fake.shape, float('nan'), device=dev())
    d1 = torch.full(fake.shape, float('nan'), device=dev())
    loss0 = torch.zeros(1, device=dev())
    ops.apply_activate_fwd(fake, y0, spans, exp_noise=expn)
    ops.cond_loss(fake, cond, mask, col2d, col2c, nd, loss0, d0, part)
    part2 = torch.zeros(2048, dtype=torch.float64, device=dev())
    for B_total in (0, 4 * tr.B):               # a rank's shard of a global batch scales loss and gradient by B / B_total
        n = ops.activate_cond_loss_fwd(fake, y1, spans, cond, mask, col2d, col2c, nd, d1, part2, exp_noise=expn,
                                       batch_total=B_total)
        k = 1.0 if B_total == 0 else tr.B / B_total
        assert torch.equal(y0, y1)
        close(d1, d0 * k, what='loss gradient', rtol=1e-6)
        loss1 = float(part2[:n].sum()) * 2.0 / ((B_total or tr.B) * nd)
        close([loss1], [float(tr.get(s, 'ce')) * k], what='ce vs reference')
        close([loss1], [float(loss0) * k], what='ce vs cond_loss kernel', rtol=1e-6)


@pytest.mark.parametrize('Dd_target, B', [(6, 1), (13, 5), (335, 37), (9001, 3)])
@pytest.mark.parametrize('shift, pad', [(0, 0), (1, 0), (2, 1), (3, 3)])
def test_activation_forward_on_rows_at_any_alignment(ops, Dd_target, B, shift, pad):
    """The kernels read a Philox block's four raw values with ONE 16-byte load at 4-byte alignment, and a block that
    hangs over a row reads into the neighbouring row's storage (the matrix's first and last row clamp instead).  Rows
    that start ``shift`` floats into a buffer, ``pad`` floats apart, with NaN everywhere around them, must give
    exactly what an aligned copy gives: no neighbour value may reach a result, with and without the loss."""
    from oracle import nets as onets
    rng_np = np.random.default_rng(Dd_target + shift)
    span, Dd = [], 0
    while Dd < Dd_target:
        k = min(int(rng_np.integers(1, 7)), Dd_target - Dd)
        if rng_np.random() < 0.3 and Dd + 1 < Dd_target:
            span.append((1, 0, 2)); Dd += 1
            k = min(k, Dd_target - Dd)
        span.append((k, 1, 1)); Dd += k
    span_info = np.array(span, dtype=np.int32)
    assert Dd == Dd_target
    spans = span_table(span_info).to(dev())
    x = torch.randn(B, Dd, generator=torch.Generator().manual_seed(4)).to(dev())
    ld = Dd + pad
    buf = torch.full((shift + B * ld + 8,), float('nan'), device=dev())
    xv = buf[shift:shift + B * ld].view(B, ld)[:, :Dd]
    xv.copy_(x)
    assert xv.data_ptr() % 16 == (4 * shift) % 16
    rng = ops.new_rng(23, dev())
    y0, y1 = torch.empty(B, Dd, device=dev()), torch.empty(B, Dd, device=dev())
    ops.apply_activate_fwd(x, y0, spans, rng=rng)
    ops.apply_activate_fwd(xv, y1, spans, rng=rng)
    assert torch.isfinite(y1).all() and torch.equal(y0, y1)
    col2d, col2c = (t.to(dev()) for t in col_maps(span_info))
    cols = onets.discrete_columns(span_info)
    nd, C = len(cols), sum(k for _, _, k in cols)
    g = torch.Generator().manual_seed(5)
    cond = torch.zeros(B, C)
    mask = torch.zeros(B, nd, dtype=torch.int32)
    for b in range(B):
        d = int(torch.randint(nd, (1,), generator=g))
        mask[b, d] = 1
        cond[b, cols[d][1] + int(torch.randint(cols[d][2], (1,), generator=g))] = 1.0
    cond, mask = cond.to(dev()), mask.to(dev())
    outs = []
    for src in (x, xv):
        part = torch.zeros(4096, dtype=torch.float64, device=dev())
        d = torch.full((B, Dd), float('nan'), device=dev())
        y = torch.empty(B, Dd, device=dev())
        n = ops.activate_cond_loss_fwd(src, y, spans, cond, mask, col2d, col2c, nd, d, part, rng=rng)
        outs.append((y, d, part[:n].sum()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], y0)
    assert torch.isfinite(outs[1][1]).all() and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize('Dd_target', [9000, 20000, 40000])
def test_activation_kernels_on_long_rows(ops, Dd_target):
    """Rows beyond the 8-row LDS tile: one row per tile (Dd 9000), one row in more than 48 KB of LDS
    (Dd 20000, TWIDE's width), and the wide-row kernels that work straight from global memory (Dd 40000).
    Forward and backward against the oracle; Philox path == injected path."""
    from oracle import nets as onets
    rng_np = np.random.default_rng(Dd_target)
    span, Dd = [], 0
    while Dd < Dd_target:
        k = int(rng_np.integers(1, 40)) if Dd % 3 else 300
        if rng_np.random() < 0.2:
            span.append((1, 0, 2)); Dd += 1
        span.append((k, 1, 1)); Dd += k
    span_info = np.array(span, dtype=np.int32)
    spans = span_table(span_info).to(dev())
    B = 24
    x = torch.randn(B, Dd, generator=torch.Generator().manual_seed(1))
    rng = ops.new_rng(11, dev())
    e = torch.empty(B, Dd, device=dev())
    ops.fill(e, ops.FILL_EXP, rng, ops.SITE_GUMBEL)
    ya, yb = torch.empty(B, Dd, device=dev()), torch.empty(B, Dd, device=dev())
    ops.apply_activate_fwd(x.to(dev()), ya, spans, rng=rng)
    ops.apply_activate_fwd(x.to(dev()), yb, spans, exp_noise=e)
    assert torch.equal(ya, yb)
    xr = x.clone().requires_grad_(True)
    yo = onets.apply_activate(xr, span_info, e.cpu())
    close(ya, yo.detach(), what='forward')
    dy = torch.randn(B, Dd, generator=torch.Generator().manual_seed(2))
    (gref,) = torch.autograd.grad(yo, xr, dy)
    dx = torch.zeros(B, Dd, device=dev())
    ops.apply_activate_bwd(yo.detach().to(dev()), dy.to(dev()), dx, spans)
    close(dx, gref, what='backward')
    # activation + conditional loss in one pass over the same wide rows == the two kernels (chunks that fit the LDS
    # tile and chunks that go straight from global memory both take the loss)
    col2d, col2c = (t.to(dev()) for t in col_maps(span_info))
    cols = onets.discrete_columns(span_info)
    nd, C = len(cols), sum(k for _, _, k in cols)
    g = torch.Generator().manual_seed(3)
    cond = torch.zeros(B, C)
    mask = torch.zeros(B, nd, dtype=torch.int32)
    for b in range(B):
        for d in torch.randperm(nd, generator=g)[:2].tolist():
            mask[b, d] = 1
            cond[b, cols[d][1] + int(torch.randint(cols[d][2], (1,), generator=g))] = 1.0
    cond, mask = cond.to(dev()), mask.to(dev())
    part0 = torch.zeros(4096, dtype=torch.float64, device=dev())
    part1 = torch.zeros(4096, dtype=torch.float64, device=dev())
    d0 = torch.full((B, Dd), float('nan'), device=dev())
    d1 = torch.full((B, Dd), float('nan'), device=dev())
    loss0 = torch.zeros(1, device=dev())
    ops.cond_loss(x.to(dev()), cond, mask, col2d, col2c, nd, loss0, d0, part0)
    y1 = torch.empty(B, Dd, device=dev())
    n = ops.activate_cond_loss_fwd(x.to(dev()), y1, spans, cond, mask, col2d, col2c, nd, d1, part1, exp_noise=e)
    assert torch.equal(y1, yb)
    close(d1, d0, what='loss gradient', rtol=1e-6)
    close([float(part1[:n].sum()) * 2.0 / (B * nd)], [float(loss0)], what='ce', rtol=1e-6)


# ------------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------------
def sampler_device_tables(ops, enc, span_info):
    from oracle.nets import discrete_columns
    cols = discrete_columns(span_info)
    cat_col = np.concatenate([np.arange(ds, ds + k) for ds, _, k in cols]).astype(np.int32)
    cond_st = np.array([c[1] for c in cols], dtype=np.int32)
    dims = np.array([c[2] for c in cols], dtype=np.int32)
    N, C = enc.shape[0], len(cat_col)
    W = (N + 63) // 64
    table = torch.from_numpy(np.ascontiguousarray(enc, dtype=np.float32)).to(dev())
    bitmaps = torch.zeros(C, W, dtype=torch.int64, device=dev())
    cnt = torch.zeros(C, C, dtype=torch.int32, device=dev())
    ops.build_bitmaps(table, torch.from_numpy(cat_col).to(dev()), bitmaps)
    ops.cooccurrence(bitmaps, cnt)
    return table, bitmaps, cnt, torch.from_numpy(cond_st).to(dev()), torch.from_numpy(dims).to(dev()), cat_col


def test_sampler_bit_exact_against_reference(ops):
    from oracle import sampler as osamp
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    table, bitmaps, cnt, cond_st, dims, cat_col = sampler_device_tables(ops, enc, span)
    onehot = enc[:, cat_col]
    assert np.array_equal(cnt.cpu().numpy(), (onehot.T @ onehot).astype(np.int32))
    B = int(z['call/B'])
    C, nd = cnt.shape[0], dims.shape[0]
    cond = torch.full((B, C), -1.0, device=dev())
    cond2 = torch.full((B, C + 3), -1.0, device=dev())
    mask = torch.full((B, nd), -1, dtype=torch.int32, device=dev())
    pid = torch.empty(B, dtype=torch.int64, device=dev())
    vid = torch.empty(B, dtype=torch.int64, device=dev())
    sel = torch.empty(B, 2, dtype=torch.int32, device=dev())
    gcount = torch.empty(B, dtype=torch.int64, device=dev())
    ops.condvec_pair_sample(cnt, cond_st, dims, enc.shape[0], B, [cond, cond2[:, 3:]], mask, pid, vid, sel, gcount,
                            pairs_inj=torch.from_numpy(z['call/pairs'].astype(np.int32)).to(dev()),
                            r_inj=torch.from_numpy(z['call/r']).to(dev()))
    assert np.array_equal(cond.cpu().numpy(), z['call/cond'])
    assert np.array_equal(cond2[:, 3:].cpu().numpy(), z['call/cond'])
    assert np.array_equal(mask.cpu().numpy(), z['call/mask'])
    assert np.array_equal(pid.cpu().numpy(), z['call/pair_id'])
    assert np.array_equal(vid.cpu().numpy(), z['call/val_idx'])
    # row pick: rank k of the reference's pick inside its ascending list -> same row id
    tb = osamp.Tables(enc, span)
    perm = z['call/perm']
    k = osamp.rank_of_row(tb, z['call/pair_id'][perm], z['call/val_idx'][perm], z['call/row_ids'])
    assert np.array_equal(gcount.cpu().numpy()[perm], osamp.group_sizes(tb, z['call/pair_id'][perm], z['call/val_idx'][perm]))
    row_id = torch.empty(B, dtype=torch.int64, device=dev())
    permd = torch.from_numpy(perm.astype(np.int64)).to(dev())
    ops.row_select(bitmaps, sel, gcount, row_id, perm=permd, k_inj=torch.from_numpy(k).to(dev()))
    assert np.array_equal(row_id.cpu().numpy(), z['call/row_ids'])
    Dd = enc.shape[1]
    out = torch.full((B, Dd + C), -1.0, device=dev())
    ops.row_gather(table, row_id, out, Dd, sel=sel, perm=permd, C=C)
    assert np.array_equal(out[:, :Dd].cpu().numpy(), z['call/real'].astype(np.float32))
    assert np.array_equal(out[:, Dd:].cpu().numpy(), z['call/cond'][perm])
    # every rank of one group returns the group's rows in ascending order
    p0, v0 = int(z['call/pair_id'][0]), int(z['call/val_idx'][0])
    rows = tb.pair_rows[p0][v0]
    sel1 = sel[0:1].repeat(len(rows), 1).contiguous()
    gc1 = gcount[0:1].repeat(len(rows)).contiguous()
    rid = torch.empty(len(rows), dtype=torch.int64, device=dev())
    ops.row_select(bitmaps, sel1, gc1, rid, k_inj=torch.arange(len(rows), device=dev()))
    assert np.array_equal(rid.cpu().numpy(), rows)


def test_sampler_philox_draws_are_valid_and_distributed(ops):
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    table, bitmaps, cnt, cond_st, dims, cat_col = sampler_device_tables(ops, enc, span)
    B = 8192
    C, nd = cnt.shape[0], dims.shape[0]
    rng = ops.new_rng(99, dev())
    cond = torch.empty(B, C, device=dev())
    mask = torch.empty(B, nd, dtype=torch.int32, device=dev())
    pid = torch.empty(B, dtype=torch.int64, device=dev())
    vid = torch.empty(B, dtype=torch.int64, device=dev())
    sel = torch.empty(B, 2, dtype=torch.int32, device=dev())
    gcount = torch.empty(B, dtype=torch.int64, device=dev())
    ops.condvec_pair_sample(cnt, cond_st, dims, enc.shape[0], B, [cond], mask, pid, vid, sel, gcount, rng=rng)
    assert (cond.sum(1) == 2).all() and (mask.sum(1) == 2).all()
    assert 0 <= pid.min().item() and pid.max().item() < nd * (nd - 1) // 2
    assert len(torch.unique(pid)) == nd * (nd - 1) // 2          # every pair shows up at B = 8192
    perm = torch.empty(B, dtype=torch.int64, device=dev())
    keys = torch.empty(2 * B, dtype=torch.int64, device=dev())
    ops.random_perm(rng, perm, keys)
    assert torch.equal(torch.sort(perm).values, torch.arange(B, device=dev()))
    row_id = torch.empty(B, dtype=torch.int64, device=dev())
    ops.row_select(bitmaps, sel, gcount, row_id, perm=perm, rng=rng)
    assert row_id.min().item() >= 0 and row_id.max().item() < enc.shape[0]
    # the picked row really has both categories of the (permuted) request
    onehot = torch.from_numpy(enc[:, cat_col].astype(np.float32)).to(dev())
    s = sel[perm].long()
    assert (onehot[row_id, s[:, 0]] == 1).all() and (onehot[row_id, s[:, 1]] == 1).all()


def test_step_inputs_equals_the_separate_launches(ops):
    """dpc_step_inputs (cond vectors + latent draw + pool fill in one launch) against dpc_condvec_pair_sample,
    dpc_fill and a memset with the same RNG state: bit-identical outputs, nothing written outside its ranges."""
    z = load('adult_sampler.npz')
    enc, span = z['encoded'], z['span_info']
    table, bitmaps, cnt, cond_st, dims, cat_col = sampler_device_tables(ops, enc, span)
    C, nd = cnt.shape[0], dims.shape[0]
    for B, E in ((8192, 128), (37, 5)):
        rng = ops.new_rng(1234, dev())
        outs = []
        for fused in (False, True):
            cond = torch.full((B, C), -1.0, device=dev())
            hg = torch.full((B, E + C + 3), -1.0, device=dev())            # z | cond inside a wider row, as in GenContext.Hg
            mask = torch.full((B, nd), -1, dtype=torch.int32, device=dev())
            pid = torch.full((B,), -1, dtype=torch.int64, device=dev())
            vid = torch.full((B,), -1, dtype=torch.int64, device=dev())
            sel = torch.full((B, 2), -1, dtype=torch.int32, device=dev())
            gcount = torch.full((B,), -1, dtype=torch.int64, device=dev())
            pool = torch.full((4096 + 256,), 7, dtype=torch.uint8, device=dev())
            zv, cv = hg[:, :E], hg[:, E:E + C]
            if fused:
                ops.step_inputs(cnt, cond_st, dims, enc.shape[0], B, [cond, cv], mask, pid, vid, sel, gcount, rng,
                                ops.SITE_CONDVEC + 64, z=zv, site_z=ops.SITE_Z + 64, zero=pool[:4096])
            else:
                pool[:4096].zero_()
                ops.fill(zv, ops.FILL_NORMAL, rng, ops.SITE_Z + 64, 0.0, 1.0)
                ops.condvec_pair_sample(cnt, cond_st, dims, enc.shape[0], B, [cond, cv], mask, pid, vid, sel, gcount,
                                        rng=rng, site=ops.SITE_CONDVEC + 64)
            outs.append([t.clone() for t in (cond, hg, mask, pid, vid, sel, gcount, pool)])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert (outs[1][7][4096:] == 7).all() and (outs[1][1][:, E + C:] == -1).all()
        assert (outs[1][0].sum(1) == 2).all()


def test_encode_bit_exact_against_reference(ops):
    from oracle import encode as oenc
    z = load('adult_encode.npz')
    table, enc, span = z['table'], z['encoded'], z['span_info'].tolist()
    N = table.shape[0]
    Dd = enc.shape[1]
    out64 = torch.full((N, Dd), -7.0, dtype=torch.float64, device=dev())
    out32 = torch.full((N, Dd), -7.0, device=dev())
    us = z['choice_u'].reshape(len(z['continuous_columns']), -1)
    cont = z['continuous_columns'].tolist()
    st, ci, si, j = 0, 0, 0, 0
    while si < len(span):
        ncol = span[si][2]
        width = sum(span[si + t][0] for t in range(ncol))
        if ci in cont:
            g = lambda k: z['gm%d/%s' % (ci, k)]
            A = oenc.gm_constants(g('weight_concentration_a'), g('weight_concentration_b'), g('mean_precision'),
                                  g('degrees_of_freedom'), g('precisions_cholesky'))
            tab = np.stack([g('means'), np.sqrt(g('covariances')), g('precisions_cholesky'), A])
            valid = np.nonzero(g('valid'))[0].astype(np.int32)
            x = torch.from_numpy(np.ascontiguousarray(table[:, ci])).to(dev())
            ops.encode_continuous(x, torch.from_numpy(tab).to(dev()), torch.from_numpy(valid).to(dev()), out64=out64,
                                  out32=out32, col_off=st, u_inj=torch.from_numpy(us[j]).to(dev()))
            j += 1
        else:
            dummies = z['ohe%d/dummies' % ci]
            codes = np.array([int(np.where(dummies == v)[0][0]) for v in table[:, ci]], dtype=np.int32)
            ops.encode_onehot(torch.from_numpy(codes).to(dev()), width, out64=out64, out32=out32, col_off=st)
        st += width
        si += ncol
        ci += 1
    assert np.array_equal(out64.cpu().numpy(), enc)
    assert np.array_equal(out32.cpu().numpy(), enc.astype(np.float32))


# ------------------------------------------------------------------------------------------------
# cond vector as hot columns
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,C,F,Kd', [(500, 38, 256, 128), (37, 9, 21, 13), (8192, 276, 335, 640), (64, 300, 64, 4900)])
def test_gather_gemm_equals_dense_linear_over_hot_columns(ops, B, C, F, Kd):
    """Linear over [dense | cond] == GEMM over the dense columns + gathered hot columns of W (two-hot,
    one-hot and empty rows), with bias and ReLU; also on a split-K shape (deferred epilogue)."""
    g = torch.Generator().manual_seed(B + C)
    d_ = dev()
    x = torch.randn(B, Kd, generator=g)
    W = torch.randn(F, Kd + C, generator=g) / (Kd + C) ** 0.5
    b = torch.randn(F, generator=g) * 0.1
    cond = torch.zeros(B, C)
    i0 = torch.randint(0, C, (B,), generator=g)
    i1 = (i0 + 1 + torch.randint(0, C - 1, (B,), generator=g)) % C
    cond[torch.arange(B), i0] = 1
    two = torch.rand(B, generator=g) < 0.7
    cond[torch.arange(B)[two], i1[two]] = 1
    cond[0] = 0                                   # a row without any hot column
    inp = torch.cat([x, cond], 1).to(d_)          # [B, Kd + C] as the generator's Hg holds it
    Wd = W.to(d_)
    sel = torch.full((B, 2), -7, dtype=torch.int32, device=d_)
    bad = torch.zeros(1, dtype=torch.int32, device=d_)
    ops.cond_to_sel(inp[:, Kd:], sel, bad)
    assert int(bad) == 0
    want = torch.full((B, 2), -1, dtype=torch.int64)
    for r in range(B):
        nz = torch.nonzero(cond[r]).reshape(-1)
        want[r, :len(nz)] = nz
    assert torch.equal(sel.cpu().long(), want)
    WcT = torch.empty(C, F, device=d_)
    ops.transpose_slices([(Wd[:, Kd:], WcT)])
    assert torch.equal(WcT.cpu(), W[:, Kd:].t())
    ref = inp.cpu().double() @ W.double().t() + b.double()
    for epi in (ops.EPI_NONE, ops.EPI_RELU):
        out = torch.full((B, F), float('nan'), device=d_)
        ops.gemm_gather(inp[:, :Kd], Wd[:, :Kd], out, WcT, sel, bias=b.to(d_), epi=epi)
        close(out, torch.relu(ref) if epi else ref, what='gather gemm epi %d' % epi)
    # both block tiles (the 64x64 one prefetches the gathered rows before its K loop, the 128x128 one stages them
    # after it), chosen per call: the library keeps no GEMM state
    for tile in (64, 128):
        out = torch.full((B, F), float('nan'), device=d_)
        ops.gemm(inp[:, :Kd], Wd[:, :Kd], out, trans_b=True, bias=b.to(d_), gat=WcT, gidx=sel, tile=tile)
        close(out, ref, what='gather gemm tile %d' % tile)
    # rows that are not one-/two-hot are counted
    inp[3, Kd:Kd + 3] = 1
    inp[5, Kd] = 0.5
    ops.cond_to_sel(inp[:, Kd:], sel, bad)
    assert int(bad) == 2


@pytest.mark.parametrize('B,C,F', [(500, 38, 256), (8192, 276, 335), (33, 5, 7)])
def test_hot_sort_and_row_sums_are_the_cond_block_of_the_weight_gradient(ops, B, C, F):
    g = torch.Generator().manual_seed(B)
    d_ = dev()
    dy = torch.randn(B, F, generator=g)
    sel = torch.stack([torch.randint(-1, C, (B,), generator=g), torch.randint(-1, C, (B,), generator=g)], 1).to(torch.int32)
    cond = torch.zeros(B, C, dtype=torch.float64)
    for h in range(2):
        ok = sel[:, h] >= 0
        cond[torch.arange(B)[ok], sel[ok, h].long()] += 1
    ref = dy.double().t() @ cond                 # [F, C] = the cond columns of dW = dY^T X
    acc = torch.zeros(C, F, device=d_)
    order = torch.full((2 * B,), -1, dtype=torch.int32, device=d_)
    sid = torch.full((2 * B,), -1, dtype=torch.int32, device=d_)
    n_items = torch.zeros(1, dtype=torch.int32, device=d_)
    ops.hot_sort(sel.to(d_), C, order, sid, n_items)
    m = int(n_items)
    assert m == int((sel >= 0).sum())
    so, ss = order.cpu()[:m].long(), sid.cpu()[:m].long()
    assert bool((ss[1:] >= ss[:-1]).all())                                  # ids ascending
    pairs = sorted((int(i), int(r)) for r in range(B) for i in sel[r].tolist() if i >= 0)
    assert sorted(zip(ss.tolist(), so.tolist())) == pairs                   # a permutation of the (id, row) pairs
    ops.hot_rows_sum(dy.to(d_), order, sid, n_items, acc)
    close(acc, ref.t(), what='scatter')
    dW = torch.full((F, C + 11), 7.0, device=d_)
    ops.transpose_slices([(acc, dW[:, 11:])])
    close(dW[:, 11:], ref, what='scatter transposed into dW')
    assert float(dW[:, :11].min()) == 7.0


@pytest.mark.parametrize('n,pac,C,d0', [(50, 10, 38, 128), (1024, 8, 276, 256), (7, 3, 5, 21), (64, 8, 18000, 64), (2100, 8, 276, 32)])
def test_discriminator_cond_block_as_gathers_and_sorted_row_sums(ops, n, pac, C, d0):
    """The cond columns of the discriminator's first layer (cat([data, cond]) per sub-row, pac sub-rows to a
    row: dp_cgan.py:66-68, 622-623) off the matrix cores.  Forward: dpc_disc_first_layer_act adds the hot rows
    of W1cT to the data-column products, == the dense product over the cond columns.  Weight gradient: the pac-
    grouped two-table sort + clip-scaled row sums == dY^T [cond_fake; cond_real] (C beyond the old one-block
    sort's 15360 ids and more pairs than the register path holds included)."""
    g = torch.Generator().manual_seed(n + C)
    d_ = dev()
    B = n * pac

    def ids():
        a = torch.randint(0, C, (B,), generator=g)
        b = (a + 1 + torch.randint(0, C - 1, (B,), generator=g)) % C
        sel = torch.stack([a, b], 1).to(torch.int32)
        sel[torch.rand(B, generator=g) < 0.1, 1] = -1           # one-hot sub-rows
        sel[0] = -1                                             # a sub-row without cond
        return sel
    sel_f, sel_r = ids(), ids()

    def dense(sel):                                             # [n, pac*C] two-hot rows
        x = torch.zeros(B, C, dtype=torch.float64)
        for h in range(2):
            ok = sel[:, h] >= 0
            x[torch.arange(B)[ok], sel[ok, h].long()] += 1
        return x.reshape(n, pac * C)
    Xf, Xr = dense(sel_f), dense(sel_r)
    # ---- forward
    W1c = torch.randn(d0, pac * C, generator=g) * 0.1           # cond columns of W1, sub-row major
    gat = W1c.t().contiguous().to(d_)
    zdata = torch.randn(2 * n, d0, generator=g)                 # stands for the data-column products
    bias = torch.randn(d0, generator=g) * 0.1
    alpha = torch.rand(n, generator=g)
    keep = (torch.rand(3 * n, d0, generator=g) > 0.5).to(torch.uint8)
    full = zdata.double() + torch.cat([Xf, Xr]) @ W1c.double().t()
    outs = []
    for stacked_gat in (False, True):
        z = (full.float() if not stacked_gat else zdata).clone().to(d_)
        H, S = torch.empty(3 * n, d0, device=d_), torch.empty(3 * n, d0, device=d_)
        kw = dict(gat=gat, sel_a=sel_f.to(d_), sel_b=sel_r.to(d_), pac=pac, C=C) if stacked_gat else {}
        ops.disc_first_layer_act(z, bias.to(d_), n, H, S, alpha_inj=alpha.to(d_), keep=keep.to(d_), **kw)
        outs.append((H, S, z))
    (H0, S0, _), (H1, S1, z1) = outs
    close(H1, H0, what='H with gathered cond columns')
    assert float((S1 != S0).float().mean()) < 1e-3              # slopes flip only where a pre-activation is within rounding of 0
    close(z1, full, what='products written back for the Gram kernel')
    # ---- weight gradient
    dy = torch.randn(2 * n, d0, generator=g)
    ks = torch.rand(2 * n, generator=g)
    ref = (dy * ks[:, None]).double().t() @ torch.cat([Xf, Xr])          # [d0, pac*C]
    order = torch.full((4 * B,), -1, dtype=torch.int32, device=d_)
    sid = torch.full((4 * B,), -1, dtype=torch.int32, device=d_)
    n_items = torch.zeros(1, dtype=torch.int32, device=d_)
    ops.hot_sort(sel_f.to(d_), C, order, sid, n_items, sel_b=sel_r.to(d_), pac=pac)
    m = int(n_items)
    assert m == int((sel_f >= 0).sum() + (sel_r >= 0).sum())
    ss = sid.cpu()[:m].long()
    assert bool((ss[1:] >= ss[:-1]).all()) and int(ss.max()) < pac * C
    acc = torch.zeros(pac * C, d0, device=d_)
    ops.hot_rows_sum(dy.to(d_), order, sid, n_items, acc, rowscale=ks.to(d_))
    close(acc, ref.t(), what='cond columns of dW1', rtol=2e-5)
    # transposed and ADDED into dW1's layout [d0, pac * (Dd + C)]
    Dd = 3
    row = Dd + C
    base = torch.randn(d0, pac * row, generator=g)
    dW = base.clone().to(d_)
    ops.transpose_slices([(acc[s_ * C:(s_ + 1) * C], dW[:, s_ * row + Dd:(s_ + 1) * row]) for s_ in range(pac)], accumulate=True)
    want = base.double().reshape(d0, pac, row).clone()
    want[:, :, Dd:] += ref.reshape(d0, pac, C)
    close(dW, want.reshape(d0, pac * row), what='accumulated into dW1', rtol=2e-5)


# ------------------------------------------------------------------------------------------------
# BatchNorm pieces
# ------------------------------------------------------------------------------------------------
def test_batchnorm_relu_forward_backward(ops):
    torch.manual_seed(0)
    B, F = 1000, 130
    z = (torch.randn(B, F) * 3 + 1.5)
    gamma, beta = torch.rand(F) + 0.5, torch.randn(F) * 0.1
    zz = z.clone().requires_grad_(True)
    gg, bb = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(F), torch.ones(F)
    y_ref = torch.relu(torch.nn.functional.batch_norm(zz, rm, rv, gg, bb, True, 0.1, 1e-5))
    dy = torch.randn(B, F)
    gz, ggam, gbet = torch.autograd.grad(y_ref, [zz, gg, bb], dy)
    d = dev()
    part = torch.empty(64 * 2 * F, dtype=torch.float64, device=d)
    mean, invstd = torch.empty(F, device=d), torch.empty(F, device=d)
    run_m, run_v = torch.zeros(F, device=d), torch.ones(F, device=d)
    zd = z.to(d)
    R = ops.colreduce_partial(0, zd, part)
    ops.colreduce_finish(part, R, 2, F, 0, mean, invstd, count=B, run_mean=run_m, run_var=run_v)
    y = torch.empty(B, F, device=d)
    ops.bn_relu_fwd(zd, y, mean, invstd, gamma.to(d), beta.to(d))
    close(y, y_ref.detach(), what='bn relu fwd')
    close(run_m, rm, what='running mean')
    close(run_v, rv, what='running var')
    dbeta, dgamma = torch.empty(F, device=d), torch.empty(F, device=d)
    R = ops.colreduce_partial(2, dy.to(d), part, y=y, z=zd, mean=mean, invstd=invstd)
    ops.colreduce_finish(part, R, 2, F, 2, dbeta, dgamma)
    close(dbeta, gbet, what='dbeta')
    close(dgamma, ggam, what='dgamma')
    dz = torch.empty(B, F, device=d)
    ops.bn_relu_bwd(dy.to(d), y, zd, mean, invstd, gamma.to(d), dbeta, dgamma, B, dz)
    close(dz, gz, what='dz')
    # the fused forms (reduction finish inside the BatchNorm kernel) are bit-identical to the two-launch ones
    mean2, invstd2 = torch.empty(F, device=d), torch.empty(F, device=d)
    run_m2, run_v2 = torch.zeros(F, device=d), torch.ones(F, device=d)
    y2 = torch.full((B, F), float('nan'), device=d)
    R = ops.colreduce_partial(0, zd, part)
    ops.bn_relu_fwd_fused(zd, y2, part, R, B, gamma.to(d), beta.to(d), mean2, invstd2, run_mean=run_m2, run_var=run_v2)
    assert torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(invstd2, invstd)
    assert torch.equal(run_m2, run_m) and torch.equal(run_v2, run_v)
    dbeta2, dgamma2 = torch.empty(F, device=d), torch.empty(F, device=d)
    dz2 = torch.full((B, F), float('nan'), device=d)
    R = ops.colreduce_partial(2, dy.to(d), part, y=y, z=zd, mean=mean, invstd=invstd)
    ops.bn_relu_bwd_fused(dy.to(d), y, zd, part, R, B, mean, invstd, gamma.to(d), dbeta2, dgamma2, dz2)
    assert torch.equal(dbeta2, dbeta) and torch.equal(dgamma2, dgamma)
    close(dz2, dz, rtol=1e-6, what='fused dz')        # same formula; FMA contraction may differ between the kernels


# ------------------------------------------------------------------------------------------------
# DP-SGD kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,d,K', [(7, 5, 13), (50, 32, 940), (64, 1, 24), (33, 256, 4888)])
def test_persample_gram_and_clip_factors(ops, n, d, K):
    g = torch.Generator().manual_seed(n + d + K)
    delta = torch.randn(3 * n, d, generator=g) * 0.1
    act = torch.randn(3 * n, K, generator=g)
    ref = torch.zeros(n, dtype=torch.float64)
    for i in range(n):
        G = sum(torch.outer(delta[p * n + i].double(), act[p * n + i].double()) for p in range(3))
        ref[i] = (G ** 2).sum() + ((delta[i] + delta[n + i]).double() ** 2).sum()
    norm2 = torch.full((n,), 3.0, dtype=torch.float64, device=dev())
    ops.persample_gram(delta.to(dev()), act.to(dev()), n, norm2, with_bias=True, accumulate=False)
    close(norm2, ref, what='gram')
    ops.persample_gram(delta.to(dev()), act.to(dev()), n, norm2, with_bias=False, accumulate=True)
    ref2 = ref * 2 - torch.stack([((delta[i] + delta[n + i]).double() ** 2).sum() for i in range(n)])
    close(norm2, ref2, what='gram accumulate')
    # unaligned views (odd leading dimension / offset): the scalar path
    big = torch.randn(3 * n, K + 3, generator=g).to(dev())
    av = big[:, 1:K + 1]
    ops.persample_gram(delta.to(dev()), av, n, norm2, with_bias=False, accumulate=False)
    a2 = av.cpu()
    ref3 = torch.stack([(sum(torch.outer(delta[p * n + i].double(), a2[p * n + i].double()) for p in range(3)) ** 2).sum()
                        for i in range(n)])
    close(norm2, ref3, what='gram unaligned')
    gscale, clip = 4.0, float(torch.sqrt(ref3).median() * 4.0)
    rowc = torch.empty(3 * n, device=dev())
    roww = torch.randn(3 * n, generator=g).to(dev())
    rowcw = torch.empty(3 * n, device=dev())
    nrm = torch.empty(n, device=dev())
    ops.clip_factors(norm2, clip, gscale, rowc, roww=roww, rowc_w=rowcw, norm_out=nrm)
    want_n = gscale * torch.sqrt(ref3)
    want_c = torch.clamp(clip / want_n, max=1.0).float()
    close(nrm, want_n, what='norms')
    close(rowc, want_c.repeat(3), what='clip factors')
    close(rowcw, want_c.repeat(3) * roww.cpu(), what='weighted clip factors')
    assert (want_c < 1).any() and (want_c == 1).any()


@pytest.mark.parametrize('n,d,K0', [(50, 256, 940), (9, 21, 37), (128, 64, 488)])
def test_first_layer_act_equals_interpolate_then_linear(ops, n, d, K0):
    """dpc_disc_first_layer_act on the plain products of the fake and real rows == the direct route
    (dpc_interpolate, then Linear + LeakyReLU + Dropout over all 3n rows through the GEMM epilogue),
    with injected alpha / keep bits and with Philox draws (same index maps)."""
    g = torch.Generator().manual_seed(n + d)
    d_ = dev()
    X = torch.zeros(3 * n, K0, device=d_)
    X[0:2 * n] = torch.randn(2 * n, K0, generator=g).to(d_)
    W = (torch.randn(d, K0, generator=g) / K0 ** 0.5).to(d_)
    b = (torch.randn(d, generator=g) * 0.1).to(d_)
    alpha = torch.rand(n, generator=g).to(d_)
    keep = (torch.rand(3 * n, d, generator=g) > 0.5).to(torch.uint8).to(d_)
    rng = ops.new_rng(77, d_)
    for inj in (True, False):
        kw = dict(keep=keep) if inj else {}
        ops.interpolate(X[n:2 * n], X[0:n], X[2 * n:], n, K0, alpha_inj=alpha if inj else None, rng=rng, site=ops.SITE_ALPHA)
        H_ref, S_ref = torch.empty(3 * n, d, device=d_), torch.empty(3 * n, d, device=d_)
        ops.gemm(X, W, H_ref, trans_b=True, bias=b, epi=ops.EPI_LEAKY_DROPOUT, aux_out=S_ref, rng=rng, site=ops.SITE_DROP0,
                 leak=0.2, **kw)
        Zp = torch.empty(2 * n, d, device=d_)
        ops.gemm(X[0:2 * n], W, Zp, trans_b=True)
        H, S = torch.full((3 * n, d), float('nan'), device=d_), torch.full((3 * n, d), float('nan'), device=d_)
        aout = torch.empty(n, device=d_)
        ops.disc_first_layer_act(Zp, b, n, H, S, alpha_inj=alpha if inj else None, rng=rng, site_alpha=ops.SITE_ALPHA,
                                 site_keep=ops.SITE_DROP0, leak=0.2, alpha_out=aout, **kw)
        if inj:
            assert torch.equal(aout, alpha)
        # the fake and real rows go through identical arithmetic up to the GEMM's summation order
        # (split-K or not): pre-activations within rounding of zero may pick the other slope
        pre = (X.double() @ W.double().t() + b.double()).cpu()
        sure = (pre.abs() > 1e-5 * pre.abs().max())
        assert torch.equal(S.cpu()[sure], S_ref.cpu()[sure])
        assert sure.float().mean() > 0.99
        close(H.cpu()[sure], H_ref.cpu()[sure], what='first layer H (inj %s)' % inj)
        # n plain rows (no interpolates), products cleared once read
        Z1 = Zp[0:n].clone()
        H1, S1 = torch.full((n, d), float('nan'), device=d_), torch.full((n, d), float('nan'), device=d_)
        ops.disc_first_layer_act(Z1, b, n, H1, S1, keep=keep[0:n] if inj else None, rng=rng, site_keep=ops.SITE_DROP0,
                                 leak=0.2, stacked=False, clear=True)
        assert torch.equal(H1, H[0:n]) and torch.equal(S1, S[0:n]) and float(Z1.abs().max()) == 0.0
        keep_bits = (S.cpu() != 0)
        if inj:
            assert torch.equal(keep_bits, keep.cpu().bool())
        else:
            kb = torch.empty(3 * n, d, dtype=torch.uint8, device=d_)
            ops.fill(kb, ops.FILL_KEEP, rng, ops.SITE_DROP0)
            assert torch.equal(keep_bits, kb.cpu().bool())


@pytest.mark.parametrize('n,d,K0', [(50, 256, 940), (7, 21, 37)])
def test_gp_gram_rows_equals_the_direct_penalty(ops, n, d, K0):
    """rowsum((u G) * u) with G = W W^T == ||u W||^2, and v1 = f (u G) S == ((f u W) W^T) S (dpc_gp_norm route)."""
    g = torch.Generator().manual_seed(n * d)
    d_ = dev()
    W = (torch.randn(d, K0, generator=g) / K0 ** 0.5)
    u = torch.randn(n, d, generator=g) * 0.7
    S = ((torch.rand(n, d, generator=g) > 0.5).float() * 2) * torch.where(torch.rand(n, d, generator=g) > 0.5, 1.0, 0.2)
    coef = 2.0 * 10.0 / n
    g0 = u.double() @ W.double()
    nrm = g0.norm(dim=1)
    f = coef * (nrm - 1) / nrm
    v1_ref = ((f[:, None] * g0) @ W.double().t()) * S.double()
    G = torch.empty(d, d, device=d_)
    ops.gemm(W.to(d_), W.to(d_), G, trans_b=True)
    T = torch.empty(n, d, device=d_)
    ops.gemm(u.to(d_), G, T)
    pen = torch.empty(n, dtype=torch.float64, device=d_)
    q = torch.empty(n, dtype=torch.float64, device=d_)
    fo, no = torch.empty(n, device=d_), torch.empty(n, device=d_)
    v1 = torch.empty(n, d, device=d_)
    ops.gp_gram_rows(T, u.to(d_), S.to(d_), coef, pen, fo, v1, q_out=q, nrm_out=no)
    close(no, nrm, what='gradient norms')
    close(q, nrm ** 2, what='squared norms')
    close(pen, (nrm - 1) ** 2, what='penalty rows')
    close(fo, f, what='f')
    close(v1, v1_ref, what='v1')


@pytest.mark.parametrize('n,dims', [(7, [(5, 13), (3, 5), (1, 3)]), (40, [(32, 940), (16, 32), (1, 16)]), (33, [(256, 4888), (1, 256)])])
def test_persample_clip_all_with_implicit_penalty_row(ops, n, dims):
    """dpc_persample_clip_all_gram == dpc_persample_clip_all on the same layers with the penalty row of
    layer 0's input materialised as v0 = f * (u W1)."""
    g = torch.Generator().manual_seed(n)
    d_ = dev()
    d0, K0 = dims[0]
    W = torch.randn(d0, K0, generator=g) / K0 ** 0.5
    X = torch.randn(2 * n, K0, generator=g)
    deltas = [torch.randn(3 * n, d, generator=g) * 0.1 for d, _ in dims]
    acts = [None] + [torch.randn(3 * n, K, generator=g) for _, K in dims[1:]]
    u = deltas[0][2 * n:]
    f = torch.randn(n, generator=g)
    g0 = u.double() @ W.double()
    q = (g0 ** 2).sum(1)
    full = torch.cat([X, (f.double()[:, None] * g0).float()], 0)
    zpre = (X.double() @ W.double().t()).float()
    roww = torch.randn(3 * n, generator=g).to(d_)

    def run(act0, **kw):
        layers = [(deltas[0].to(d_), act0.to(d_))] + [(dl.to(d_), ac.to(d_)) for dl, ac in zip(deltas[1:], acts[1:])]
        norm2 = torch.empty(n, dtype=torch.float64, device=d_)
        rowc, rowcw, nrm = torch.empty(3 * n, device=d_), torch.empty(3 * n, device=d_), torch.empty(n, device=d_)
        ops.persample_clip_all(layers, n, kw.pop('clip'), 3.0, norm2, rowc, roww=roww, rowc_w=rowcw, norm_out=nrm, **kw)
        return norm2, rowc, rowcw, nrm
    ref = run(full, clip=1e30)
    clip = float(ref[3].median())
    ref = run(full, clip=clip)
    fc = torch.empty(n, device=d_)
    got = run(X, clip=clip, zpre=zpre.to(d_), gp_f=f.to(d_), gp_q=q.to(d_), fc_out=fc)
    for a, b, what in zip(got, ref, ('norm2', 'rowc', 'rowc_w', 'norms')):
        close(a, b, what=what)
    close(fc, got[1][0:n].cpu() * f, what='fc')
    assert float(got[1].min()) < 1.0 and float(got[1].max()) == 1.0


def test_gemm_kscale_is_the_clip_scale_sum(ops):
    g = torch.Generator().manual_seed(21)
    for rows, d, K in [(150, 32, 940), (3072, 256, 611), (96, 24, 48)]:
        dl = torch.randn(rows, d, generator=g)
        x = torch.randn(rows, K, generator=g)
        c = torch.rand(rows, generator=g)
        ref = (dl.double() * c.double()[:, None]).t() @ x.double()
        out = torch.full((d, K), float('nan'), device=dev())
        ops.gemm(dl.to(dev()), x.to(dev()), out, trans_a=True, kscale=c.to(dev()))
        close(out, ref, what='kscale gemm %d' % rows)
    with pytest.raises(RuntimeError):
        ops.gemm(dl.to(dev()), dl.to(dev()), torch.empty(96, 96, device=dev()), trans_b=True, kscale=c.to(dev()))


@pytest.mark.parametrize('N,S_expected', [(5000, 4), (70001, 64)])
def test_row_select_superblocks_exact_on_synthetic_table(ops, N, S_expected):
    """k-th matching row through the superblock counts == numpy's nonzero()[k] == the full-scan path."""
    from dp_cgans_amd.data_sampler import DataSampler
    from dp_cgans_amd.synthetic import schema_span_info
    g = np.random.default_rng(N)
    span = schema_span_info('TINY')                      # 4 continuous + 5 categorical -> 23 binary discrete columns
    Dd = int(span[:, 0].sum())
    enc = np.zeros((N, Dd), dtype=np.float32)
    col = 0
    for j in range(4):
        m = 1 + j % 3
        enc[:, col] = g.uniform(-0.9, 0.9, N)
        enc[np.arange(N), col + 1 + g.integers(m, size=N)] = 1
        col += 1 + m
    for j in range(5):
        k = [2, 3, 4, 6, 8][j]
        p = 1.0 / np.arange(1, k + 1)
        codes = g.choice(k, size=N, p=p / p.sum())
        for v in range(k):
            enc[:, col] = codes == v
            enc[:, col + 1] = codes != v
            col += 2
    ds = DataSampler(enc, span, device=dev(), seed=1)
    assert ds.cntblk is not None and ds.cntblk.shape[2] == S_expected
    assert torch.equal(ds.cntblk.sum(2, dtype=torch.int64).int(), ds.cnt)
    onehot = enc[:, ds.cat_col.cpu().numpy()] != 0
    assert np.array_equal(ds.cnt.cpu().numpy(), (onehot.T.astype(np.int64) @ onehot.astype(np.int64)).astype(np.int32))
    B = 4096
    cond, mask, pid, vid = ds.sample_condvec_pair(B)
    sel, gcount = ds._last_sel[0].cpu().numpy(), ds._last_sel[1].cpu().numpy()
    u = g.random(B)
    perm = g.permutation(B)
    gp = gcount[perm]                                    # request b asks for a row of group perm[b]
    k = np.minimum((u * gp).astype(np.int64), gp - 1)
    k[0], k[1] = 0, gp[1] - 1                            # first and last member of a group
    rid, rows = ds.rows_for(perm, k=k)
    for b in range(B):
        ca, cb = sel[perm[b]]
        match = np.nonzero(onehot[:, ca] & onehot[:, cb])[0]
        assert len(match) == gcount[perm[b]] and rid[b] == match[k[b]], b
    assert np.array_equal(rows, enc[rid])
    full = torch.empty(B, dtype=torch.int64, device=dev())
    ops.row_select(ds.bitmaps, ds._last_sel[0], ds._last_sel[1], full, perm=torch.from_numpy(perm).to(dev()),
                   k_inj=torch.from_numpy(k).to(dev()))
    assert np.array_equal(full.cpu().numpy(), rid)
    bad = torch.full((3,), 7, dtype=torch.int64, device=dev())
    ops.row_select(ds.bitmaps, ds._last_sel[0], ds._last_sel[1], bad, k_inj=torch.tensor([-1, 1 << 40, 0], device=dev()),
                   cntblk=ds.cntblk)
    assert bad[0].item() == -1 and bad[1].item() == -1 and bad[2].item() >= 0


@pytest.mark.parametrize('B', [1, 2, 5, 64, 100, 8192, 10007])
def test_random_perm_is_a_permutation_and_looks_uniform(ops, B):
    """Feistel / cycle-walking permutation: a bijection of 0..B-1 for every size; for small B the value
    at each position is uniform over many steps (chi-square against the uniform law)."""
    rng = ops.new_rng(123, dev())
    perm = torch.empty(B, dtype=torch.int64, device=dev())
    ops.random_perm(rng, perm, None)
    assert torch.equal(torch.sort(perm).values, torch.arange(B, device=dev()))
    if B == 64:
        T = 4000
        counts = np.zeros((B, B))
        fixed = 0
        for _ in range(T):
            ops.state_advance(rng)
            ops.random_perm(rng, perm, None)
            p = perm.cpu().numpy()
            counts[np.arange(B), p] += 1
            fixed += int((p == np.arange(B)).sum())
        chi2 = ((counts - T / B) ** 2 / (T / B)).sum()
        dof = (B - 1) ** 2
        assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), chi2          # within 6 sigma of the chi-square mean
        assert abs(fixed / T - 1.0) < 0.15                             # a uniform permutation has 1 fixed point on average


def test_cooccurrence_tiles_and_bitmaps_on_a_wide_table(ops):
    """C = 330 categories (6 x 6 tiles of the AND/popcount kernel, mirrored off-diagonal tiles, a
    ragged last tile), N not a multiple of 64: bitmaps, counts and superblock counts against numpy."""
    g = np.random.default_rng(5)
    N, ncol, k = 3001, 55, 3
    codes = g.integers(k, size=(N, ncol))
    onehot = np.zeros((N, ncol * k * 2), dtype=np.float32)          # per category v: (is v, is not v)
    for j in range(ncol):
        for v in range(k):
            onehot[:, (j * k + v) * 2] = codes[:, j] == v
            onehot[:, (j * k + v) * 2 + 1] = codes[:, j] != v
    C = onehot.shape[1]
    assert C == 330
    W = (N + 63) // 64
    table = torch.from_numpy(onehot).to(dev())
    cat_col = torch.arange(C, dtype=torch.int32, device=dev())
    bitmaps = torch.zeros(C, W, dtype=torch.int64, device=dev())
    ops.build_bitmaps(table, cat_col, bitmaps)
    bits = np.unpackbits(bitmaps.cpu().numpy().view(np.uint8).reshape(C, W * 8), axis=1, bitorder='little')[:, :N]
    assert np.array_equal(bits.T.astype(np.float32), onehot)
    pad = np.unpackbits(bitmaps.cpu().numpy().view(np.uint8).reshape(C, W * 8), axis=1, bitorder='little')[:, N:]
    assert pad.sum() == 0
    S = 5
    cnt = torch.full((C, C), -7, dtype=torch.int32, device=dev())
    cntblk = torch.full((C, C, S), -7, dtype=torch.int32, device=dev())
    ops.cooccurrence(bitmaps, cnt, cntblk)
    ref = (onehot.T.astype(np.int64) @ onehot.astype(np.int64)).astype(np.int32)
    assert np.array_equal(cnt.cpu().numpy(), ref)
    Wb = -(-W // S)
    for sb in range(S):
        rows = slice(sb * Wb * 64, min(N, (sb + 1) * Wb * 64))
        part = (onehot[rows].T.astype(np.int64) @ onehot[rows].astype(np.int64)).astype(np.int32)
        assert np.array_equal(cntblk[:, :, sb].cpu().numpy(), part), sb
    cnt1 = torch.empty(C, C, dtype=torch.int32, device=dev())
    ops.cooccurrence(bitmaps, cnt1)                                   # S = 1, no superblock table
    assert torch.equal(cnt1, cnt)
