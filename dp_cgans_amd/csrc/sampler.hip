// DataSampler on the device (reference: data_sampler.py:9-198 tables, :245-343 condvec pairs,
// :382-396 real-row pick) and the mode-specific encode (data_transformer.py:102-150).
//
// The reference materialises, per column pair and joint value, the list of matching row ids
// (O(n_pairs * N) memory: 75 GB at 1 M rows x 9453 pairs).  Here the tables are
//   bitmaps[C][W]  one N-bit row set per category (C = length of the cond vector, W = ceil(N/64))
//   cnt[C][C]      co-occurrence counts  cnt[a][b] = popcount(bitmaps[a] & bitmaps[b])
// and everything else is derived on the fly: the joint-value probabilities of a pair are a
// d1 x d2 block of cnt, walked in the order numpy's unique() gives the packed codes (descending
// value of the first column, then of the second), and "the k-th matching row" is the k-th set bit
// of the AND of two bitmaps (popcount select).  All integer work; HBM/L2-bound.
#include "dpc_common.h"

// ---------------------------------------------------------------------------------------------
// table construction
// ---------------------------------------------------------------------------------------------
// One block per 64-row word of the bitmaps.  Lane l of every wave owns category c0 + l of the
// current chunk of 64 categories and walks its wave's 16 rows, so each row read is one coalesced
// 256-byte segment when the categories' columns are adjacent (they are: one-hot spans); the four
// 16-bit partial masks of a category meet in LDS.  cat_col[a] = column of the encoded matrix that
// holds category a.
__global__ __launch_bounds__(256) void build_bitmaps_kernel(const float* __restrict__ table, int64_t ld, int64_t N,
                                                            const int32_t* __restrict__ cat_col, int C,
                                                            uint64_t* __restrict__ bitmaps, int64_t W) {
  __shared__ unsigned short part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t w = blockIdx.x;
  const int64_t row0 = w * 64 + wave * 16;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int a = c0 + lane;
    const int col = a < C ? cat_col[a] : 0;
    unsigned m = 0;
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      int64_t row = row0 + rr;
      bool bit = (a < C) && (row < N) && (table[row * ld + col] != 0.f);
      m |= (bit ? 1u : 0u) << rr;
    }
    part[wave][lane] = (unsigned short)m;
    __syncthreads();
    if (wave == 0 && a < C) {
      uint64_t word = (uint64_t)part[0][lane] | ((uint64_t)part[1][lane] << 16) | ((uint64_t)part[2][lane] << 32) |
                      ((uint64_t)part[3][lane] << 48);
      bitmaps[(int64_t)a * W + w] = word;
    }
    __syncthreads();
  }
}

extern "C" int dpc_build_bitmaps(const float* table, int64_t ld, int64_t N, const int32_t* cat_col, int C,
                                 uint64_t* bitmaps, int64_t W, void* stream) {
  DPC_CHECK_ARG(table && cat_col && bitmaps, "build_bitmaps: null pointer");
  DPC_CHECK_ARG(N > 0 && C > 0 && W == (N + 63) / 64, "build_bitmaps: W must be ceil(N/64)");
  hipLaunchKernelGGL(build_bitmaps_kernel, dim3((unsigned)W), dim3(256), 0, (hipStream_t)stream, table, ld, N, cat_col, C,
                     bitmaps, W);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// cnt[a][b] = popcount(bitmaps[a] & bitmaps[b]) (and, when cntblk != NULL, the same count split over
// S superblocks of Wb = ceil(W/S) bitmap words).  A GEMM-shaped problem over the AND/popcount
// "semiring": block (ta, tb, sb) owns the 64 x 64 category pairs of tile (ta, tb) for superblock sb,
// stages 64 + 64 bitmap rows x 32 words through LDS and keeps a 4 x 4 block of counts per thread,
// so every bitmap word is fetched once per tile row/column instead of once per pair (C = 18 000 on
// the wide table: 3.2e8 pairs).  Only tiles with tb >= ta are computed; the result is mirrored.
// Integer atomics into cnt (zeroed here): order-independent.
__global__ __launch_bounds__(256) void cooccurrence_kernel(const uint64_t* __restrict__ bitmaps, int64_t W, int C,
                                                           int32_t* __restrict__ cnt, int32_t* __restrict__ cntblk,
                                                           int S, int64_t Wb) {
  constexpr int T = 64, WT = 32;
  __shared__ uint64_t As[T][WT + 1];
  __shared__ uint64_t Bs[T][WT + 1];
  const int ta = blockIdx.x, tb = blockIdx.y, sb = blockIdx.z;
  if (tb < ta) return;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t w_begin = (int64_t)sb * Wb, w_end = min(W, w_begin + Wb);
  int acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (int64_t w0 = w_begin; w0 < w_end; w0 += WT) {
#pragma unroll
    for (int e = threadIdx.x; e < T * WT; e += 256) {
      int r = e / WT, c = e % WT;
      int64_t w = w0 + c;
      int ra = ta * T + r, rb = tb * T + r;
      As[r][c] = (ra < C && w < w_end) ? bitmaps[(int64_t)ra * W + w] : 0ull;
      Bs[r][c] = (rb < C && w < w_end) ? bitmaps[(int64_t)rb * W + w] : 0ull;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < WT; ++c) {
      uint64_t av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[ty * 4 + i][c];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[tx * 4 + j][c];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += __popcll(av[i] & bv[j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int a = ta * T + ty * 4 + i, b = tb * T + tx * 4 + j;
      if (a < C && b < C) {
        int v = acc[i][j];
        if (cntblk) {
          cntblk[((int64_t)a * C + b) * S + sb] = v;
          cntblk[((int64_t)b * C + a) * S + sb] = v;
        }
        if (v) {
          atomicAdd(&cnt[(int64_t)a * C + b], v);
          if (ta != tb) atomicAdd(&cnt[(int64_t)b * C + a], v);
        }
      }
    }
  }
}

extern "C" int dpc_cooccurrence(const uint64_t* bitmaps, int64_t W, int C, int32_t* cnt, int32_t* cntblk, int S,
                                void* stream) {
  DPC_CHECK_ARG(bitmaps && cnt && W > 0 && C > 0, "cooccurrence: bad argument");
  DPC_CHECK_ARG(S >= 1 && S <= 64 && (cntblk || S == 1), "cooccurrence: 1 <= S <= 64, S > 1 needs cntblk");
  const int64_t nt = dpc_cdiv(C, 64);
  DPC_CHECK_ARG(nt <= 65535, "cooccurrence: more than 65535 x 64 categories");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(cnt, 0, (size_t)C * C * sizeof(int32_t), st);
  if (e != hipSuccess) { dpc_set_error(hipGetErrorString(e)); return DPC_EHIP; }
  hipLaunchKernelGGL(cooccurrence_kernel, dim3((unsigned)nt, (unsigned)nt, (unsigned)S), dim3(256), 0, st, bitmaps, W,
                     C, cnt, cntblk, S, dpc_cdiv(W, S));
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// sample_condvec_pair: one wave per sample.
// ---------------------------------------------------------------------------------------------
typedef float f32x4s __attribute__((ext_vector_type(4)));

struct CondvecArgs {
  const int32_t* cnt; const int32_t* cond_st; const int32_t* dims; int nd, C; int64_t N;
  const int32_t* pairs_inj; const double* r_inj; const uint64_t* rng; uint32_t site; int64_t B;
  float* cond0; int64_t ld0; float* cond1; int64_t ld1; float* cond2; int64_t ld2;
  int32_t* mask; int64_t ldm; int64_t* pair_id; int64_t* val_idx; int32_t* sel; int64_t* gcount;
};

// The walk over the pair's joint values is sequential by definition (an fp64 running sum in cumsum's order), but
// its LOADS are not: the wave fetches 64 entries of the d1 x d2 count block at a time, every lane converts its own
// entry, and the running sum then takes the positive entries lane by lane out of registers (ballot + readlane) -
// one L2 round trip per 64 entries instead of one per entry (lane 0 used to walk alone, load by load).
__device__ __forceinline__ double condvec_readlane_f64(double v, int l) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

__device__ __forceinline__ void condvec_pair_wave(const CondvecArgs& a, const int64_t b, const int lane) {
  int i, j;
  double r;
  // draws 4b .. 4b+3 of the site are ONE Philox block (dpc_draw_u32 would compute it four times over)
  DpcPhilox4 ph = {{0u, 0u, 0u, 0u}};
  if (!a.pairs_inj || !a.r_inj) ph = dpc_philox(a.rng[0], a.rng[1], a.site, (uint64_t)b);
  if (a.pairs_inj) {
    i = a.pairs_inj[2 * b];
    j = a.pairs_inj[2 * b + 1];
  } else {
    // ordered draw of two distinct columns (np.random.choice(nd, 2, replace=False) in distribution)
    float u1 = dpc_u32_to_unit(ph.v[0]);
    float u2 = dpc_u32_to_unit(ph.v[1]);
    i = min(a.nd - 1, (int)(u1 * a.nd));
    j = min(a.nd - 2, (int)(u2 * (a.nd - 1)));
    if (j >= i) ++j;
  }
  if (a.r_inj) r = a.r_inj[b];
  else r = dpc_u64_to_unit(ph.v[2], ph.v[3]);
  const int lo = min(i, j), hi = max(i, j);
  const int ka = a.dims[lo], kb = a.dims[hi], sa = a.cond_st[lo], sb = a.cond_st[hi];
  // the pair's joint values in ascending packed-code order = (va desc, vb desc): entry p is
  // (va, vb) = (ka-1 - p / kb, kb-1 - p % kb); fp64 running sum of count/N in that order, exactly as cumsum over
  // counts/sum(counts) does
  double cum = 0.0;
  const double dn = (double)a.N;
  const int total = ka * kb;
  int found = 0, idx = 0, have_first = 0, fp = 0, fcnt = 0, sp = 0, sidx = 0, scnt = 0;
  for (int base = 0; base < total && !found; base += 64) {
    const int p = base + lane;
    int c = 0;
    if (p < total) {
      const int qa = p / kb;
      c = a.cnt[(int64_t)(sa + ka - 1 - qa) * a.C + sb + (kb - 1 - (p - qa * kb))];
    }
    const double t = (double)c / dn;
    uint64_t live = __ballot(c > 0);
    while (live) {
      const int l = __builtin_ctzll(live);
      live &= live - 1;
      const int cl = __builtin_amdgcn_readlane(c, l);
      if (!have_first) { have_first = 1; fp = base + l; fcnt = cl; }
      cum += condvec_readlane_f64(t, l);
      if (cum > r) { found = 1; sp = base + l; sidx = idx; scnt = cl; break; }
      ++idx;
    }
  }
  if (!found) { sp = fp; sidx = 0; scnt = fcnt; }   // argmax of an all-False row is 0
  const int qa = sp / kb;
  const int ca = sa + ka - 1 - qa, cb = sb + kb - 1 - (sp - qa * kb);
  if (lane == 0) {
    a.pair_id[b] = (int64_t)lo * a.nd - (int64_t)lo * (lo + 1) / 2 + (hi - lo - 1);
    a.val_idx[b] = sidx;
    if (a.sel) { a.sel[2 * b] = ca; a.sel[2 * b + 1] = cb; }
    if (a.gcount) a.gcount[b] = scnt;
  }
  if (a.cond0 || a.cond1 || a.cond2)        // (a caller that carries the ids in `sel` asks for no dense copy at all)
  for (int c = lane; c < a.C; c += 64) {
    float v = (c == ca || c == cb) ? 1.f : 0.f;
    if (a.cond0) a.cond0[b * a.ld0 + c] = v;
    if (a.cond1) a.cond1[b * a.ld1 + c] = v;
    if (a.cond2) a.cond2[b * a.ld2 + c] = v;
  }
  if (a.mask)
    for (int d = lane; d < a.nd; d += 64) a.mask[b * a.ldm + d] = (d == lo || d == hi) ? 1 : 0;
}

__global__ __launch_bounds__(256) void condvec_pair_kernel(CondvecArgs a) {
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;
  condvec_pair_wave(a, b, threadIdx.x & 63);
}

extern "C" int dpc_condvec_pair_sample(const int32_t* cnt, const int32_t* cond_st, const int32_t* dims, int nd, int C,
                                       int64_t N, const int32_t* pairs_inj, const double* r_inj, const uint64_t* rng,
                                       uint32_t site, int64_t B, float* cond0, int64_t ld0, float* cond1, int64_t ld1,
                                       float* cond2, int64_t ld2, int32_t* mask, int64_t ldm, int64_t* pair_id,
                                       int64_t* val_idx, int32_t* sel, int64_t* gcount, void* stream) {
  DPC_CHECK_ARG(cnt && cond_st && dims && pair_id && val_idx, "condvec_pair_sample: null pointer");
  DPC_CHECK_ARG(nd >= 2 && C >= 2 && N > 0 && B >= 0, "condvec_pair_sample: needs >= 2 discrete columns");
  DPC_CHECK_ARG((pairs_inj && r_inj) || rng, "condvec_pair_sample: no random source");
  if (B == 0) return DPC_OK;
  CondvecArgs a;
  a.cnt = cnt; a.cond_st = cond_st; a.dims = dims; a.nd = nd; a.C = C; a.N = N; a.pairs_inj = pairs_inj;
  a.r_inj = r_inj; a.rng = rng; a.site = site; a.B = B; a.cond0 = cond0; a.ld0 = ld0; a.cond1 = cond1; a.ld1 = ld1;
  a.cond2 = cond2; a.ld2 = ld2; a.mask = mask; a.ldm = ldm; a.pair_id = pair_id; a.val_idx = val_idx; a.sel = sel;
  a.gcount = gcount;
  hipLaunchKernelGGL(condvec_pair_kernel, dim3((unsigned)dpc_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// The inputs of one network update in ONE launch: the cond vectors (condvec_pair_wave above), the latent batch
// z ~ N(0,1) (the same draws as dpc_fill kind 1: normals 4q..4q+3 of `site_z` are one Philox block) and the zero
// fill of the update's accumulator pool.  Three independent jobs that used to be three kernel nodes in a row at
// the top of the step's critical path; here they are block ranges of one grid, the longest job (the cond
// vectors) first.
// ---------------------------------------------------------------------------------------------
struct InputsArgs {
  CondvecArgs cv;
  float* z; int64_t ldz, zrows, zcols; uint32_t site_z;
  f32x4s* zero; int64_t zero_n16;
  int nb_cv, nb_z;
};

__global__ __launch_bounds__(256) void step_inputs_kernel(InputsArgs a) {
  const int bid = blockIdx.x;
  if (bid < a.nb_cv) {
    const int64_t b = (int64_t)bid * 4 + (threadIdx.x >> 6);
    if (b < a.cv.B) condvec_pair_wave(a.cv, b, threadIdx.x & 63);
  } else if (bid < a.nb_cv + a.nb_z) {
    const int64_t n = a.zrows * a.zcols, ngrp = (n + 3) / 4;
    for (int64_t q = (int64_t)(bid - a.nb_cv) * 256 + threadIdx.x; q < ngrp; q += (int64_t)a.nb_z * 256) {
      float z[4];
      dpc_normal4(a.cv.rng, a.site_z, (uint64_t)q, z);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t i = q * 4 + j;
        if (i < n) {
          const int64_t r = i / a.zcols, c = i - r * a.zcols;
          a.z[r * a.ldz + c] = 0.f + 1.f * z[j];
        }
      }
    }
  } else {
    const int64_t nb = (int64_t)gridDim.x - a.nb_cv - a.nb_z;
    const f32x4s zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)(bid - a.nb_cv - a.nb_z) * 256 + threadIdx.x; i < a.zero_n16; i += nb * 256) a.zero[i] = zero4;
  }
}

extern "C" int dpc_step_inputs(const int32_t* cnt, const int32_t* cond_st, const int32_t* dims, int nd, int C, int64_t N,
                               const uint64_t* rng, uint32_t site, int64_t B, float* cond0, int64_t ld0, float* cond1,
                               int64_t ld1, float* cond2, int64_t ld2, int32_t* mask, int64_t ldm, int64_t* pair_id,
                               int64_t* val_idx, int32_t* sel, int64_t* gcount, float* z, int64_t ldz, int64_t zrows,
                               int64_t zcols, uint32_t site_z, void* zero, int64_t zero_bytes, void* stream) {
  DPC_CHECK_ARG(cnt && cond_st && dims && pair_id && val_idx && rng, "step_inputs: null pointer");
  DPC_CHECK_ARG(nd >= 2 && C >= 2 && N > 0 && B > 0, "step_inputs: needs >= 2 discrete columns and a batch");
  DPC_CHECK_ARG(!z || (zrows > 0 && zcols > 0 && ldz >= zcols), "step_inputs: bad latent shape");
  DPC_CHECK_ARG(zero_bytes >= 0 && zero_bytes % 16 == 0 && (zero || zero_bytes == 0) && (((uintptr_t)zero & 15) == 0),
                "step_inputs: the zero-filled range must be 16-byte aligned and a multiple of 16 bytes");
  InputsArgs a;
  a.cv.cnt = cnt; a.cv.cond_st = cond_st; a.cv.dims = dims; a.cv.nd = nd; a.cv.C = C; a.cv.N = N; a.cv.pairs_inj = nullptr;
  a.cv.r_inj = nullptr; a.cv.rng = rng; a.cv.site = site; a.cv.B = B; a.cv.cond0 = cond0; a.cv.ld0 = ld0; a.cv.cond1 = cond1;
  a.cv.ld1 = ld1; a.cv.cond2 = cond2; a.cv.ld2 = ld2; a.cv.mask = mask; a.cv.ldm = ldm; a.cv.pair_id = pair_id;
  a.cv.val_idx = val_idx; a.cv.sel = sel; a.cv.gcount = gcount;
  a.z = z; a.ldz = ldz; a.zrows = z ? zrows : 0; a.zcols = z ? zcols : 1; a.site_z = site_z;
  a.zero = (f32x4s*)zero; a.zero_n16 = zero_bytes / 16;
  a.nb_cv = (int)dpc_cdiv(B, 4);
  a.nb_z = z ? dpc_grid((zrows * zcols + 3) / 4, 256) : 0;
  const int nb_zero = a.zero_n16 ? dpc_grid(dpc_cdiv(a.zero_n16, 4), 256) : 0;       // ~4 x 16 bytes per thread
  hipLaunchKernelGGL(step_inputs_kernel, dim3((unsigned)(a.nb_cv + a.nb_z + nb_zero)), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// sample_data_pair: row_id[b] = k-th (0-based) set bit of bitmaps[ca] & bitmaps[cb], where
// (ca, cb) = sel[src], src = perm ? perm[b] : b.  One wave per request: the superblock that holds
// the k-th match comes from a wave-wide scan of the pair's S superblock counts (cntblk), then only
// that superblock's Wb words of the two bitmaps are scanned, 64 words per step (popcount, wave
// prefix sum, ballot).  Reads ~2*Wb*8 + 4*S bytes per request instead of half of both bitmaps.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int dpc_wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

__global__ __launch_bounds__(256) void row_select_kernel(const uint64_t* __restrict__ bitmaps, int64_t W,
                                                         const int32_t* __restrict__ sel,
                                                         const int64_t* __restrict__ gcount,
                                                         const int64_t* __restrict__ perm,
                                                         const int64_t* __restrict__ k_inj, const uint64_t* rng,
                                                         uint32_t site, int64_t B, int64_t* __restrict__ row_id,
                                                         const int32_t* __restrict__ cntblk, int C, int S, int64_t Wb) {
  const int lane = threadIdx.x & 63;
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * 4) {
    int64_t src = perm ? perm[b] : b;
    const int ca = sel[2 * src], cb = sel[2 * src + 1];
    const uint64_t* pa = bitmaps + (int64_t)ca * W;
    const uint64_t* pb = bitmaps + (int64_t)cb * W;
    int64_t cntg = gcount[src];
    int64_t k;
    if (k_inj) k = k_inj[b];
    else {
      double u = dpc_u64_to_unit(dpc_draw_u32(rng, site, 2 * (uint64_t)b), dpc_draw_u32(rng, site, 2 * (uint64_t)b + 1));
      k = (int64_t)(u * (double)cntg);
      if (k > cntg - 1) k = cntg - 1;
    }
    if (k < 0 || k >= cntg) {           // empty group or bad injected rank: defined result, no scan
      if (lane == 0) row_id[b] = -1;
      continue;
    }
    int64_t w0 = 0, w1 = W;
    if (cntblk) {
      int c = lane < S ? cntblk[((int64_t)ca * C + cb) * S + lane] : 0;
      int incl = dpc_wave_incl_scan(c, lane);
      unsigned long long m = __ballot((int64_t)incl > k);      // non-empty: the S counts sum to cntg > k
      int sb = __ffsll((long long)m) - 1;
      int excl = __shfl(incl - c, sb, 64);
      k -= excl;
      w0 = (int64_t)sb * Wb;
      w1 = min(W, w0 + Wb);
    }
    int64_t found = -1;
    for (int64_t w = w0; w < w1; w += 64) {
      uint64_t x = (w + lane < w1) ? (pa[w + lane] & pb[w + lane]) : 0ull;
      int c = __popcll(x);
      int incl = dpc_wave_incl_scan(c, lane);
      unsigned long long m = __ballot((int64_t)incl > k);
      if (m) {
        int l = __ffsll((long long)m) - 1;
        if (lane == l) {
          int kk = (int)(k - (incl - c));
          uint64_t wv = x;
          for (int i = 0; i < kk; ++i) wv &= wv - 1;
          found = (w + lane) * 64 + (__ffsll((long long)wv) - 1);
          row_id[b] = found;
        }
        break;                           // wave-uniform
      }
      k -= __shfl(incl, 63, 64);
    }
  }
}

extern "C" int dpc_row_select(const uint64_t* bitmaps, int64_t W, const int32_t* sel, const int64_t* gcount,
                              const int64_t* perm, const int64_t* k_inj, const uint64_t* rng, uint32_t site, int64_t B,
                              int64_t* row_id, const int32_t* cntblk, int C, int S, void* stream) {
  DPC_CHECK_ARG(bitmaps && sel && gcount && row_id, "row_select: null pointer");
  DPC_CHECK_ARG(k_inj || rng, "row_select: no random source");
  DPC_CHECK_ARG(W > 0 && B >= 0, "row_select: bad shape");
  DPC_CHECK_ARG(!cntblk || (C > 0 && S >= 1 && S <= 64), "row_select: superblock table needs C and 1 <= S <= 64");
  if (B == 0) return DPC_OK;
  int64_t blocks = dpc_cdiv(B, 4);
  int grid = (int)(blocks < 8192 ? blocks : 8192);
  hipLaunchKernelGGL(row_select_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bitmaps, W, sel, gcount, perm,
                     k_inj, rng, site, B, row_id, cntblk, C, S, cntblk ? dpc_cdiv(W, S) : W);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Tables without discrete columns: row_id[b] = floor(u_b * N), u from u_inj or Philox
// (np.random.randint(len(data), size=n), data_sampler.py:388-390, in distribution).
__global__ __launch_bounds__(256) void row_uniform_kernel(int64_t N, const double* __restrict__ u_inj,
                                                          const uint64_t* rng, uint32_t site, int64_t B,
                                                          int64_t* __restrict__ row_id) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double u = u_inj ? u_inj[b]
                   : dpc_u64_to_unit(dpc_draw_u32(rng, site, 2 * (uint64_t)b), dpc_draw_u32(rng, site, 2 * (uint64_t)b + 1));
  int64_t r = (int64_t)(u * (double)N);
  row_id[b] = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
}

extern "C" int dpc_row_uniform(int64_t N, const double* u_inj, const uint64_t* rng, uint32_t site, int64_t B,
                               int64_t* row_id, void* stream) {
  DPC_CHECK_ARG(row_id && N > 0 && B >= 0, "row_uniform: bad argument");
  DPC_CHECK_ARG(u_inj || rng, "row_uniform: no random source");
  if (B == 0) return DPC_OK;
  hipLaunchKernelGGL(row_uniform_kernel, dim3((unsigned)dpc_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, N, u_inj,
                     rng, site, B, row_id);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// out[b, 0:Dd] = table[row_id[b], 0:Dd]; optionally out[b, Dd + c] = 1 at the two categories of
// request src = perm ? perm[b] : b (the cond vector that travels with the real row, c1[perm]).
__global__ __launch_bounds__(256) void row_gather_kernel(const float* __restrict__ table, int64_t ld,
                                                         const int64_t* __restrict__ row_id, int64_t B, int Dd,
                                                         float* __restrict__ out, int64_t ldo,
                                                         const int32_t* __restrict__ sel,
                                                         const int64_t* __restrict__ perm, int C,
                                                         int32_t* __restrict__ sel_out) {
  const int lane = threadIdx.x & 63;
  int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  int64_t rid = row_id[b];
  const float* src = table + rid * ld;
  float* dst = out + b * ldo;
  for (int j = lane; j < Dd; j += 64) dst[j] = rid >= 0 ? src[j] : 0.f;
  if (sel) {
    int64_t s = perm ? perm[b] : b;
    int ca = sel[2 * s], cb = sel[2 * s + 1];
    for (int c = lane; c < C; c += 64) dst[Dd + c] = (c == ca || c == cb) ? 1.f : 0.f;
    if (sel_out && lane == 0) {              // the cond vector that travels with the row, as its two hot ids
      sel_out[2 * b] = ca;
      sel_out[2 * b + 1] = cb;
    }
  }
}

extern "C" int dpc_row_gather(const float* table, int64_t ld, const int64_t* row_id, int64_t B, int Dd, float* out,
                              int64_t ldo, const int32_t* sel, const int64_t* perm, int C, int32_t* sel_out,
                              void* stream) {
  DPC_CHECK_ARG(table && row_id && out, "row_gather: null pointer");
  DPC_CHECK_ARG(!sel_out || sel, "row_gather: sel_out needs sel");
  DPC_CHECK_ARG(B >= 0 && Dd > 0 && ld >= Dd && ldo >= Dd + (sel ? C : 0), "row_gather: bad shape");
  if (B == 0) return DPC_OK;
  hipLaunchKernelGGL(row_gather_kernel, dim3((unsigned)dpc_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, table, ld,
                     row_id, B, Dd, out, ldo, sel, perm, C, sel_out);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Random permutation of 0..B-1 (np.random.shuffle(arange(B)), dp_cgan.py:594-595, in distribution)
// without sorting: perm[i] = P(i) for a keyed bijection P of [0, 2^bits) restricted to [0, B) by
// cycle walking.  P is a 6-round Feistel network over the (a | b)-bit halves of i whose round
// functions are Philox words of this (seed, step, site); every element is independent (one thread
// each, O(B) work, one launch; ranking 64-bit keys cost O(B^2) compares and three launches).
// Cycle walking ends: the orbit of i under P returns to i < B.
__device__ __forceinline__ uint32_t dpc_feistel(uint32_t x, int abits, int bbits, const uint64_t* rng, uint32_t site) {
  const uint32_t amask = (1u << abits) - 1u, bmask = (1u << bbits) - 1u;
  uint32_t L = x >> bbits, R = x & bmask;                 // L: abits, R: bbits
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    // round r mixes the half that is currently R into L; halves swap (and trade widths) every round
    uint32_t f = dpc_draw_u32(rng, site, ((uint64_t)r << 34) | ((uint64_t)R << 2));
    if ((r & 1) == 0) {
      uint32_t nl = R, nr = (L ^ f) & amask;               // new halves: (bbits | abits)
      L = nl; R = nr;
    } else {
      uint32_t nl = R, nr = (L ^ f) & bmask;               // back to (abits | bbits)
      L = nl; R = nr;
    }
  }
  return (L << bbits) | R;                                  // 6 rounds: widths are (abits | bbits) again
}

__global__ __launch_bounds__(256) void perm_feistel_kernel(const uint64_t* rng, uint32_t site, int64_t B, int abits,
                                                           int bbits, int64_t* __restrict__ perm) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  uint32_t x = (uint32_t)i;
  do {
    x = dpc_feistel(x, abits, bbits, rng, site);
  } while ((int64_t)x >= B);
  perm[i] = (int64_t)x;
}

extern "C" int dpc_random_perm(const uint64_t* rng, uint32_t site, int64_t B, int64_t* perm, uint64_t* keys_ws,
                               void* stream) {
  (void)keys_ws;                                            // (workspace of the former sort-based version)
  DPC_CHECK_ARG(rng && perm && B >= 0, "random_perm: bad argument");
  DPC_CHECK_ARG(B <= (1LL << 30), "random_perm: at most 2^30 elements");
  if (B == 0) return DPC_OK;
  int bits = 2;
  while ((1LL << bits) < B) ++bits;
  int abits = bits / 2, bbits = bits - abits;
  hipLaunchKernelGGL(perm_feistel_kernel, dim3((unsigned)dpc_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, rng, site,
                     B, abits, bbits, perm);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// encode (DataTransformer.transform)
// ---------------------------------------------------------------------------------------------
// Continuous column: tab = [mu(K) | sigma(K) | pc(K) | A(K)] (fp64), valid_idx = indices of the
// kept mixture components.  Writes value at column col_off and the one-hot at col_off+1.. .
#define DPC_MAX_COMP 16
__global__ __launch_bounds__(256) void encode_continuous_kernel(const double* __restrict__ x, int64_t xs, int64_t N,
                                                                const double* __restrict__ tab, int K,
                                                                const int32_t* __restrict__ valid_idx, int nv,
                                                                const double* __restrict__ u_inj, const uint64_t* rng,
                                                                uint32_t site, double* out64, float* out32,
                                                                int64_t ldo, int col_off, int32_t* comp_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* mu = tab;
  const double* sg = tab + K;
  const double* pc = tab + 2 * K;
  const double* A = tab + 3 * K;
  double xv = x[i * xs];
  double wlp[DPC_MAX_COMP];
  double mx = -INFINITY;
  for (int k = 0; k < K; ++k) {
    double y = (xv - mu[k]) * pc[k];
    wlp[k] = A[k] - 0.5 * (y * y);
    mx = fmax(mx, wlp[k]);
  }
  double se = 0.0;
  for (int k = 0; k < K; ++k) se += exp(wlp[k] - mx);
  double lse = mx + log(se);
  double pt[DPC_MAX_COMP];
  double ps = 0.0;
  for (int v = 0; v < nv; ++v) {
    pt[v] = exp(wlp[valid_idx[v]] - lse) + 1e-6;
    ps += pt[v];
  }
  double cum = 0.0, cdf[DPC_MAX_COMP];
  for (int v = 0; v < nv; ++v) {
    cum += pt[v] / ps;
    cdf[v] = cum;
  }
  double u = u_inj ? u_inj[i]
                   : dpc_u64_to_unit(dpc_draw_u32(rng, site, 2 * (uint64_t)i), dpc_draw_u32(rng, site, 2 * (uint64_t)i + 1));
  int comp = 0;
  for (int v = 0; v < nv; ++v) comp += (cdf[v] / cum <= u) ? 1 : 0;    // searchsorted(cdf/cdf[-1], u, 'right')
  if (comp > nv - 1) comp = nv - 1;
  int kc = valid_idx[comp];
  double val = (xv - mu[kc]) / (4.0 * sg[kc]);
  val = fmin(fmax(val, -0.99), 0.99);
  if (out64) {
    double* o = out64 + i * ldo + col_off;
    o[0] = val;
    for (int v = 0; v < nv; ++v) o[1 + v] = (v == comp) ? 1.0 : 0.0;
  }
  if (out32) {
    float* o = out32 + i * ldo + col_off;
    o[0] = (float)val;
    for (int v = 0; v < nv; ++v) o[1 + v] = (v == comp) ? 1.f : 0.f;
  }
  if (comp_out) comp_out[i] = comp;
}

extern "C" int dpc_encode_continuous(const double* x, int64_t x_stride, int64_t N, const double* tab, int K,
                                     const int32_t* valid_idx, int nv, const double* u_inj, const uint64_t* rng,
                                     uint32_t site, double* out64, float* out32, int64_t ldo, int col_off,
                                     int32_t* comp_out, void* stream) {
  DPC_CHECK_ARG(x && tab && valid_idx, "encode_continuous: null pointer");
  DPC_CHECK_ARG(K >= 1 && K <= DPC_MAX_COMP && nv >= 1 && nv <= K, "encode_continuous: 1 <= n_valid <= K <= 16");
  DPC_CHECK_ARG(u_inj || rng, "encode_continuous: no random source");
  DPC_CHECK_ARG(out64 || out32, "encode_continuous: no output");
  DPC_CHECK_ARG(ldo >= col_off + 1 + nv, "encode_continuous: output row too short");
  if (N <= 0) return N == 0 ? DPC_OK : DPC_EINVAL;
  hipLaunchKernelGGL(encode_continuous_kernel, dim3((unsigned)dpc_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     x_stride, N, tab, K, valid_idx, nv, u_inj, rng, site, out64, out32, ldo, col_off, comp_out);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Discrete column: codes[i] in [0, dim) or -1 (unseen -> all zeros, as OneHotEncoder does).
__global__ __launch_bounds__(256) void encode_onehot_kernel(const int32_t* __restrict__ codes, int64_t N, int dim,
                                                            double* out64, float* out32, int64_t ldo, int col_off) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * dim) return;
  int64_t i = idx / dim;
  int j = (int)(idx - i * dim);
  float v = (codes[i] == j) ? 1.f : 0.f;
  if (out64) out64[i * ldo + col_off + j] = v;
  if (out32) out32[i * ldo + col_off + j] = v;
}

extern "C" int dpc_encode_onehot(const int32_t* codes, int64_t N, int dim, double* out64, float* out32, int64_t ldo,
                                 int col_off, void* stream) {
  DPC_CHECK_ARG(codes && (out64 || out32), "encode_onehot: null pointer");
  DPC_CHECK_ARG(dim >= 1 && ldo >= col_off + dim, "encode_onehot: bad shape");
  if (N <= 0) return N == 0 ? DPC_OK : DPC_EINVAL;
  hipLaunchKernelGGL(encode_onehot_kernel, dim3((unsigned)dpc_cdiv(N * dim, 256)), dim3(256), 0, (hipStream_t)stream,
                     codes, N, dim, out64, out32, ldo, col_off);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// sample() path
// ---------------------------------------------------------------------------------------------
// sample_original_condvec (data_sampler.py:348-363): per sample a random table row and a random
// discrete column; the cond vector gets a single one at cond_st[col] + argmax(table[row, st:st+dim]),
// st = matrix_st[col].  The reference never fills its matrix offsets (they are all 0):
// pass matrix_st = NULL for that behaviour, the real span offsets for the intended one.
// rows_inj / cols_inj (int64, nullable) inject np.random.randint's draws.  One wave per sample.
__global__ __launch_bounds__(256) void condvec_original_kernel(const float* __restrict__ table, int64_t ld, int64_t N,
                                                               const int32_t* __restrict__ cond_st,
                                                               const int32_t* __restrict__ dims,
                                                               const int32_t* __restrict__ matrix_st, int nd, int C,
                                                               const int64_t* __restrict__ rows_inj,
                                                               const int64_t* __restrict__ cols_inj, const uint64_t* rng,
                                                               uint32_t site, int64_t B, float* __restrict__ cond,
                                                               int64_t ldc) {
  const int lane = threadIdx.x & 63;
  int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  int64_t row;
  int col;
  if (rows_inj) {
    row = rows_inj[b];
    col = (int)cols_inj[b];
  } else {
    double u = dpc_u64_to_unit(dpc_draw_u32(rng, site, 4 * (uint64_t)b), dpc_draw_u32(rng, site, 4 * (uint64_t)b + 1));
    row = min(N - 1, (int64_t)(u * (double)N));
    float v = dpc_u32_to_unit(dpc_draw_u32(rng, site, 4 * (uint64_t)b + 2));
    col = min(nd - 1, (int)(v * nd));
  }
  const int st = matrix_st ? matrix_st[col] : 0, dim = dims[col];
  // first maximum of table[row, st : st+dim] (numpy argmax), wave-parallel
  float best = -INFINITY;
  int arg = 0x7fffffff;
  for (int j = lane; j < dim; j += 64) {
    float x = table[row * ld + st + j];
    if (x > best) { best = x; arg = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64);
    int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  const int hot = cond_st[col] + arg;
  for (int c = lane; c < C; c += 64) cond[b * ldc + c] = (c == hot) ? 1.f : 0.f;
}

extern "C" int dpc_condvec_original(const float* table, int64_t ld, int64_t N, const int32_t* cond_st,
                                    const int32_t* dims, const int32_t* matrix_st, int nd, int C,
                                    const int64_t* rows_inj, const int64_t* cols_inj, const uint64_t* rng, uint32_t site,
                                    int64_t B, float* cond, int64_t ldc, void* stream) {
  DPC_CHECK_ARG(table && cond_st && dims && cond, "condvec_original: null pointer");
  DPC_CHECK_ARG(N > 0 && nd > 0 && C > 0 && ldc >= C && B >= 0, "condvec_original: bad shape");
  DPC_CHECK_ARG((rows_inj && cols_inj) || rng, "condvec_original: no random source");
  if (B == 0) return DPC_OK;
  hipLaunchKernelGGL(condvec_original_kernel, dim3((unsigned)dpc_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, table,
                     ld, N, cond_st, dims, matrix_st, nd, C, rows_inj, cols_inj, rng, site, B, cond, ldc);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// DataTransformer.inverse_transform (data_transformer.py:152-213) for generated rows: per output
// column j, desc[j] = (kind, data offset, width, parameter offset).  kind 0 (continuous): value =
// clip(data[off], -1, 1), component = valid_idx[argmax(data[off+1 : off+width])],
// out = value * 4 * std[component] + mean[component]  (fp64; params = [mean(Kc) | std(Kc) | valid_idx(nv)]
// at the parameter offset, Kc = desc width of the mixture = params_k).  kind 1 (discrete):
// out = argmax(data[off : off+width]) as a category index.  One thread per (row, column).
__global__ __launch_bounds__(256) void inverse_transform_kernel(const float* __restrict__ data, int64_t ld, int64_t n,
                                                                const int32_t* __restrict__ desc, int ncol,
                                                                const double* __restrict__ params, int Kc,
                                                                double* __restrict__ out, int64_t ldo) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * ncol) return;
  int64_t r = idx / ncol;
  int j = (int)(idx - r * ncol);
  const int kind = desc[4 * j], off = desc[4 * j + 1], width = desc[4 * j + 2], po = desc[4 * j + 3];
  const float* d = data + r * ld + off;
  if (kind == 0) {
    float v = fminf(fmaxf(d[0], -1.f), 1.f);
    int arg = 0;
    float best = d[1];
    for (int t = 1; t < width - 1; ++t) {
      float x = d[1 + t];
      if (x > best) { best = x; arg = t; }
    }
    const double* mean = params + po;
    const double* sd = params + po + Kc;
    int comp = (int)params[po + 2 * Kc + arg];
    // numpy: float32 value * 4 (float32), then * std + mean in float64 with two roundings: keep
    // hipcc from contracting them into one fma (HIP's __dmul_rn / __dadd_rn are plain operators)
    {
#pragma clang fp contract(off)
      double t = (double)(v * 4.f) * sd[comp];
      out[r * ldo + j] = t + mean[comp];
    }
  } else {
    int arg = 0;
    float best = d[0];
    for (int t = 1; t < width; ++t) {
      float x = d[t];
      if (x > best) { best = x; arg = t; }
    }
    out[r * ldo + j] = (double)arg;
  }
}

extern "C" int dpc_inverse_transform(const float* data, int64_t ld, int64_t n, const int32_t* desc, int ncol,
                                     const double* params, int Kc, double* out, int64_t ldo, void* stream) {
  DPC_CHECK_ARG(data && desc && out && ncol > 0 && n >= 0 && ldo >= ncol && Kc >= 1, "inverse_transform: bad argument");
  if (n == 0) return DPC_OK;
  hipLaunchKernelGGL(inverse_transform_kernel, dim3((unsigned)dpc_cdiv(n * ncol, 256)), dim3(256), 0, (hipStream_t)stream,
                     data, ld, n, desc, ncol, params, Kc, out, ldo);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
