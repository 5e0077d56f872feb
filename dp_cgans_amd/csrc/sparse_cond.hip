// The conditional vector as what it is: one or two hot columns per row.
//
// sample_condvec_pair sets exactly two ones per row of cond (data_sampler.py:284-330),
// sample_original_condvec one (:348-363).  The generator reads cond as the trailing columns of every
// layer's input (cat([z, cond]) dp_cgan.py:606, cat([out, input]) :83), so for each Linear
//     x W^T = x_dense W_dense^T + sum over the hot columns c of W[:, c]
// and, in the backward pass, the cond columns of dW are sums of the rows of dY that have that column
// hot.  The kernels here keep those two pieces off the matrix cores: the forward share is a row gather
// fused into the GEMM epilogue (dpc_gemm_f32_gather), fed by the transposed column slices kept by
// dpc_transpose_slices; the backward share is dpc_hot_sort + dpc_hot_rows_sum.  All HBM/L2-bound index work.
#include "dpc_common.h"

// ---------------------------------------------------------------------------------------------
// sel[b] = column ids of the (up to two) nonzero entries of cond[b, :], ascending, -1 padded.
// One wave per row.  *bad (nullable) counts rows that are not one-/two-hot {0,1} vectors.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cond_to_sel_kernel(const float* __restrict__ cond, int64_t ld, int64_t B, int C,
                                                          int32_t* __restrict__ sel, int32_t* __restrict__ bad) {
  const int lane = threadIdx.x & 63;
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * 4) {
    int first = 0x7fffffff, second = 0x7fffffff, cnt = 0, odd = 0;
    for (int j = lane; j < C; j += 64) {
      const float v = cond[b * ld + j];
      if (v != 0.f) {
        ++cnt;
        if (v != 1.f) odd = 1;
        if (j < first) { second = first; first = j; }
        else if (j < second) second = j;
      }
    }
    // wave minimum, then the minimum of what is left
    int m0 = first;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m0 = min(m0, __shfl_xor(m0, o, 64));
    int rest = (first == m0) ? second : first;
    int m1 = rest;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m1 = min(m1, __shfl_xor(m1, o, 64));
    cnt = dpc_wave_sum(cnt);
    odd = dpc_wave_sum(odd);
    if (lane == 0) {
      sel[2 * b] = m0 == 0x7fffffff ? -1 : m0;
      sel[2 * b + 1] = m1 == 0x7fffffff ? -1 : m1;
      if (bad && (cnt > 2 || odd)) atomicAdd(bad, 1);
    }
  }
}

extern "C" int dpc_cond_to_sel(const float* cond, int64_t ld, int64_t B, int C, int32_t* sel, int32_t* bad, void* stream) {
  DPC_CHECK_ARG(cond && sel && B > 0 && C > 0 && ld >= C, "cond_to_sel: bad argument");
  int64_t blocks = dpc_cdiv(B, 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cond_to_sel_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, cond, ld, B, C, sel, bad);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Up to 16 transposes in one launch: dst_k[c][r] = src_k[r][c], r < rows_k, c < cols_k (32 x 32 LDS
// tiles, both sides coalesced).  Keeps W[:, cond columns]^T of every generator layer current after
// an optimiser step, and writes the scattered cond-column gradients back into dW's layout.
// ---------------------------------------------------------------------------------------------
#define DPC_MAX_SLICES 16
struct TransArgs {
  const float* src[DPC_MAX_SLICES]; int64_t ld_src[DPC_MAX_SLICES];
  float* dst[DPC_MAX_SLICES]; int64_t ld_dst[DPC_MAX_SLICES];
  int rows[DPC_MAX_SLICES], cols[DPC_MAX_SLICES];
  int accumulate;      // dst += src^T instead of dst = src^T
};

__global__ __launch_bounds__(256) void transpose_slices_kernel(TransArgs a) {
  __shared__ float tile[32][33];
  const int k = blockIdx.z;
  const int rows = a.rows[k], cols = a.cols[k];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
  const int tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
  for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      tile[ty + 8 * i][tx] = (r < rows && c < cols) ? a.src[k][(int64_t)r * a.ld_src[k] + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (c < cols && r < rows) {
        float* dp = a.dst[k] + (int64_t)c * a.ld_dst[k] + r;
        *dp = a.accumulate ? *dp + tile[tx][ty + 8 * i] : tile[tx][ty + 8 * i];
      }
    }
    __syncthreads();
  }
}

// src / dst / ld_src / ld_dst / rows / cols: HOST arrays of nslices (<= 8) entries.
extern "C" int dpc_transpose_slices(int nslices, const float* const* src, const int64_t* ld_src, const int* rows,
                                    const int* cols, float* const* dst, const int64_t* ld_dst, int accumulate,
                                    void* stream) {
  DPC_CHECK_ARG(nslices >= 1 && nslices <= DPC_MAX_SLICES, "transpose_slices: 1..16 slices");
  DPC_CHECK_ARG(src && ld_src && rows && cols && dst && ld_dst, "transpose_slices: null pointer");
  TransArgs a;
  a.accumulate = accumulate ? 1 : 0;
  int maxt = 1;
  for (int k = 0; k < nslices; ++k) {
    DPC_CHECK_ARG(src[k] && dst[k] && rows[k] > 0 && cols[k] > 0 && ld_src[k] >= cols[k] && ld_dst[k] >= rows[k],
                  "transpose_slices: bad slice");
    a.src[k] = src[k]; a.ld_src[k] = ld_src[k]; a.dst[k] = dst[k]; a.ld_dst[k] = ld_dst[k]; a.rows[k] = rows[k];
    a.cols[k] = cols[k];
    int t = (int)(dpc_cdiv(rows[k], 32) * dpc_cdiv(cols[k], 32));
    if (t > maxt) maxt = t;
  }
  if (maxt > 1024) maxt = 1024;
  hipLaunchKernelGGL(transpose_slices_kernel, dim3((unsigned)maxt, 1, (unsigned)nslices), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Counting sort of (row, hot id) pairs by id: order[p] = row, sid[p] = id for p < *n_items, ids ascending
// (rows of one id in arbitrary order).  It lets dpc_hot_rows_sum add the rows of one id inside a block
// instead of with one atomic per element: the ids of a batch are Zipf-distributed, and thousands of
// same-address atomics serialise in L2 (30 us per layer at T1M against 5 us this way).
// Pairs come from one or two id tables sel_a / sel_b int32 [B][2] (negative = no id).  With pac > 1 the B
// sub-rows are grouped pac to a row, as the discriminator's input is (dp_cgan.py:66-68): sub-row b of table
// t yields row t * (B / pac) + b / pac and id (b % pac) * C + sel[b][h] - the cond column inside the pac row.
// Multi-block without a grid-wide pass: block k owns the id range [lo, hi), reads ALL pairs (a few hundred
// KB, L2-resident), counts those below its range in registers (their total is where its output starts)
// and those inside in an LDS histogram, scans it and scatters its own pairs.  No global atomics, no
// workspace, any number of ids (the one-block version stopped at 15360).
// ---------------------------------------------------------------------------------------------
#define DPC_SORT_MAX_BINS 12288          // (bins + 1024) * 4 bytes of dynamic LDS <= 64 KB
struct SortArgs {
  const int32_t* sel[2]; int ntab; int64_t B; int pac; int C; int bins;
  int32_t* order; int32_t* sid; int32_t* n_items;
  float inv_pac;
};

// (row, id) of pair p, p < ntab * 2B < 2^31: 32-bit arithmetic only (a 64-bit division per pair made the
// first version of this kernel slower than the one-block sort it replaced)
__device__ __forceinline__ bool sort_item(const SortArgs& a, uint32_t p, int raw, int& id, int& row) {
  if (raw < 0 || raw >= a.C) return false;
  const uint32_t per = 2u * (uint32_t)a.B;
  const uint32_t tab = p >= per ? 1u : 0u;
  const uint32_t b = (p - tab * per) >> 1;
  uint32_t grp = b;
  if (a.pac > 1) {
    grp = (uint32_t)((float)b * a.inv_pac);          // b / pac, corrected (b < 2^24: exact float)
    if ((grp + 1) * (uint32_t)a.pac <= b) ++grp;
    if (grp * (uint32_t)a.pac > b) --grp;
  }
  id = (int)(b - grp * (uint32_t)a.pac) * a.C + raw;
  row = (int)(tab * (uint32_t)(a.B / a.pac) + grp);
  return true;
}

// IPT pairs per thread stay in registers between the counting and the scatter pass, loaded with
// independent loads up front (a load -> LDS-atomic loop is a chain of L2 round trips: 16 iterations took
// 25 us); IPT == 0: any number of pairs, read twice.
template <int IPT>
__global__ __launch_bounds__(1024) void hot_sort_kernel(SortArgs a) {
  extern __shared__ int32_t hist[];          // bins + 1024 scan slots
  __shared__ int32_t red[16];
  int32_t* scan = hist + a.bins;
  const int t = threadIdx.x, nt = blockDim.x;
  const int lo = blockIdx.x * a.bins, hi = min(lo + a.bins, a.pac * a.C);
  const int nb = hi - lo;
  const uint32_t nitem = (uint32_t)a.ntab * 2u * (uint32_t)a.B, per = 2u * (uint32_t)a.B;
  int raw[IPT > 0 ? IPT : 1];
  if (IPT > 0) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const uint32_t p = (uint32_t)t + (uint32_t)k * 1024u;
      raw[k] = p < nitem ? (p >= per ? a.sel[1][p - per] : a.sel[0][p]) : -1;
    }
  }
  for (int c = t; c < nb; c += nt) hist[c] = 0;
  __syncthreads();
  int below = 0;
  if (IPT > 0) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      int id, row;
      if (!sort_item(a, (uint32_t)t + (uint32_t)k * 1024u, raw[k], id, row)) continue;
      if (id < lo) ++below;
      else if (id < hi) atomicAdd(&hist[id - lo], 1);
    }
  } else {
    for (uint32_t p = t; p < nitem; p += nt) {
      int id, row;
      if (!sort_item(a, p, p >= per ? a.sel[1][p - per] : a.sel[0][p], id, row)) continue;
      if (id < lo) ++below;
      else if (id < hi) atomicAdd(&hist[id - lo], 1);
    }
  }
  // pairs below this block's range: where its output starts
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o, 64);
  if ((t & 63) == 0) red[t >> 6] = below;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (nt >> 6); ++w) base += red[w];
  // exclusive scan of the histogram: each thread owns a contiguous run of bins
  const int perb = (nb + nt - 1) / nt;
  const int c0 = min(nb, t * perb), c1 = min(nb, c0 + perb);
  int sum = 0;
  for (int c = c0; c < c1; ++c) sum += hist[c];
  scan[t] = sum;
  __syncthreads();
  for (int o = 1; o < nt; o <<= 1) {
    int v = t >= o ? scan[t - o] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  int run = base + scan[t] - sum;
  for (int c = c0; c < c1; ++c) {
    const int h = hist[c];
    hist[c] = run;
    run += h;
  }
  if (t == nt - 1 && blockIdx.x == gridDim.x - 1) a.n_items[0] = base + scan[t];
  __syncthreads();
  if (IPT > 0) {
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      int id, row;
      if (!sort_item(a, (uint32_t)t + (uint32_t)k * 1024u, raw[k], id, row) || id < lo || id >= hi) continue;
      const int pos = atomicAdd(&hist[id - lo], 1);
      a.order[pos] = row;
      a.sid[pos] = id;
    }
  } else {
    for (uint32_t p = t; p < nitem; p += nt) {
      int id, row;
      if (!sort_item(a, p, p >= per ? a.sel[1][p - per] : a.sel[0][p], id, row) || id < lo || id >= hi) continue;
      const int pos = atomicAdd(&hist[id - lo], 1);
      a.order[pos] = row;
      a.sid[pos] = id;
    }
  }
}

extern "C" int dpc_hot_sort(const int32_t* sel_a, const int32_t* sel_b, int64_t B, int pac, int C, int32_t* order,
                            int32_t* sid, int32_t* n_items, void* stream) {
  DPC_CHECK_ARG(sel_a && order && sid && n_items && B > 0 && C > 0 && pac >= 1, "hot_sort: bad argument");
  DPC_CHECK_ARG(B % pac == 0, "hot_sort: B must be a multiple of pac");
  DPC_CHECK_ARG(4 * B < (1ll << 31) && (int64_t)pac * C < (1ll << 30), "hot_sort: too many pairs or ids");
  const int64_t ids = (int64_t)pac * C;
  int64_t nblk = dpc_cdiv(ids, 64);                           // ~64 ids per block until 128 blocks are busy
  if (nblk > 128) nblk = 128;
  if (nblk < dpc_cdiv(ids, DPC_SORT_MAX_BINS)) nblk = dpc_cdiv(ids, DPC_SORT_MAX_BINS);
  SortArgs a;
  a.sel[0] = sel_a; a.sel[1] = sel_b; a.ntab = sel_b ? 2 : 1; a.B = B; a.pac = pac; a.C = C;
  a.bins = (int)dpc_cdiv(ids, nblk);
  nblk = dpc_cdiv(ids, a.bins);
  a.order = order; a.sid = sid; a.n_items = n_items; a.inv_pac = 1.0f / (float)pac;
  DPC_CHECK_ARG(B < (1 << 24), "hot_sort: at most 2^24 sub-rows");
  const int64_t nitem = (int64_t)a.ntab * 2 * B;
  const size_t lds = (size_t)(a.bins + 1024) * sizeof(int32_t);
  hipStream_t st = (hipStream_t)stream;
  if (nitem <= 1024 * 8) hipLaunchKernelGGL(hot_sort_kernel<8>, dim3((unsigned)nblk), dim3(1024), lds, st, a);
  else if (nitem <= 1024 * 32) hipLaunchKernelGGL(hot_sort_kernel<32>, dim3((unsigned)nblk), dim3(1024), lds, st, a);
  else hipLaunchKernelGGL(hot_sort_kernel<0>, dim3((unsigned)nblk), dim3(1024), lds, st, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// acc[id][j] += sum of dy[row][j] over the sorted (row, id) pairs: the cond columns of a weight
// gradient dW = dY^T X, transposed ([C, F]).  A block takes 64 consecutive pairs and one thread a
// column: it adds rows in registers while the id stays the same and issues one fp32 atomic per id
// change, ~(2B/64 + C) * F atomics in all.  acc must be zeroed (or hold the running sum) by the
// caller.  The summation order varies from run to run (pair order inside an id, atomics between
// blocks): callers that need reproducible steps take the dense product instead.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hot_rows_sum_kernel(const float* __restrict__ dy, int64_t ld, int F,
                                                           const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ sid,
                                                           const int32_t* __restrict__ n_items, float* __restrict__ acc,
                                                           int64_t ld_acc, const float* __restrict__ rowscale) {
  constexpr int CH = 64;
  __shared__ int32_t row_s[CH], id_s[CH];
  __shared__ float sc_s[CH];
  const int n = n_items[0];
  const int p0 = blockIdx.x * CH;
  if (p0 >= n) return;
  const int cnt = min(CH, n - p0);
  if (threadIdx.x < cnt) {
    row_s[threadIdx.x] = order[p0 + threadIdx.x];
    id_s[threadIdx.x] = sid[p0 + threadIdx.x];
    sc_s[threadIdx.x] = rowscale ? rowscale[row_s[threadIdx.x]] : 1.f;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < F; j += blockDim.x) {
    float sum = 0.f;
    int cur = id_s[0];
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {            // 8 independent row loads in flight
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = dy[(int64_t)row_s[i + q] * ld + j] * sc_s[i + q];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int id = id_s[i + q];
        if (id != cur) {
          atomicAdd(acc + (int64_t)cur * ld_acc + j, sum);
          sum = 0.f;
          cur = id;
        }
        sum += v[q];
      }
    }
    for (; i < cnt; ++i) {
      const int id = id_s[i];
      if (id != cur) {
        atomicAdd(acc + (int64_t)cur * ld_acc + j, sum);
        sum = 0.f;
        cur = id;
      }
      sum += dy[(int64_t)row_s[i] * ld + j] * sc_s[i];
    }
    atomicAdd(acc + (int64_t)cur * ld_acc + j, sum);
  }
}

// max_items: the capacity of order / sid (2B); the grid is sized for it, blocks beyond *n_items exit.
extern "C" int dpc_hot_rows_sum(const float* dy, int64_t ld, int F, const int32_t* order, const int32_t* sid,
                                const int32_t* n_items, int64_t max_items, float* acc, int64_t ld_acc,
                                const float* rowscale, void* stream) {
  DPC_CHECK_ARG(dy && order && sid && n_items && acc && F > 0 && ld >= F && ld_acc >= F && max_items > 0,
                "hot_rows_sum: bad argument");
  hipLaunchKernelGGL(hot_rows_sum_kernel, dim3((unsigned)dpc_cdiv(max_items, 64)), dim3(256), 0, (hipStream_t)stream, dy, ld,
                     F, order, sid, n_items, acc, ld_acc, rowscale);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
