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
// Up to 8 transposes in one launch: dst_k[c][r] = src_k[r][c], r < rows_k, c < cols_k (32 x 32 LDS
// tiles, both sides coalesced).  Keeps W[:, cond columns]^T of every generator layer current after
// an optimiser step, and writes the scattered cond-column gradients back into dW's layout.
// ---------------------------------------------------------------------------------------------
#define DPC_MAX_SLICES 8
struct TransArgs {
  const float* src[DPC_MAX_SLICES]; int64_t ld_src[DPC_MAX_SLICES];
  float* dst[DPC_MAX_SLICES]; int64_t ld_dst[DPC_MAX_SLICES];
  int rows[DPC_MAX_SLICES], cols[DPC_MAX_SLICES];
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
      if (c < cols && r < rows) a.dst[k][(int64_t)c * a.ld_dst[k] + r] = tile[tx][ty + 8 * i];
    }
    __syncthreads();
  }
}

// src / dst / ld_src / ld_dst / rows / cols: HOST arrays of nslices (<= 8) entries.
extern "C" int dpc_transpose_slices(int nslices, const float* const* src, const int64_t* ld_src, const int* rows,
                                    const int* cols, float* const* dst, const int64_t* ld_dst, void* stream) {
  DPC_CHECK_ARG(nslices >= 1 && nslices <= DPC_MAX_SLICES, "transpose_slices: 1..8 slices");
  DPC_CHECK_ARG(src && ld_src && rows && cols && dst && ld_dst, "transpose_slices: null pointer");
  TransArgs a;
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
// Counting sort of the (row, hot id) pairs of sel by id: order[p] = row, sid[p] = id for p < *n_items,
// ids ascending (rows of one id in arbitrary order).  One block; histogram, scan and cursors live in
// LDS (C <= 15360).  It lets dpc_hot_rows_sum add the rows of one id inside a block instead of with one
// atomic per element: the ids of a batch are Zipf-distributed, and thousands of same-address atomics
// serialise in L2 (30 us per layer at T1M against 5 us this way).
// ---------------------------------------------------------------------------------------------
#define DPC_SORT_MAX_C 15360          // (C + 1024) * 4 bytes of dynamic LDS <= 64 KB
__global__ __launch_bounds__(1024) void hot_sort_kernel(const int32_t* __restrict__ sel, int64_t B, int C,
                                                        int32_t* __restrict__ order, int32_t* __restrict__ sid,
                                                        int32_t* __restrict__ n_items) {
  extern __shared__ int32_t hist[];          // C bins + 1024 scan slots
  int32_t* scan = hist + C;
  const int t = threadIdx.x, nt = blockDim.x;
  for (int c = t; c < C; c += nt) hist[c] = 0;
  __syncthreads();
  for (int64_t p = t; p < 2 * B; p += nt) {
    const int id = sel[p];
    if (id >= 0 && id < C) atomicAdd(&hist[id], 1);
  }
  __syncthreads();
  // exclusive scan: each thread owns a contiguous run of bins
  const int per = (C + nt - 1) / nt;
  const int c0 = min(C, t * per), c1 = min(C, c0 + per);
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
  int run = scan[t] - sum;
  for (int c = c0; c < c1; ++c) {
    const int h = hist[c];
    hist[c] = run;
    run += h;
  }
  if (t == nt - 1) n_items[0] = scan[t];
  __syncthreads();
  for (int64_t p = t; p < 2 * B; p += nt) {
    const int id = sel[p];
    if (id >= 0 && id < C) {
      const int pos = atomicAdd(&hist[id], 1);
      order[pos] = (int32_t)(p >> 1);
      sid[pos] = id;
    }
  }
}

extern "C" int dpc_hot_sort(const int32_t* sel, int64_t B, int C, int32_t* order, int32_t* sid, int32_t* n_items,
                            void* stream) {
  DPC_CHECK_ARG(sel && order && sid && n_items && B > 0 && C > 0, "hot_sort: bad argument");
  DPC_CHECK_ARG(C <= DPC_SORT_MAX_C && 2 * B < (1ll << 31), "hot_sort: more than 15360 ids");
  hipLaunchKernelGGL(hot_sort_kernel, dim3(1), dim3(1024), (size_t)(C + 1024) * sizeof(int32_t), (hipStream_t)stream, sel, B,
                     C, order, sid, n_items);
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
                                                           int64_t ld_acc) {
  constexpr int CH = 64;
  __shared__ int32_t row_s[CH], id_s[CH];
  const int n = n_items[0];
  const int p0 = blockIdx.x * CH;
  if (p0 >= n) return;
  const int cnt = min(CH, n - p0);
  if (threadIdx.x < cnt) {
    row_s[threadIdx.x] = order[p0 + threadIdx.x];
    id_s[threadIdx.x] = sid[p0 + threadIdx.x];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < F; j += blockDim.x) {
    float sum = 0.f;
    int cur = id_s[0];
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {            // 8 independent row loads in flight
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = dy[(int64_t)row_s[i + q] * ld + j];
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
      sum += dy[(int64_t)row_s[i] * ld + j];
    }
    atomicAdd(acc + (int64_t)cur * ld_acc + j, sum);
  }
}

// max_items: the capacity of order / sid (2B); the grid is sized for it, blocks beyond *n_items exit.
extern "C" int dpc_hot_rows_sum(const float* dy, int64_t ld, int F, const int32_t* order, const int32_t* sid,
                                const int32_t* n_items, int64_t max_items, float* acc, int64_t ld_acc, void* stream) {
  DPC_CHECK_ARG(dy && order && sid && n_items && acc && F > 0 && ld >= F && ld_acc >= F && max_items > 0,
                "hot_rows_sum: bad argument");
  hipLaunchKernelGGL(hot_rows_sum_kernel, dim3((unsigned)dpc_cdiv(max_items, 64)), dim3(256), 0, (hipStream_t)stream, dy, ld,
                     F, order, sid, n_items, acc, ld_acc);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
