// Non-GEMM pieces of the generator / discriminator / gradient-penalty passes.
// Reference: Residual (Linear+BatchNorm1d+ReLU) dp_cgan.py:71-83, Discriminator :31-68,
// calc_gradient_penalty :46-64 (analytic form).  All HBM-bound elementwise / column or
// row reductions; reductions accumulate in fp64 and are bit-reproducible (fixed partial layout,
// fixed summation order, no atomics).
#include "dpc_common.h"
#include "../../include/dpcgan_hip.h"
#include <mutex>

// ---------------------------------------------------------------------------------------------
// Column reductions over a [rows, F] matrix: grid (ceil(F/64), R), block 256 = 64 columns x 4 row
// lanes.  part layout: [R][NOUT][F] doubles.
//   MODE 0: sum x, sum x^2                       (BatchNorm forward statistics)
//   MODE 1: sum w[r] x                           (bias gradients, weighted row sums)
//   MODE 2: sum dyh, sum dyh * xhat              (BatchNorm backward; dyh = dy * (y > 0))
// ---------------------------------------------------------------------------------------------
struct ColArgs {
  const float* x; int64_t ldx; int64_t rows; int F;
  const float* w;                                   // MODE 1 row weights (nullable -> 1)
  const float* y; int64_t ldy; const float* z; int64_t ldz; const float* mean; const float* invstd;  // MODE 2
  double* part; int R;
};

template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(ColArgs a) {
  constexpr int NOUT = (MODE == 1) ? 1 : 2;
  __shared__ double red[NOUT][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + c;
  const int64_t per = (a.rows + a.R - 1) / a.R;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(a.rows, r0 + per);
  double s0 = 0.0, s1 = 0.0;
  if (f < a.F) {
    float mu = 0.f, is = 0.f;
    if (MODE == 2) { mu = a.mean[f]; is = a.invstd[f]; }
    // unrolled so that 8 independent row loads are in flight per thread (the sums are cheap; what
    // bounds this kernel is load latency at 1-2 blocks per CU)
#pragma unroll 8
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      float v = a.x[r * a.ldx + f];
      if (MODE == 0) {
        s0 += (double)v;
        s1 += (double)v * (double)v;
      } else if (MODE == 1) {
        s0 += (double)(a.w ? a.w[r] * v : v);
      } else {
        float dyh = a.y[r * a.ldy + f] > 0.f ? v : 0.f;
        float xh = (a.z[r * a.ldz + f] - mu) * is;
        s0 += (double)dyh;
        s1 += (double)dyh * (double)xh;
      }
    }
  }
  red[0][rl][c] = s0;
  if (NOUT == 2) red[1][rl][c] = s1;
  __syncthreads();
  if (rl == 0 && f < a.F) {
    double t0 = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    a.part[((int64_t)blockIdx.y * NOUT + 0) * a.F + f] = t0;
    if (NOUT == 2) {
      double t1 = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
      a.part[((int64_t)blockIdx.y * NOUT + 1) * a.F + f] = t1;
    }
  }
}

static int pick_R(int64_t rows, int F, int max_R) {
  // enough row chunks to spread the reduction over the chip, at least 256 rows per chunk; the
  // finish kernel walks the chunks serially, so keep them few
  int64_t colblocks = dpc_cdiv(F, 64);
  int64_t R = dpc_cdiv(512, colblocks);
  int64_t cap = dpc_cdiv(rows, 64);
  if (R > cap) R = cap;
  if (R > max_R) R = max_R;
  if (R < 1) R = 1;
  return (int)R;
}

// part must hold R * nout * F doubles; returns R through *R_out (R <= max_R).
extern "C" int dpc_colreduce_partial(int mode, const float* x, int64_t ldx, int64_t rows, int F, const float* w,
                                     const float* y, int64_t ldy, const float* z, int64_t ldz, const float* mean,
                                     const float* invstd, double* part, int max_R, int* R_out, void* stream) {
  DPC_CHECK_ARG(x && part && R_out, "colreduce_partial: null pointer");
  DPC_CHECK_ARG(mode >= 0 && mode <= 2, "colreduce_partial: unknown mode");
  DPC_CHECK_ARG(rows > 0 && F > 0 && ldx >= F && max_R >= 1, "colreduce_partial: bad shape");
  DPC_CHECK_ARG(mode != 2 || (y && z && mean && invstd), "colreduce_partial: BatchNorm backward needs y, z, mean, invstd");
  ColArgs a;
  a.x = x; a.ldx = ldx; a.rows = rows; a.F = F; a.w = w; a.y = y; a.ldy = ldy; a.z = z; a.ldz = ldz; a.mean = mean;
  a.invstd = invstd; a.part = part; a.R = pick_R(rows, F, max_R);
  *R_out = a.R;
  dim3 grid((unsigned)dpc_cdiv(F, 64), a.R), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(colreduce_kernel<0>, grid, block, 0, st, a);
  else if (mode == 1) hipLaunchKernelGGL(colreduce_kernel<1>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(colreduce_kernel<2>, grid, block, 0, st, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Several weighted column sums in two launches (partials for all, finish for all): out_p[f] (+)= sum_r
// w_p[r] * x_p[r, f].  The bias gradients and the head-layer weight gradient of a network's backward pass are
// such sums over different matrices; one launch pair for all of them instead of one per sum (a kernel node
// costs ~5 us whatever it does).  Same kernels' arithmetic and order as dpc_colreduce_partial(mode 1) +
// dpc_colreduce_finish(kind 1) per problem: bit-reproducible.
// ---------------------------------------------------------------------------------------------
#define DPC_MAX_COLSUMS 8
struct MultiColArgs {
  const float* x[DPC_MAX_COLSUMS]; int64_t ldx[DPC_MAX_COLSUMS]; int64_t rows[DPC_MAX_COLSUMS]; int F[DPC_MAX_COLSUMS];
  const float* w[DPC_MAX_COLSUMS]; float* out[DPC_MAX_COLSUMS]; int R[DPC_MAX_COLSUMS];
  int blk0[DPC_MAX_COLSUMS + 1];        // first block of problem p in the partial launch (strips * R blocks each)
  int fin0[DPC_MAX_COLSUMS + 1];        // ... in the finish launch (strips blocks each)
  int64_t part0[DPC_MAX_COLSUMS];       // offset of problem p's partials
  double* part; int n; int accumulate;
  int adv_on; dpc_step_advance adv;     // end-of-step bookkeeping riding in this launch (dpc_colsum_multi_adv)
};

__global__ __launch_bounds__(256) void colsum_multi_partial_kernel(MultiColArgs a) {
  __shared__ double red[4][64];
  if (a.adv_on && blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.adv.rng_a) a.adv.rng_a[1] += 1;
    if (a.adv.rng_b) a.adv.rng_b[1] += 1;
    if (a.adv.t_a) a.adv.t_a[0] += a.adv.add_a;
    if (a.adv.t_b) a.adv.t_b[0] += a.adv.add_b;
  }
  int p = 0;
  while (p + 1 < a.n && (int)blockIdx.x >= a.blk0[p + 1]) ++p;
  const int local = blockIdx.x - a.blk0[p];
  const int strips = (a.F[p] + 63) / 64;
  const int strip = local % strips, chunk = local / strips;
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = strip * 64 + c;
  const int64_t per = (a.rows[p] + a.R[p] - 1) / a.R[p];
  const int64_t r0 = (int64_t)chunk * per, r1 = min(a.rows[p], r0 + per);
  const float* __restrict__ x = a.x[p];
  const float* __restrict__ w = a.w[p];
  const int64_t ld = a.ldx[p];
  double s0 = 0.0;
  if (f < a.F[p]) {
#pragma unroll 8
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float v = x[r * ld + f];
      s0 += (double)(w ? w[r] * v : v);
    }
  }
  red[rl][c] = s0;
  __syncthreads();
  if (rl == 0 && f < a.F[p]) {
    const double t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    if (a.accumulate == 2) atomicAdd(a.out[p] + f, (float)t);       // no finish launch: the chunks meet in out
    else a.part[a.part0[p] + (int64_t)chunk * a.F[p] + f] = t;
  }
}

__global__ __launch_bounds__(256) void colsum_multi_finish_kernel(MultiColArgs a) {
  __shared__ double red[4][64];
  int p = 0;
  while (p + 1 < a.n && (int)blockIdx.x >= a.fin0[p + 1]) ++p;
  const int strip = blockIdx.x - a.fin0[p];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = strip * 64 + c;
  const int F = a.F[p], R = a.R[p];
  const double* __restrict__ part = a.part + a.part0[p];
  double s0 = 0.0;
  if (f < F) {
#pragma unroll 8
    for (int r = rl; r < R; r += 4) s0 += part[(int64_t)r * F + f];
  }
  red[rl][c] = s0;
  __syncthreads();
  if (rl != 0 || f >= F) return;
  s0 = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  a.out[p][f] = a.accumulate ? a.out[p][f] + (float)s0 : (float)s0;
}

// x / ldx / rows / F / w / out: HOST arrays of n (<= 8) entries (w[p] nullable = all ones); part: fp64 workspace
// of part_cap elements (sum over p of R_p * F_p, R_p <= 64, is enough: 64 * sum F always is).
extern "C" int dpc_colsum_multi(int n, const float* const* x, const int64_t* ldx, const int64_t* rows, const int* F,
                                const float* const* w, float* const* out, int accumulate, double* part,
                                int64_t part_cap, void* stream) {
  return dpc_colsum_multi_adv(n, x, ldx, rows, F, w, out, accumulate, part, part_cap, nullptr, stream);
}

extern "C" int dpc_colsum_multi_adv(int n, const float* const* x, const int64_t* ldx, const int64_t* rows, const int* F,
                                    const float* const* w, float* const* out, int accumulate, double* part,
                                    int64_t part_cap, const dpc_step_advance* adv, void* stream) {
  DPC_CHECK_ARG(n >= 1 && n <= DPC_MAX_COLSUMS, "colsum_multi: 1..8 sums");
  DPC_CHECK_ARG(x && ldx && rows && F && w && out && (part || accumulate == 2), "colsum_multi: null pointer");
  DPC_CHECK_ARG(accumulate >= 0 && accumulate <= 2, "colsum_multi: accumulate must be 0, 1 or 2");
  MultiColArgs a;
  a.n = n; a.part = part; a.accumulate = accumulate;
  a.adv_on = adv != nullptr;
  a.adv = adv ? *adv : dpc_step_advance{};
  int blk = 0, fin = 0;
  int64_t off = 0;
  for (int p = 0; p < n; ++p) {
    DPC_CHECK_ARG(x[p] && out[p] && rows[p] > 0 && F[p] > 0 && ldx[p] >= F[p], "colsum_multi: bad problem");
    a.x[p] = x[p]; a.ldx[p] = ldx[p]; a.rows[p] = rows[p]; a.F[p] = F[p]; a.w[p] = w[p]; a.out[p] = out[p];
    a.R[p] = pick_R(rows[p], F[p], 64);
    const int strips = (int)dpc_cdiv(F[p], 64);
    a.blk0[p] = blk; a.fin0[p] = fin; a.part0[p] = off;
    blk += strips * a.R[p];
    fin += strips;
    off += (int64_t)a.R[p] * F[p];
  }
  a.blk0[n] = blk; a.fin0[n] = fin;
  DPC_CHECK_ARG(accumulate == 2 || off <= part_cap, "colsum_multi: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_multi_partial_kernel, dim3((unsigned)blk), dim3(256), 0, st, a);
  if (accumulate != 2) hipLaunchKernelGGL(colsum_multi_finish_kernel, dim3((unsigned)fin), dim3(256), 0, st, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// BatchNorm1d statistics from the column sums, and the running-statistics update
// running = (1 - momentum) * running + momentum * new.  One function with FMA contraction off: both
// kernels that apply it (colreduce_finish_kernel and the fused BatchNorm forward) must agree to the bit.
__device__ __forceinline__ void bn_stats(double s0, double s1, double count, float eps, float& mean, float& invstd,
                                         float* run_mean, float* run_var, int f, float momentum) {
#pragma clang fp contract(off)
  const double meand = s0 / count;
  double var = s1 / count - meand * meand;
  if (var < 0.0) var = 0.0;
  mean = (float)meand;
  invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[f] = (1.f - momentum) * run_mean[f] + momentum * mean;
    run_var[f] = (1.f - momentum) * run_var[f] + momentum * (float)unbiased;
  }
}

// Finish a column reduction.  kind 0: BatchNorm statistics -> mean, invstd (+ running stats);
// kind 1: plain sums -> out0 (float), accumulate optional; kind 2: two sums -> out0, out1 (dbeta, dgamma).
__global__ __launch_bounds__(256) void colreduce_finish_kernel(const double* __restrict__ part, int R, int nout, int F,
                                                               int kind, double count, float eps, float momentum,
                                                               float* out0, float* out1, float* run_mean,
                                                               float* run_var, int accumulate) {
  // block = 64 columns x 4 row lanes: lane rl adds chunks rl, rl+4, ... (all loads of a lane independent,
  // 8 in flight), then the four lane sums are added in a fixed order - the walk over up to 64 chunks by
  // one thread per column was a chain of dependent L2 round trips (8 us per launch at T1M)
  __shared__ double red[2][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + c;
  double s0 = 0.0, s1 = 0.0;
  if (f < F) {
    if (nout == 2) {
#pragma unroll 4
      for (int r = rl; r < R; r += 4) {
        s0 += part[((int64_t)r * 2 + 0) * F + f];
        s1 += part[((int64_t)r * 2 + 1) * F + f];
      }
    } else {
#pragma unroll 8
      for (int r = rl; r < R; r += 4) s0 += part[(int64_t)r * F + f];
    }
  }
  red[0][rl][c] = s0;
  red[1][rl][c] = s1;
  __syncthreads();
  if (rl != 0 || f >= F) return;
  s0 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  s1 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  if (kind == 0) {
    float mean, invstd;
    bn_stats(s0, s1, count, eps, mean, invstd, run_mean, run_var, f, momentum);
    out0[f] = mean;
    out1[f] = invstd;
  } else {
    out0[f] = accumulate ? out0[f] + (float)s0 : (float)s0;
    if (nout == 2) out1[f] = accumulate ? out1[f] + (float)s1 : (float)s1;
  }
}

extern "C" int dpc_colreduce_finish(const double* part, int R, int nout, int F, int kind, double count, float eps,
                                    float momentum, float* out0, float* out1, float* run_mean, float* run_var,
                                    int accumulate, void* stream) {
  DPC_CHECK_ARG(part && out0, "colreduce_finish: null pointer");
  DPC_CHECK_ARG(R >= 1 && (nout == 1 || nout == 2) && F > 0 && kind >= 0 && kind <= 2, "colreduce_finish: bad argument");
  DPC_CHECK_ARG(nout == 1 || out1, "colreduce_finish: second output missing");
  DPC_CHECK_ARG(kind != 0 || (nout == 2 && count > 0.0), "colreduce_finish: statistics need two sums and a count");
  hipLaunchKernelGGL(colreduce_finish_kernel, dim3((unsigned)dpc_cdiv(F, 64)), dim3(256), 0, (hipStream_t)stream, part,
                     R, nout, F, kind, count, eps, momentum, out0, out1, run_mean, run_var, accumulate);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Fold the R row-chunk partials of a column reduction into ONE fp64 vector per set: out[set][j] = sum_r
// part[set][r][j], j < width (= nout * F), in the finish kernels' order (lane rl adds chunks rl, rl+4, ..; then
// (s0 + s1) + (s2 + s3)) - a consumer given the folded vector with R = 1 gets bit-identical sums.  This is what a
// data-parallel step all-reduces: width doubles per BatchNorm layer instead of R x width.  In place is fine
// (out == part: every thread reads its column of all chunks before it writes).
__global__ __launch_bounds__(256) void colreduce_fold_kernel(const double* part, int R, int width, int64_t set_stride,
                                                             double* out, int64_t out_stride) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const double* p = part + (int64_t)blockIdx.y * set_stride;
  double s = 0.0;
  if (j < width) {
#pragma unroll 8
    for (int r = rl; r < R; r += 4) s += p[(int64_t)r * width + j];
  }
  red[rl][c] = s;
  __syncthreads();
  if (rl != 0 || j >= width) return;
  out[(int64_t)blockIdx.y * out_stride + j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

extern "C" int dpc_colreduce_fold(const double* part, int R, int width, int nsets, int64_t set_stride, double* out,
                                  int64_t out_stride, void* stream) {
  DPC_CHECK_ARG(part && out, "colreduce_fold: null pointer");
  DPC_CHECK_ARG(R >= 1 && width > 0 && nsets >= 1 && nsets <= 65535, "colreduce_fold: bad argument");
  DPC_CHECK_ARG(nsets == 1 || (set_stride >= (int64_t)R * width && out_stride >= width), "colreduce_fold: sets overlap");
  hipLaunchKernelGGL(colreduce_fold_kernel, dim3((unsigned)dpc_cdiv(width, 64), (unsigned)nsets), dim3(256), 0,
                     (hipStream_t)stream, part, R, width, set_stride, out, out_stride);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// dpc_colsum_noalloc: the weighted column sums of dpc_colreduce_partial(mode 1) + dpc_colreduce_finish
// (kind 1) in one call whose row-chunk partials live in a file-scope device array instead of a
// caller-provided workspace - for callers that cannot allocate (code under graph capture, serving
// loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
// `scratch` is a HOST-SIDE STAND-IN in host code, not a device address: each GPU holds its own copy at
// its own address, looked up once per device with hipGetSymbolAddress and kept in g_scratch_ptr;
// kernels, memsets and copies only ever see that pointer.
// ---------------------------------------------------------------------------------------------
#define DPC_NOALLOC_MAX_F 8192
#define DPC_NOALLOC_MAX_R 64
#define DPC_NOALLOC_MAX_DEV 64
static __device__ __align__(256) double scratch[(size_t)DPC_NOALLOC_MAX_R * DPC_NOALLOC_MAX_F];     // 4 MiB per GPU
static double* g_scratch_ptr[DPC_NOALLOC_MAX_DEV];
static std::mutex g_scratch_mu;

static int noalloc_scratch(double** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { dpc_set_error(hipGetErrorString(e)); return DPC_EHIP; }
  DPC_CHECK_ARG(dev >= 0 && dev < DPC_NOALLOC_MAX_DEV, "colsum_noalloc: device index beyond 63");
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));      // this device's copy of the array
    if (e != hipSuccess || !p) { dpc_set_error(e != hipSuccess ? hipGetErrorString(e) : "hipGetSymbolAddress returned null"); return DPC_EHIP; }
    g_scratch_ptr[dev] = static_cast<double*>(p);
  }
  *out = g_scratch_ptr[dev];
  return DPC_OK;
}

// out[f] (+)= sum_r w[r] * x[r, f] (w nullable = 1): the same two kernels, the same row-chunk count R
// and the same summation order as dpc_colreduce_partial(mode 1, max_R 64) followed by
// dpc_colreduce_finish(kind 1) - bit-identical results.  Allocates nothing; F <= 8192.  Every call on
// a GPU uses that GPU's one scratch array (all R x F partials it reads are rewritten first, nothing
// needs clearing): calls that overlap on one GPU - two streams, two threads - race.
// The FIRST call on a device looks the array's address up (hipGetSymbolAddress, under a mutex): make it outside any
// stream capture - a warm-up call at set-up - and every later call on that device is launches only.
extern "C" int dpc_colsum_noalloc(const float* x, int64_t ldx, int64_t rows, int F, const float* w, float* out,
                                  int accumulate, void* stream) {
  DPC_CHECK_ARG(x && out, "colsum_noalloc: null pointer");
  DPC_CHECK_ARG(rows > 0 && F > 0 && ldx >= F, "colsum_noalloc: bad shape");
  DPC_CHECK_ARG(F <= DPC_NOALLOC_MAX_F, "colsum_noalloc: more than 8192 columns (the static scratch holds 64 x 8192 partial sums)");
  // The scratch array used is the CURRENT device's copy, and the kernels run on `stream`: all of x, out (and w) must
  // live on the current device, or the launch would hand one GPU's scratch address to another GPU's kernel.
  {
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess) { dpc_set_error(hipGetErrorString(e)); return DPC_EHIP; }
    const void* ptrs[3] = {x, out, w};
    for (int i = 0; i < 3; ++i) {
      if (!ptrs[i]) continue;
      hipPointerAttribute_t at;
      e = hipPointerGetAttributes(&at, ptrs[i]);
      if (e != hipSuccess) { (void)hipGetLastError(); dpc_set_error("colsum_noalloc: x, out and w must be device pointers"); return DPC_EINVAL; }
      DPC_CHECK_ARG(at.type == hipMemoryTypeDevice && at.device == cur,
                    "colsum_noalloc: x, out and w must live on the current device (its scratch array is the one used)");
    }
  }
  double* part = nullptr;
  int rc = noalloc_scratch(&part);
  if (rc != DPC_OK) return rc;
  ColArgs a;
  a.x = x; a.ldx = ldx; a.rows = rows; a.F = F; a.w = w; a.y = nullptr; a.ldy = 0; a.z = nullptr; a.ldz = 0; a.mean = nullptr;
  a.invstd = nullptr; a.part = part; a.R = pick_R(rows, F, DPC_NOALLOC_MAX_R);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3((unsigned)dpc_cdiv(F, 64), a.R), dim3(256), 0, st, a);
  hipLaunchKernelGGL(colreduce_finish_kernel, dim3((unsigned)dpc_cdiv(F, 64)), dim3(256), 0, st, part, a.R, 1, F, 1, 1.0,
                     0.f, 0.f, out, (float*)nullptr, (float*)nullptr, (float*)nullptr, accumulate);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// y = relu(gamma * (z - mean) * invstd + beta)
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ z, int64_t ldz, float* __restrict__ y,
                                                          int64_t ldy, int64_t B, int F, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta) {
  int64_t total = B * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / F;
    int f = (int)(i - r * F);
    float v = (z[r * ldz + f] - mean[f]) * invstd[f] * gamma[f] + beta[f];
    y[r * ldy + f] = fmaxf(v, 0.f);
  }
}

extern "C" int dpc_bn_relu_fwd(const float* z, int64_t ldz, float* y, int64_t ldy, int64_t B, int F, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, void* stream) {
  DPC_CHECK_ARG(z && y && mean && invstd && gamma && beta, "bn_relu_fwd: null pointer");
  DPC_CHECK_ARG(B > 0 && F > 0 && ldz >= F && ldy >= F, "bn_relu_fwd: bad shape");
  hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(dpc_grid(B * F, 256)), dim3(256), 0, (hipStream_t)stream, z, ldz, y, ldy,
                     B, F, mean, invstd, gamma, beta);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// dz = gamma * invstd * (dyh - dbeta/count - xhat * dgamma/count),  dyh = dy * (y > 0)
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                          const float* __restrict__ y, int64_t ldy,
                                                          const float* __restrict__ z, int64_t ldz, int64_t B, int F,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ dbeta,
                                                          const float* __restrict__ dgamma, float inv_count,
                                                          float* __restrict__ dz, int64_t lddz) {
  int64_t total = B * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / F;
    int f = (int)(i - r * F);
    float dyh = y[r * ldy + f] > 0.f ? dy[r * lddy + f] : 0.f;
    float xh = (z[r * ldz + f] - mean[f]) * invstd[f];
    dz[r * lddz + f] = gamma[f] * invstd[f] * (dyh - dbeta[f] * inv_count - xh * dgamma[f] * inv_count);
  }
}

extern "C" int dpc_bn_relu_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* z, int64_t ldz,
                               int64_t B, int F, const float* mean, const float* invstd, const float* gamma,
                               const float* dbeta, const float* dgamma, double count, float* dz, int64_t lddz,
                               void* stream) {
  DPC_CHECK_ARG(dy && y && z && mean && invstd && gamma && dbeta && dgamma && dz, "bn_relu_bwd: null pointer");
  DPC_CHECK_ARG(B > 0 && F > 0 && count > 0.0, "bn_relu_bwd: bad shape");
  hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(dpc_grid(B * F, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y,
                     ldy, z, ldz, B, F, mean, invstd, gamma, dbeta, dgamma, (float)(1.0 / count), dz, lddz);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm + ReLU with the reduction's finish inside: a block owns a 64-column strip of a row chunk,
// first adds the R row-chunk partials of its columns (what colreduce_finish_kernel does, same order:
// lane rl takes chunks rl, rl+4, ..), then normalises its rows.  One kernel node less on the step's
// critical path per BatchNorm (a node costs ~5 us whatever it does); the partials are re-read by
// every row chunk's block from L2 (R*2*64 doubles = 64 KB per block at T1M).
// Blocks of row chunk 0 write mean / invstd (the backward needs them) and update the running statistics.
// ---------------------------------------------------------------------------------------------
struct BnFusedArgs {
  const float* z; int64_t ldz; float* y; int64_t ldy; int64_t B; int F;
  const double* part; int R; double count; float eps, momentum;
  const float* gamma; const float* beta;
  float* mean_out; float* invstd_out; float* run_mean; float* run_var;
  int64_t rows_per_block;
};

__device__ __forceinline__ void bn_strip_sums(const double* __restrict__ part, int R, int F, int f, int c, int rl,
                                              double (*red)[4][64], double& s0, double& s1) {
  s0 = 0.0; s1 = 0.0;
  if (f < F) {
#pragma unroll 4
    for (int r = rl; r < R; r += 4) {
      s0 += part[((int64_t)r * 2 + 0) * F + f];
      s1 += part[((int64_t)r * 2 + 1) * F + f];
    }
  }
  red[0][rl][c] = s0;
  red[1][rl][c] = s1;
  __syncthreads();
  s0 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  s1 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
}

// VEC: F % 4 == 0 and 16-byte aligned rows - the row walk then moves four columns per thread (16 threads per row of
// the strip, 16 rows per pass) with the per-column constants handed over through LDS; same arithmetic per element.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_relu_fwd_fused_kernel(BnFusedArgs a) {
  __shared__ double red[2][4][64];
  __shared__ __attribute__((aligned(16))) float cst[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + c;
  double s0, s1;
  bn_strip_sums(a.part, a.R, a.F, f, c, rl, red, s0, s1);
  float mean = 0.f, invstd = 0.f, ga = 0.f, be = 0.f;
  if (f < a.F) {
    const bool writer = blockIdx.y == 0 && rl == 0;
    bn_stats(s0, s1, a.count, a.eps, mean, invstd, writer ? a.run_mean : nullptr, a.run_var, f, a.momentum);
    if (writer) {
      a.mean_out[f] = mean;
      a.invstd_out[f] = invstd;
    }
    ga = a.gamma[f];
    be = a.beta[f];
  }
  const int64_t r0 = (int64_t)blockIdx.y * a.rows_per_block, r1 = min(a.B, r0 + a.rows_per_block);
  if (VEC) {
    if (rl == 0) { cst[0][c] = mean; cst[1][c] = invstd; cst[2][c] = ga; cst[3][c] = be; }
    __syncthreads();
    const int c4 = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;
    const int f4 = blockIdx.x * 64 + c4;
    if (f4 >= a.F) return;
    const float4 m4 = *reinterpret_cast<const float4*>(&cst[0][c4]), i4 = *reinterpret_cast<const float4*>(&cst[1][c4]);
    const float4 g4 = *reinterpret_cast<const float4*>(&cst[2][c4]), b4 = *reinterpret_cast<const float4*>(&cst[3][c4]);
#pragma unroll 4
    for (int64_t r = r0 + rr; r < r1; r += 16) {
      const float4 z = *reinterpret_cast<const float4*>(a.z + r * a.ldz + f4);
      float4 y;
      y.x = fmaxf((z.x - m4.x) * i4.x * g4.x + b4.x, 0.f);
      y.y = fmaxf((z.y - m4.y) * i4.y * g4.y + b4.y, 0.f);
      y.z = fmaxf((z.z - m4.z) * i4.z * g4.z + b4.z, 0.f);
      y.w = fmaxf((z.w - m4.w) * i4.w * g4.w + b4.w, 0.f);
      *reinterpret_cast<float4*>(a.y + r * a.ldy + f4) = y;
    }
    return;
  }
  if (f >= a.F) return;
#pragma unroll 8
  for (int64_t r = r0 + rl; r < r1; r += 4) {
    const float v = (a.z[r * a.ldz + f] - mean) * invstd * ga + be;
    a.y[r * a.ldy + f] = fmaxf(v, 0.f);
  }
}

// part: the R x 2 x F partial sums of dpc_colreduce_partial mode 0 (all-reduced over ranks if any).
extern "C" int dpc_bn_relu_fwd_fused(const float* z, int64_t ldz, float* y, int64_t ldy, int64_t B, int F,
                                     const double* part, int R, double count, float eps, float momentum,
                                     const float* gamma, const float* beta, float* mean_out, float* invstd_out,
                                     float* run_mean, float* run_var, void* stream) {
  DPC_CHECK_ARG(z && y && part && gamma && beta && mean_out && invstd_out, "bn_relu_fwd_fused: null pointer");
  DPC_CHECK_ARG(B > 0 && F > 0 && ldz >= F && ldy >= F && R >= 1 && count > 0.0, "bn_relu_fwd_fused: bad shape");
  DPC_CHECK_ARG(!run_mean == !run_var, "bn_relu_fwd_fused: running mean and variance go together");
  BnFusedArgs a;
  a.z = z; a.ldz = ldz; a.y = y; a.ldy = ldy; a.B = B; a.F = F; a.part = part; a.R = R; a.count = count; a.eps = eps;
  a.momentum = momentum; a.gamma = gamma; a.beta = beta; a.mean_out = mean_out; a.invstd_out = invstd_out;
  a.run_mean = run_mean; a.run_var = run_var;
  const int64_t strips = dpc_cdiv(F, 64);
  int64_t chunks = dpc_cdiv(1024, strips);            // ~4 blocks per CU (256 measured slower: long serial row walks)
  if (chunks > dpc_cdiv(B, 32)) chunks = dpc_cdiv(B, 32);
  if (chunks < 1) chunks = 1;
  a.rows_per_block = dpc_cdiv(B, chunks);
  chunks = dpc_cdiv(B, a.rows_per_block);
  const bool vec = (F % 4 == 0) && (ldz % 4 == 0) && (ldy % 4 == 0) && ((((uintptr_t)z | (uintptr_t)y) & 15) == 0);
  if (vec) hipLaunchKernelGGL(bn_relu_fwd_fused_kernel<true>, dim3((unsigned)strips, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bn_relu_fwd_fused_kernel<false>, dim3((unsigned)strips, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

struct BnBwdFusedArgs {
  const float* dy; int64_t lddy; const float* y; int64_t ldy; const float* z; int64_t ldz; int64_t B; int F;
  const double* part; int R; double count;
  const float* mean; const float* invstd; const float* gamma;
  float* dbeta; float* dgamma; float* dz; int64_t lddz;
  int64_t rows_per_block;
  float out_scale;     // dbeta / dgamma are written times this (1/world: they are global sums already, see the entry point)
};

template <bool VEC>      // (as the forward: four columns per thread in the row walk)
__global__ __launch_bounds__(256) void bn_relu_bwd_fused_kernel(BnBwdFusedArgs a) {
  __shared__ double red[2][4][64];
  __shared__ __attribute__((aligned(16))) float cst[5][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + c;
  double s0, s1;
  bn_strip_sums(a.part, a.R, a.F, f, c, rl, red, s0, s1);
  const float db = (float)s0, dg = (float)s1;
  float mu = 0.f, is = 0.f, ga = 0.f;
  if (f < a.F) {
    if (blockIdx.y == 0 && rl == 0) {
      a.dbeta[f] = db * a.out_scale;
      a.dgamma[f] = dg * a.out_scale;
    }
    mu = a.mean[f]; is = a.invstd[f]; ga = a.gamma[f];
  }
  const float inv_count = (float)(1.0 / a.count);
  const int64_t r0 = (int64_t)blockIdx.y * a.rows_per_block, r1 = min(a.B, r0 + a.rows_per_block);
  if (VEC) {
    if (rl == 0) { cst[0][c] = mu; cst[1][c] = is; cst[2][c] = ga; cst[3][c] = db; cst[4][c] = dg; }
    __syncthreads();
    const int c4 = (threadIdx.x & 15) * 4, rr = threadIdx.x >> 4;
    const int f4 = blockIdx.x * 64 + c4;
    if (f4 >= a.F) return;
    float mu4[4], is4[4], ga4[4], db4[4], dg4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mu4[e] = cst[0][c4 + e]; is4[e] = cst[1][c4 + e]; ga4[e] = cst[2][c4 + e]; db4[e] = cst[3][c4 + e]; dg4[e] = cst[4][c4 + e];
    }
#pragma unroll 2
    for (int64_t r = r0 + rr; r < r1; r += 16) {
      const float4 y = *reinterpret_cast<const float4*>(a.y + r * a.ldy + f4);
      const float4 dy = *reinterpret_cast<const float4*>(a.dy + r * a.lddy + f4);
      const float4 z = *reinterpret_cast<const float4*>(a.z + r * a.ldz + f4);
      const float yv[4] = {y.x, y.y, y.z, y.w}, dv[4] = {dy.x, dy.y, dy.z, dy.w}, zv[4] = {z.x, z.y, z.z, z.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dyh = yv[e] > 0.f ? dv[e] : 0.f;
        const float xh = (zv[e] - mu4[e]) * is4[e];
        o[e] = ga4[e] * is4[e] * (dyh - db4[e] * inv_count - xh * dg4[e] * inv_count);
      }
      *reinterpret_cast<float4*>(a.dz + r * a.lddz + f4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  if (f >= a.F) return;
#pragma unroll 4
  for (int64_t r = r0 + rl; r < r1; r += 4) {
    const float dyh = a.y[r * a.ldy + f] > 0.f ? a.dy[r * a.lddy + f] : 0.f;
    const float xh = (a.z[r * a.ldz + f] - mu) * is;
    a.dz[r * a.lddz + f] = ga * is * (dyh - db * inv_count - xh * dg * inv_count);
  }
}

// part: the R x 2 x F partial sums of dpc_colreduce_partial mode 2 (all-reduced over ranks if any);
// writes dbeta, dgamma (= the finished sums) and dz as dpc_colreduce_finish kind 2 + dpc_bn_relu_bwd do.
// grad_scale multiplies the dbeta / dgamma that are WRITTEN (dz uses the unscaled sums): a data-parallel
// caller passes 1/world, because the sums are global already and the flat gradient buffer they land in
// is summed over ranks afterwards.
extern "C" int dpc_bn_relu_bwd_fused(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* z,
                                     int64_t ldz, int64_t B, int F, const double* part, int R, double count,
                                     const float* mean, const float* invstd, const float* gamma, float* dbeta,
                                     float* dgamma, float* dz, int64_t lddz, float grad_scale, void* stream) {
  DPC_CHECK_ARG(dy && y && z && part && mean && invstd && gamma && dbeta && dgamma && dz, "bn_relu_bwd_fused: null pointer");
  DPC_CHECK_ARG(B > 0 && F > 0 && R >= 1 && count > 0.0, "bn_relu_bwd_fused: bad shape");
  BnBwdFusedArgs a;
  a.dy = dy; a.lddy = lddy; a.y = y; a.ldy = ldy; a.z = z; a.ldz = ldz; a.B = B; a.F = F; a.part = part; a.R = R;
  a.count = count; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.dbeta = dbeta; a.dgamma = dgamma; a.dz = dz;
  a.lddz = lddz; a.out_scale = grad_scale;
  const int64_t strips = dpc_cdiv(F, 64);
  int64_t chunks = dpc_cdiv(1024, strips);
  if (chunks > dpc_cdiv(B, 32)) chunks = dpc_cdiv(B, 32);
  if (chunks < 1) chunks = 1;
  a.rows_per_block = dpc_cdiv(B, chunks);
  chunks = dpc_cdiv(B, a.rows_per_block);
  const bool vec = (F % 4 == 0) && (lddy % 4 == 0) && (ldy % 4 == 0) && (ldz % 4 == 0) && (lddz % 4 == 0) &&
                   ((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)z | (uintptr_t)dz) & 15) == 0);
  if (vec) hipLaunchKernelGGL(bn_relu_bwd_fused_kernel<true>, dim3((unsigned)strips, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(bn_relu_bwd_fused_kernel<false>, dim3((unsigned)strips, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Discriminator head: y[r] = h[r,:] . w + b.  One wave per row.  When `delta` is given the same
// pass also writes the seed of the downward chain, delta[r,j] = rowscale[r] * w[j] * slope[r,j]
// (what dpc_disc_seed computes), saving a launch per discriminator pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disc_out_kernel(const float* __restrict__ h, int64_t ldh,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       int64_t rows, int d, float* __restrict__ y,
                                                       const float* __restrict__ rowscale,
                                                       const float* __restrict__ slope, int64_t lds,
                                                       float* __restrict__ delta, int64_t ldd) {
  const int lane = threadIdx.x & 63;
  int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  float acc = 0.f;
  const float rs = delta ? rowscale[r] : 0.f;
  for (int j = lane; j < d; j += 64) {
    float wj = w[j];
    acc += h[r * ldh + j] * wj;
    if (delta) delta[r * ldd + j] = rs * wj * slope[r * lds + j];
  }
  acc = dpc_wave_sum(acc);
  if (lane == 0) y[r] = acc + b[0];
}

extern "C" int dpc_disc_out_fwd(const float* h, int64_t ldh, const float* w, const float* b, int64_t rows, int d,
                                float* y, const float* rowscale, const float* slope, int64_t lds, float* delta,
                                int64_t ldd, void* stream) {
  DPC_CHECK_ARG(h && w && b && y && rows > 0 && d > 0, "disc_out_fwd: bad argument");
  DPC_CHECK_ARG(!delta || (rowscale && slope && lds >= d && ldd >= d), "disc_out_fwd: seed output needs rowscale and slope");
  hipLaunchKernelGGL(disc_out_kernel, dim3((unsigned)dpc_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, h, ldh, w, b,
                     rows, d, y, rowscale, slope, lds, delta, ldd);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Seed of the downward chain: delta[r,j] = rowscale[r] * w[j] * slope[r,j].
// With rowscale = +1/n (fake rows), -1/n (real rows), 1 (penalty rows) one chain serves both the
// first-order backward of loss_d and the input-gradient ("g") chain of the penalty.
__global__ __launch_bounds__(256) void disc_seed_kernel(const float* __restrict__ rowscale, const float* __restrict__ w,
                                                        const float* __restrict__ slope, int64_t lds, int64_t rows,
                                                        int d, float* __restrict__ delta, int64_t ldd) {
  int64_t total = rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / d;
    int j = (int)(i - r * d);
    delta[r * ldd + j] = rowscale[r] * w[j] * slope[r * lds + j];
  }
}

extern "C" int dpc_disc_seed(const float* rowscale, const float* w, const float* slope, int64_t lds, int64_t rows, int d,
                             float* delta, int64_t ldd, void* stream) {
  DPC_CHECK_ARG(rowscale && w && slope && delta && rows > 0 && d > 0, "disc_seed: bad argument");
  hipLaunchKernelGGL(disc_seed_kernel, dim3(dpc_grid(rows * d, 256)), dim3(256), 0, (hipStream_t)stream, rowscale, w,
                     slope, lds, rows, d, delta, ldd);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// xhat rows of the stacked discriminator input: out = a*real + (1-a)*fake with one a per pac group
// (pac consecutive rows of length rowlen).  alpha_inj[g] or Philox draw g; alpha_out optional.
// One grid row (blockIdx.y) per pac group: alpha is drawn once per thread, not once per element, and
// the rows move as 16-byte vectors when the group length allows.
template <bool VEC>
__global__ __launch_bounds__(256) void interpolate_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                          float* __restrict__ out, int64_t n, int64_t group_len,
                                                          const float* __restrict__ alpha_inj, const uint64_t* rng,
                                                          uint32_t site, float* alpha_out) {
  for (int64_t g = blockIdx.y; g < n; g += gridDim.y) {
    const float a = alpha_inj ? alpha_inj[g] : dpc_u32_to_unit(dpc_draw_u32(rng, site, (uint64_t)g));
    const int64_t base = g * group_len;
    if (VEC) {
      const float4* r4 = reinterpret_cast<const float4*>(real + base);
      const float4* f4 = reinterpret_cast<const float4*>(fake + base);
      float4* o4 = reinterpret_cast<float4*>(out + base);
      for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < (group_len >> 2); j += (int64_t)gridDim.x * blockDim.x) {
        float4 r = r4[j], f = f4[j], o;
        o.x = a * r.x + (1.f - a) * f.x;
        o.y = a * r.y + (1.f - a) * f.y;
        o.z = a * r.z + (1.f - a) * f.z;
        o.w = a * r.w + (1.f - a) * f.w;
        o4[j] = o;
      }
    } else {
      for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < group_len; j += (int64_t)gridDim.x * blockDim.x)
        out[base + j] = a * real[base + j] + (1.f - a) * fake[base + j];
    }
    if (alpha_out && blockIdx.x == 0 && threadIdx.x == 0) alpha_out[g] = a;
  }
}

extern "C" int dpc_interpolate(const float* real, const float* fake, float* out, int64_t n, int64_t group_len,
                               const float* alpha_inj, const uint64_t* rng, uint32_t site, float* alpha_out,
                               void* stream) {
  DPC_CHECK_ARG(real && fake && out && n > 0 && group_len > 0, "interpolate: bad argument");
  DPC_CHECK_ARG(alpha_inj || rng, "interpolate: no random source");
  const bool vec = (group_len % 4 == 0) && ((((uintptr_t)real | (uintptr_t)fake | (uintptr_t)out) & 15) == 0);
  const int64_t per_row = vec ? group_len / 4 : group_len;
  int64_t gx = dpc_cdiv(per_row, 256);
  if (gx > 8) gx = 8;
  dim3 grid((unsigned)gx, (unsigned)(n < 65535 ? n : 65535));
  if (vec)
    hipLaunchKernelGGL(interpolate_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, real, fake, out, n, group_len,
                       alpha_inj, rng, site, alpha_out);
  else
    hipLaunchKernelGGL(interpolate_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, real, fake, out, n, group_len,
                       alpha_inj, rng, site, alpha_out);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Gradient-penalty row pass, in place: g0[r,:] -> v0[r,:] = coef * (nrm-1)/nrm * g0[r,:],
// coef = 2*lambda/n_total; pen_rows[r] = (nrm-1)^2; nrm_out[r] = nrm.  One block per row.
// One block per row.  VEC: the row (up to 8 float4 per thread = 8192 floats) stays in registers
// between the norm and the scaling pass, all loads issued before the first use.
template <bool VEC>
__global__ __launch_bounds__(256) void gp_norm_kernel(float* __restrict__ g0, int64_t ld, int64_t n, int K0, float coef,
                                                      double* __restrict__ pen_rows, float* __restrict__ nrm_out) {
  __shared__ double red[16];
  for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
    float* row = g0 + r * ld;
    double acc = 0.0;
    float4 c[8];
    if (VEC) {
      const float4* r4 = reinterpret_cast<const float4*>(row);
      const int n4 = K0 >> 2;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = threadIdx.x + q * 256;
        c[q] = j < n4 ? r4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        acc += (double)c[q].x * (double)c[q].x + (double)c[q].y * (double)c[q].y + (double)c[q].z * (double)c[q].z +
               (double)c[q].w * (double)c[q].w;
    } else {
      for (int j = threadIdx.x; j < K0; j += blockDim.x) {
        float v = row[j];
        acc += (double)v * (double)v;
      }
    }
    double tot = dpc_block_sum(acc, red);
    float nrm = (float)sqrt(tot);
    float f = nrm > 0.f ? coef * (nrm - 1.f) / nrm : 0.f;
    if (VEC) {
      float4* r4 = reinterpret_cast<float4*>(row);
      const int n4 = K0 >> 2;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = threadIdx.x + q * 256;
        if (j < n4) r4[j] = make_float4(c[q].x * f, c[q].y * f, c[q].z * f, c[q].w * f);
      }
    } else {
      for (int j = threadIdx.x; j < K0; j += blockDim.x) row[j] *= f;
    }
    if (threadIdx.x == 0) {
      pen_rows[r] = (double)(nrm - 1.f) * (double)(nrm - 1.f);
      if (nrm_out) nrm_out[r] = nrm;
    }
    __syncthreads();
  }
}

extern "C" int dpc_gp_norm(float* g0, int64_t ld, int64_t n, int K0, float coef, double* pen_rows, float* nrm_out,
                           void* stream) {
  DPC_CHECK_ARG(g0 && pen_rows && n > 0 && K0 > 0 && ld >= K0, "gp_norm: bad argument");
  const bool vec = (K0 % 4 == 0) && (K0 <= 8192) && (ld % 4 == 0) && (((uintptr_t)g0 & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(gp_norm_kernel<true>, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, (hipStream_t)stream, g0,
                       ld, n, K0, coef, pen_rows, nrm_out);
  else
    hipLaunchKernelGGL(gp_norm_kernel<false>, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, (hipStream_t)stream, g0,
                       ld, n, K0, coef, pen_rows, nrm_out);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Discriminator update, first hidden layer over the stacked batch [fake | real | interpolates]
// without ever forming the interpolates (dp_cgan.py:47-53): the layer is linear before its
// activation, so with one alpha per pac group (= per row of the [n, K0] view)
//     (a*real + (1-a)*fake) W^T = a * (real W^T) + (1-a) * (fake W^T).
// zpre [2n, d] holds the plain products of the fake rows (0..n-1) and the real rows (n..2n-1);
// this pass writes, for all 3n stacked rows p*n + r, the Linear + LeakyReLU + Dropout(0.5) output
// H and the slope S = (pre > 0 ? 1 : leak) * keep * 2 exactly as the GEMM's fused epilogue does
// (keep bit: keep[row, col] or Philox draw row*d + col of site_keep; alpha: alpha_inj[r] or draw r
// of site_alpha - the same index maps as dpc_gemm_f32 / dpc_interpolate).
// ---------------------------------------------------------------------------------------------
struct FirstActArgs {
  const float* zpre; int64_t ldz; const float* bias; int64_t n; int d;
  const float* alpha_inj; const uint8_t* keep; int64_t ld_keep;
  const uint64_t* rng; uint32_t site_alpha, site_keep; float leak;
  float* H; int64_t ldh; float* S; int64_t lds; float* alpha_out;
  int stacked;      // 1: zpre [2n, d] -> 3n stacked rows (above); 0: zpre [n, d] -> n rows, no interpolates
  float* zero;      // == zpre (mutable) when the products are to be cleared once read (see the entry point)
  // cond columns taken off the matrix cores (gat != NULL): zpre holds the products over the DATA columns of
  // the pac-row input only, and the two-hot cond vector of each of a row's pac sub-rows adds the matching
  // rows of gat = W1[:, cond columns]^T [pac*C][ldg]: sub-row s with hot id c -> row s*C + c.  sel_a: ids of
  // the first block of rows (fake), sel_b: of the second (real; stacked only), int32 [n*pac][2], negative = none.
  const float* gat; int64_t ldg; const int32_t* sel_a; const int32_t* sel_b; int pac; int C;
  float* wb;        // == zpre when the gathered share is to be added to the stored products (they are kept for
                    // the per-example Gram kernel, which needs x W1^T over ALL columns)
};

// sum over a row's pac sub-rows of the (up to two) hot rows of gat, 4 consecutive columns from j.  Branch-free
// and four sub-rows at a time: the 8 ids of a group are loaded together, then the 8 rows of gat (an absent id
// reads row 0 and adds +0) - with a branch around every load the 2 x pac loads of a row went one L2 round trip
// after the other.  Same summation order as the plain loop (sub-row, then hot id).
__device__ __forceinline__ float4 first_act_gather4(const FirstActArgs& a, const int32_t* __restrict__ sel, int64_t r, int j) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const int32_t* ids = sel + r * a.pac * 2;
  for (int s0 = 0; s0 < a.pac; s0 += 4) {
    int id[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int s = min(s0 + (q >> 1), a.pac - 1);
      id[q] = ids[2 * s + (q & 1)];
      if (s0 + (q >> 1) >= a.pac) id[q] = -1;
    }
    float4 w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int s = min(s0 + (q >> 1), a.pac - 1);
      w[q] = *reinterpret_cast<const float4*>(a.gat + ((int64_t)s * a.C + max(id[q], 0)) * a.ldg + j);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool on = id[q] >= 0;
      v.x += on ? w[q].x : 0.f; v.y += on ? w[q].y : 0.f; v.z += on ? w[q].z : 0.f; v.w += on ? w[q].w : 0.f;
    }
  }
  return v;
}

__device__ __forceinline__ float first_act_gather1(const FirstActArgs& a, const int32_t* __restrict__ sel, int64_t r, int j) {
  float v = 0.f;
  const int32_t* ids = sel + r * a.pac * 2;
  for (int s = 0; s < a.pac; ++s) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int id = ids[2 * s + h];
      if (id >= 0) v += a.gat[((int64_t)s * a.C + id) * a.ldg + j];
    }
  }
  return v;
}

template <bool VEC>
__global__ __launch_bounds__(256) void disc_first_act_kernel(FirstActArgs a) {
  constexpr int W = VEC ? 4 : 1;
  const int per_row = a.d / W;
  const int64_t total = a.n * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    const int j = (int)(i - r * per_row) * W;
    float al = 0.f;
    if (a.stacked) {
      al = a.alpha_inj ? a.alpha_inj[r] : dpc_u32_to_unit(dpc_draw_u32(a.rng, a.site_alpha, (uint64_t)r));
      if (a.alpha_out && j == 0) a.alpha_out[r] = al;
    }
    float zf[W], zr[W], b[W];
    if constexpr (VEC) {
      const float4 f4 = *reinterpret_cast<const float4*>(a.zpre + r * a.ldz + j);
      const float4 r4 = a.stacked ? *reinterpret_cast<const float4*>(a.zpre + (a.n + r) * a.ldz + j) : f4;
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + j);
      if (a.zero) {
        *reinterpret_cast<float4*>(a.zero + r * a.ldz + j) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.stacked) *reinterpret_cast<float4*>(a.zero + (a.n + r) * a.ldz + j) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      zf[0] = f4.x; zf[1] = f4.y; zf[2] = f4.z; zf[3] = f4.w;
      zr[0] = r4.x; zr[1] = r4.y; zr[2] = r4.z; zr[3] = r4.w;
      b[0] = b4.x; b[1] = b4.y; b[2] = b4.z; b[3] = b4.w;
      if (a.gat) {
        const float4 ga = first_act_gather4(a, a.sel_a, r, j);
        zf[0] += ga.x; zf[1] += ga.y; zf[2] += ga.z; zf[3] += ga.w;
        if (a.wb) *reinterpret_cast<float4*>(a.wb + r * a.ldz + j) = make_float4(zf[0], zf[1], zf[2], zf[3]);
        if (a.stacked) {
          const float4 gb = first_act_gather4(a, a.sel_b, r, j);
          zr[0] += gb.x; zr[1] += gb.y; zr[2] += gb.z; zr[3] += gb.w;
          if (a.wb) *reinterpret_cast<float4*>(a.wb + (a.n + r) * a.ldz + j) = make_float4(zr[0], zr[1], zr[2], zr[3]);
        }
      }
    } else {
      zf[0] = a.zpre[r * a.ldz + j];
      zr[0] = a.stacked ? a.zpre[(a.n + r) * a.ldz + j] : 0.f;
      b[0] = a.bias[j];
      if (a.zero) {
        a.zero[r * a.ldz + j] = 0.f;
        if (a.stacked) a.zero[(a.n + r) * a.ldz + j] = 0.f;
      }
      if (a.gat) {
        zf[0] += first_act_gather1(a, a.sel_a, r, j);
        if (a.wb) a.wb[r * a.ldz + j] = zf[0];
        if (a.stacked) {
          zr[0] += first_act_gather1(a, a.sel_b, r, j);
          if (a.wb) a.wb[(a.n + r) * a.ldz + j] = zr[0];
        }
      }
    }
    // (one call per stacked block, no unrolled loop with an early exit over arrays picked by the block index:
    // hipcc kept those in scratch memory - a store, a drained load queue and a load in every iteration)
    auto emit = [&](const int p, const float (&z)[W]) {
      const int64_t row = (int64_t)p * a.n + r;
      uint32_t bits[W];
      if (a.keep) {
#pragma unroll
        for (int e = 0; e < W; ++e) bits[e] = a.keep[row * a.ld_keep + j + e] ? 0x80000000u : 0u;
      } else if constexpr (VEC) {
        // d % 4 == 0: the four draws row*d + j .. + 3 are one Philox block
        DpcPhilox4 ph = dpc_philox(a.rng[0], a.rng[1], a.site_keep, ((uint64_t)row * a.d + j) >> 2);
        bits[0] = ph.v[0]; bits[1] = ph.v[1]; bits[2] = ph.v[2]; bits[3] = ph.v[3];
      } else {
        bits[0] = dpc_draw_u32(a.rng, a.site_keep, (uint64_t)row * a.d + j);
      }
      float h[W], s[W];
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const float pre = z[e] + b[e];
        s[e] = (pre > 0.f ? 1.f : a.leak) * ((bits[e] >> 31) ? 2.f : 0.f);
        h[e] = pre * s[e];
      }
      if constexpr (VEC) {
        *reinterpret_cast<float4*>(a.H + row * a.ldh + j) = make_float4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<float4*>(a.S + row * a.lds + j) = make_float4(s[0], s[1], s[2], s[3]);
      } else {
        a.H[row * a.ldh + j] = h[0];
        a.S[row * a.lds + j] = s[0];
      }
    };
    emit(0, zf);
    if (a.stacked) {
      emit(1, zr);
      float zi[W];
#pragma unroll
      for (int e = 0; e < W; ++e) zi[e] = al * zr[e] + (1.f - al) * zf[e];
      emit(2, zi);
    }
  }
}

extern "C" int dpc_disc_first_layer_act(float* zpre, int64_t ldz, const float* bias, int64_t n, int d,
                                        const float* alpha_inj, const uint8_t* keep, int64_t ld_keep,
                                        const uint64_t* rng, uint32_t site_alpha, uint32_t site_keep, float leak,
                                        float* H, int64_t ldh, float* S, int64_t lds, float* alpha_out, int stacked,
                                        int clear_zpre, const float* gat, int64_t ldg, const int32_t* sel_a,
                                        const int32_t* sel_b, int pac, int C, void* stream) {
  DPC_CHECK_ARG(zpre && bias && H && S && n > 0 && d > 0, "disc_first_layer_act: bad argument");
  DPC_CHECK_ARG(!gat || (sel_a && (sel_b || !stacked) && pac >= 1 && C >= 1 && ldg >= d), "disc_first_layer_act: cond gather needs ids, pac, C and ldg >= d");
  DPC_CHECK_ARG(ldz >= d && ldh >= d && lds >= d, "disc_first_layer_act: leading dimension < d");
  DPC_CHECK_ARG(((alpha_inj || !stacked) && keep) || rng, "disc_first_layer_act: no random source");
  DPC_CHECK_ARG(!keep || ld_keep >= d, "disc_first_layer_act: ld_keep < d");
  FirstActArgs a;
  a.zpre = zpre; a.ldz = ldz; a.bias = bias; a.n = n; a.d = d; a.alpha_inj = alpha_inj; a.keep = keep; a.ld_keep = ld_keep;
  a.rng = rng; a.site_alpha = site_alpha; a.site_keep = site_keep; a.leak = leak; a.H = H; a.ldh = ldh; a.S = S; a.lds = lds;
  a.alpha_out = alpha_out; a.stacked = stacked ? 1 : 0; a.zero = clear_zpre ? zpre : nullptr;
  a.gat = gat; a.ldg = ldg; a.sel_a = sel_a; a.sel_b = sel_b; a.pac = pac; a.C = C;
  a.wb = (gat && !clear_zpre) ? zpre : nullptr;
  const bool vec = (d % 4 == 0) && (ldz % 4 == 0) && (ldh % 4 == 0) && (lds % 4 == 0) &&
                   ((((uintptr_t)zpre | (uintptr_t)bias | (uintptr_t)H | (uintptr_t)S) & 15) == 0) &&
                   (!gat || ((ldg % 4 == 0) && (((uintptr_t)gat & 15) == 0)));
  if (vec)
    hipLaunchKernelGGL(disc_first_act_kernel<true>, dim3(dpc_grid(n * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(disc_first_act_kernel<false>, dim3(dpc_grid(n * d, 256)), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Gradient penalty through the first layer's Gram matrix G = W1 W1^T (d x d) instead of the K0-wide
// products (calc_gradient_penalty, dp_cgan.py:55-62, analytic form of SURVEY a4).  With u = the
// penalty rows of the first layer's backward delta and T = u G:
//     ||grad_xhat D||^2 = ||u W1||^2 = u G u^T = rowsum(T * u)                       (g0 is never formed)
//     v0 = f * (u W1),  f = coef * (nrm - 1) / nrm   =>   v1 = (v0 W1^T) * S1 = f * T * S1
// One wave per row.  Outputs: pen_rows[r] = (nrm-1)^2, f_out[r] = f, q_out[r] = ||g0||^2 (fp64),
// v1[r,:] (the penalty rows of the first hidden layer's v-chain).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_gram_rows_kernel(const float* __restrict__ T, int64_t ldt,
                                                           const float* __restrict__ u, int64_t ldu,
                                                           const float* __restrict__ slope, int64_t lds, int64_t n,
                                                           int d, float coef, double* __restrict__ pen_rows,
                                                           float* __restrict__ f_out, double* __restrict__ q_out,
                                                           float* __restrict__ nrm_out, float* __restrict__ v1,
                                                           int64_t ldv) {
  const int lane = threadIdx.x & 63;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
    double acc = 0.0;
    for (int j = lane; j < d; j += 64) acc += (double)T[r * ldt + j] * (double)u[r * ldu + j];
    double q = dpc_wave_sum(acc);
    if (q < 0.0) q = 0.0;
    const float nrm = (float)sqrt(q);
    const float f = nrm > 0.f ? coef * (nrm - 1.f) / nrm : 0.f;
    for (int j = lane; j < d; j += 64) v1[r * ldv + j] = f * T[r * ldt + j] * slope[r * lds + j];
    if (lane == 0) {
      pen_rows[r] = (double)(nrm - 1.f) * (double)(nrm - 1.f);
      f_out[r] = f;
      if (q_out) q_out[r] = q;
      if (nrm_out) nrm_out[r] = nrm;
    }
  }
}

extern "C" int dpc_gp_gram_rows(const float* T, int64_t ldt, const float* u, int64_t ldu, const float* slope,
                                int64_t lds, int64_t n, int d, float coef, double* pen_rows, float* f_out,
                                double* q_out, float* nrm_out, float* v1, int64_t ldv, void* stream) {
  DPC_CHECK_ARG(T && u && slope && pen_rows && f_out && v1 && n > 0 && d > 0, "gp_gram_rows: bad argument");
  DPC_CHECK_ARG(ldt >= d && ldu >= d && lds >= d && ldv >= d, "gp_gram_rows: leading dimension < d");
  int64_t blocks = dpc_cdiv(n, 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gp_gram_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, T, ldt, u, ldu, slope,
                     lds, n, d, coef, pen_rows, f_out, q_out, nrm_out, v1, ldv);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// out[0] = scale * sum(v[0:n]) (+ out[0] if accumulate); fixed order, one block.  v is fp64.
__global__ void sum_f64_kernel(const double* __restrict__ v, int64_t n, double scale, float* out, int accumulate) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
  double tot = dpc_block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = (float)(tot * scale) + (accumulate ? out[0] : 0.f);
}

extern "C" int dpc_sum_f64(const double* v, int64_t n, double scale, float* out, int accumulate, void* stream) {
  DPC_CHECK_ARG(v && out && n > 0, "sum_f64: bad argument");
  hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, n, scale, out, accumulate);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// out[0] = sa * mean(y[0:n]) + sb * mean(y[n:2n]) (second term skipped when sb == 0) (+ out[0]).
__global__ void mean_pair_kernel(const float* __restrict__ y, int64_t n, float sa, float sb, float* out, int accumulate) {
  __shared__ double red[16];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    a += (double)y[i];
    if (sb != 0.f) b += (double)y[n + i];
  }
  double ta = dpc_block_sum(a, red);
  double tb = dpc_block_sum(b, red);
  if (threadIdx.x == 0) out[0] = (float)(sa * ta / (double)n + sb * tb / (double)n) + (accumulate ? out[0] : 0.f);
}

extern "C" int dpc_mean_pair(const float* y, int64_t n, float sa, float sb, float* out, int accumulate, void* stream) {
  DPC_CHECK_ARG(y && out && n > 0, "mean_pair: bad argument");
  hipLaunchKernelGGL(mean_pair_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, y, n, sa, sb, out, accumulate);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
