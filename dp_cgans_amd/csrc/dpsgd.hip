// DP-SGD pieces of the discriminator update (north_star; no counterpart in the reference, whose
// private=True is the grad-hook noise of dp_cgan.py:638-647 - that one is dpc_hook_noise below).
//
// An example is one pac group i.  Its loss  l_i = D(fake_i) - D(real_i) + lambda (||grad D(xhat_i)|| - 1)^2
// touches three rows of the stacked discriminator batch: i (fake), n+i (real), 2n+i (penalty).  For
// a layer with weight W the example's gradient is the sum of three outer products
//     G_i = sum_p  delta[p*n+i]^T  act[p*n+i]          p = 0,1,2
// (for the penalty row delta = u-chain, act = v-chain), so
//     ||G_i||_F^2 = sum_{p,q} (delta_p . delta_q)(act_p . act_q)
// i.e. two 3x3 Gram matrices per example and layer: the per-example gradient is never formed
// ("ghost norm").  Bias gradient of the example: delta_0 + delta_1 (the penalty has none).
// HBM-bound: every delta / act element is read once, 4*(3n)*(d + K) bytes per layer.
#include "dpc_common.h"

struct GramArgs {
  const float* delta; int64_t ldd; int d;
  const float* act; int64_t lda; int K;
  int64_t n; int with_bias; int accumulate;
  double* norm2;
};

// 6 unique entries of the symmetric 3x3 Gram of rows r0, r1, r2 (each `len` long): this thread's
// strided share.  VEC: 16-byte loads (rows 16-byte aligned, len % 4 == 0).
// (first, step): the positions this thread takes - threadIdx.x, blockDim.x for a whole block; lane, 64 for one wave.
template <bool VEC>
__device__ __forceinline__ void gram3_partial(const float* __restrict__ r0, const float* __restrict__ r1,
                                              const float* __restrict__ r2, int len, float g[6], const int first,
                                              const int step) {
#pragma unroll
  for (int q = 0; q < 6; ++q) g[q] = 0.f;
  if (VEC) {
    const float4* a = reinterpret_cast<const float4*>(r0);
    const float4* b = reinterpret_cast<const float4*>(r1);
    const float4* c = reinterpret_cast<const float4*>(r2);
    const int n4 = len >> 2, bd = step;
    auto fma6 = [&](const float4& x, const float4& y, const float4& z) {
      g[0] += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      g[1] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      g[2] += x.x * z.x + x.y * z.y + x.z * z.z + x.w * z.w;
      g[3] += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
      g[4] += y.x * z.x + y.y * z.y + y.z * z.z + y.w * z.w;
      g[5] += z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w;
    };
    int j = first;
    // four positions (12 loads) in flight per thread: with a few thousand examples per batch the
    // launch is a few blocks per CU and nothing else hides the memory latency
    for (; j + 3 * bd < n4; j += 4 * bd) {
      float4 x0 = a[j], y0 = b[j], z0 = c[j];
      float4 x1 = a[j + bd], y1 = b[j + bd], z1 = c[j + bd];
      float4 x2 = a[j + 2 * bd], y2 = b[j + 2 * bd], z2 = c[j + 2 * bd];
      float4 x3 = a[j + 3 * bd], y3 = b[j + 3 * bd], z3 = c[j + 3 * bd];
      fma6(x0, y0, z0);
      fma6(x1, y1, z1);
      fma6(x2, y2, z2);
      fma6(x3, y3, z3);
    }
    for (; j < n4; j += bd) {
      float4 x = a[j], y = b[j], z = c[j];
      fma6(x, y, z);
    }
  } else {
    for (int j = first; j < len; j += step) {
      float x = r0[j], y = r1[j], z = r2[j];
      g[0] += x * x; g[1] += x * y; g[2] += x * z; g[3] += y * y; g[4] += y * z; g[5] += z * z;
    }
  }
}

// This thread's share of x.x, x.y, y.y for two rows.
template <bool VEC>
__device__ __forceinline__ void gram2_partial(const float* __restrict__ r0, const float* __restrict__ r1, int len,
                                              float& xx, float& xy, float& yy, const int first, const int step) {
  xx = xy = yy = 0.f;
  if (VEC) {
    const float4* a = reinterpret_cast<const float4*>(r0);
    const float4* b = reinterpret_cast<const float4*>(r1);
    const int n4 = len >> 2, bd = step;
    auto fma3 = [&](const float4& x, const float4& y) {
      xx += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      xy += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      yy += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
    };
    int j = first;
    for (; j + 3 * bd < n4; j += 4 * bd) {      // 8 loads in flight per thread
      float4 x0 = a[j], y0 = b[j], x1 = a[j + bd], y1 = b[j + bd];
      float4 x2 = a[j + 2 * bd], y2 = b[j + 2 * bd], x3 = a[j + 3 * bd], y3 = b[j + 3 * bd];
      fma3(x0, y0);
      fma3(x1, y1);
      fma3(x2, y2);
      fma3(x3, y3);
    }
    for (; j < n4; j += bd) {
      float4 x = a[j], y = b[j];
      fma3(x, y);
    }
  } else {
    for (int j = first; j < len; j += step) {
      float x = r0[j], y = r1[j];
      xx += x * x; xy += x * y; yy += y * y;
    }
  }
}

// This thread's share of u.a and u.b (short rows: the first layer's width).
__device__ __forceinline__ void dot2_partial(const float* __restrict__ u, const float* __restrict__ a,
                                             const float* __restrict__ b, int len, float& ua, float& ub, const int first,
                                             const int step) {
  ua = ub = 0.f;
  for (int j = first; j < len; j += step) {
    float x = u[j];
    ua += x * a[j];
    ub += x * b[j];
  }
}

template <bool VEC_D, bool VEC_A>
__global__ __launch_bounds__(256) void persample_gram_kernel(GramArgs a) {
  __shared__ double red[4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t i = blockIdx.x; i < a.n; i += gridDim.x) {
    float g[12];
    gram3_partial<VEC_D>(a.delta + i * a.ldd, a.delta + (a.n + i) * a.ldd, a.delta + (2 * a.n + i) * a.ldd, a.d, g, threadIdx.x, blockDim.x);
    gram3_partial<VEC_A>(a.act + i * a.lda, a.act + (a.n + i) * a.lda, a.act + (2 * a.n + i) * a.lda, a.K, g + 6, threadIdx.x, blockDim.x);
    // wavefront-level reduction of the 12 partial dot products, then 4 waves through LDS
    double s[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) s[q] = dpc_wave_sum((double)g[q]);
    __syncthreads();                       // red[] of the previous example has been consumed
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 12; ++q) red[wave][q] = s[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) t[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
      // sum_pq (D D^T)_pq (A A^T)_pq, off-diagonal entries twice
      double v = t[0] * t[6] + t[3] * t[9] + t[5] * t[11] + 2.0 * (t[1] * t[7] + t[2] * t[8] + t[4] * t[10]);
      if (a.with_bias) v += t[0] + 2.0 * t[1] + t[3];       // ||delta_0 + delta_1||^2
      a.norm2[i] = a.accumulate ? a.norm2[i] + v : v;
    }
  }
}

// norm2[i] (+)= squared Frobenius norm of example i's gradient of one Linear layer (weight, and
// bias when with_bias).  delta [3n, d], act [3n, K]; rows i, n+i, 2n+i belong to example i.
extern "C" int dpc_persample_gram(const float* delta, int64_t ldd, int d, const float* act, int64_t lda, int K,
                                  int64_t n, int with_bias, int accumulate, double* norm2, void* stream) {
  DPC_CHECK_ARG(delta && act && norm2, "persample_gram: null pointer");
  DPC_CHECK_ARG(n > 0 && d > 0 && K > 0 && ldd >= d && lda >= K, "persample_gram: bad shape");
  GramArgs a;
  a.delta = delta; a.ldd = ldd; a.d = d; a.act = act; a.lda = lda; a.K = K; a.n = n; a.with_bias = with_bias;
  a.accumulate = accumulate; a.norm2 = norm2;
  bool vd = (((uintptr_t)delta & 15) == 0) && (ldd % 4 == 0) && (d % 4 == 0);
  bool va = (((uintptr_t)act & 15) == 0) && (lda % 4 == 0) && (K % 4 == 0);
  dim3 grid((unsigned)(n < 16384 ? n : 16384)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (vd && va) hipLaunchKernelGGL((persample_gram_kernel<true, true>), grid, block, 0, st, a);
  else if (vd) hipLaunchKernelGGL((persample_gram_kernel<true, false>), grid, block, 0, st, a);
  else if (va) hipLaunchKernelGGL((persample_gram_kernel<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((persample_gram_kernel<false, false>), grid, block, 0, st, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// All layers of the discriminator in one launch, clip factors included: block i walks the layers,
// adds their Gram contractions and writes c_i (what dpc_persample_gram per layer + dpc_clip_factors
// compute, in the same summation order).
#define DPC_MAX_LAYERS 8
struct GramAllArgs {
  const float* delta[DPC_MAX_LAYERS]; int64_t ldd[DPC_MAX_LAYERS]; int d[DPC_MAX_LAYERS];
  const float* act[DPC_MAX_LAYERS]; int64_t lda[DPC_MAX_LAYERS]; int K[DPC_MAX_LAYERS];
  int vec_d[DPC_MAX_LAYERS], vec_a[DPC_MAX_LAYERS];
  int nlayers; int64_t n;
  float clip, gscale;
  double* norm2; float* rowc; const float* roww; float* rowc_w; float* norm_out;
  // Implicit penalty row of layer 0 (zpre != NULL): act[0] holds only the fake and real rows [2n, K];
  // the third row of example i would be v0_i = f_i * (u_i W1), u_i = delta[0] row 2n+i, and its inner
  // products follow from the first layer's plain products zpre = x W1^T [2n, d] without forming it:
  //   fake_i . v0_i = f_i * (u_i . zpre_i),  real_i . v0_i = f_i * (u_i . zpre_{n+i}),  v0_i . v0_i = f_i^2 q_i
  const float* zpre; int64_t ldz; const float* gp_f; const double* gp_q; float* fc_out;
  float* zero;     // == zpre when its rows are to be cleared once read (the next step's split-K product adds into them)
  // cond columns kept out of act[0] (sel_a != NULL): act[0] rows hold the DATA columns of the pac-row input
  // only (zero padded), and the inner products of the two-hot cond parts are counted from the hot ids:
  // sel_a / sel_b int32 [n*pac][2] ids of the fake / real sub-rows (negative = none).
  const int32_t* sel_a; const int32_t* sel_b; int pac;
};

// |A|, |B| and |A n B| summed over the pac sub-rows of example i, A / B = the sets of hot ids of a sub-row: this
// lane's share (sub-rows first, first + step, ..)
__device__ __forceinline__ void cond_gram_counts(const GramAllArgs& a, int64_t i, float& aa, float& ab, float& bb,
                                                 const int first, const int step) {
  int naa = 0, nab = 0, nbb = 0;
  for (int s = first; s < a.pac; s += step) {
    const int64_t q = (i * a.pac + s) * 2;
    const int a0 = a.sel_a[q], a1 = a.sel_a[q + 1], b0 = a.sel_b[q], b1 = a.sel_b[q + 1];
    const bool va0 = a0 >= 0, va1 = a1 >= 0 && a1 != a0, vb0 = b0 >= 0, vb1 = b1 >= 0 && b1 != b0;
    naa += (int)va0 + (int)va1;
    nbb += (int)vb0 + (int)vb1;
    nab += (int)(va0 && ((vb0 && a0 == b0) || (vb1 && a0 == b1))) + (int)(va1 && ((vb0 && a1 == b0) || (vb1 && a1 == b1)));
  }
  aa = (float)naa; ab = (float)nab; bb = (float)nbb;       // (exact: counts far below 2^24)
}

// One block of GA_WAVES waves per example.  The example's work is a list of ITEMS - per layer the 3x3 Gram of its
// three delta rows and the one of its three input rows, plus (implicit penalty row) the two dot products with the
// first layer's plain products.  Most items are a few short rows (the layer widths: 64 16-byte loads per row at
// width 256), and a block that walked them one after the other - load, block-wide reduction, barrier, next layer -
// spent its time in latency chains with three waves out of four idle (27 us at T1M for 40 MB, all of it in L2 /
// Infinity Cache).  Here a short item belongs to ONE wave (item k to wave k % GA_WAVES: the items proceed side by
// side), a long one (>= GA_LONG floats per row: the K0-wide input rows) is strided over all waves; every wave
// leaves its wave-summed fp64 entries in LDS, ONE barrier, and thread 0 combines the layers.
// (GA_WAVES 8 - one item per wave, the whole chip's wave slots in one round - measured in round 3: 13 -> 19.7 us)
#define GA_WAVES 4
#define GA_LONG 1024
__global__ __launch_bounds__(64 * GA_WAVES) void persample_gram_all_kernel(GramAllArgs a) {
  __shared__ double red[2 * DPC_MAX_LAYERS + 2][GA_WAVES][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nitems = 2 * a.nlayers + (a.zpre ? 1 : 0) + (a.sel_a ? 1 : 0);       // (sel_a goes with zpre)
  for (int64_t i = blockIdx.x; i < a.n; i += gridDim.x) {
    if (i != (int64_t)blockIdx.x) __syncthreads();          // red[] of the previous example has been consumed
    for (int k = 0; k < nitems; ++k) {
      float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int l = k >> 1;
      const bool is_cnt = (k == 2 * a.nlayers + 1);
      const bool is_dot = (k == 2 * a.nlayers), is_act = !is_dot && !is_cnt && (k & 1);
      const int len = is_cnt ? a.pac : (is_dot ? a.d[0] : (is_act ? a.K[l] : a.d[l]));
      const bool spread = len >= GA_LONG;
      if (!spread && (k % GA_WAVES) != wave) continue;
      const int first = spread ? (int)threadIdx.x : lane, step = spread ? 64 * GA_WAVES : 64;
      if (is_cnt) {
        cond_gram_counts(a, i, g[0], g[1], g[3], first, step);
      } else if (is_dot) {
        const float* u = a.delta[0] + (2 * a.n + i) * a.ldd[0];
        dot2_partial(u, a.zpre + i * a.ldz, a.zpre + (a.n + i) * a.ldz, len, g[0], g[1], first, step);
        if (a.zero) {
          for (int j = first; j < len; j += step) {          // the elements this thread just read
            a.zero[i * a.ldz + j] = 0.f;
            a.zero[(a.n + i) * a.ldz + j] = 0.f;
          }
        }
      } else if (is_act) {
        const float* ac = a.act[l];
        const int64_t ld = a.lda[l];
        if (l == 0 && a.zpre != nullptr) {
          if (a.vec_a[l]) gram2_partial<true>(ac + i * ld, ac + (a.n + i) * ld, len, g[0], g[1], g[3], first, step);
          else gram2_partial<false>(ac + i * ld, ac + (a.n + i) * ld, len, g[0], g[1], g[3], first, step);
        } else if (a.vec_a[l]) gram3_partial<true>(ac + i * ld, ac + (a.n + i) * ld, ac + (2 * a.n + i) * ld, len, g, first, step);
        else gram3_partial<false>(ac + i * ld, ac + (a.n + i) * ld, ac + (2 * a.n + i) * ld, len, g, first, step);
      } else {
        const float* dl = a.delta[l];
        const int64_t ld = a.ldd[l];
        if (a.vec_d[l]) gram3_partial<true>(dl + i * ld, dl + (a.n + i) * ld, dl + (2 * a.n + i) * ld, len, g, first, step);
        else gram3_partial<false>(dl + i * ld, dl + (a.n + i) * ld, dl + (2 * a.n + i) * ld, len, g, first, step);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double sq = (double)dpc_wave_sum_dpp(g[q]);
        if (lane == 0) red[k][wave][q] = sq;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      auto item = [&](int k, int len, double* t) {           // the finished entries of item k
        if (len >= GA_LONG) {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            double sq = 0.0;
            for (int w = 0; w < GA_WAVES; ++w) sq += red[k][w][q];
            t[q] = sq;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) t[q] = red[k][k % GA_WAVES][q];
        }
      };
      double total = 0.0;
      for (int l = 0; l < a.nlayers; ++l) {
        double t[12];
        item(2 * l, a.d[l], t);
        item(2 * l + 1, a.K[l], t + 6);
        if (l == 0 && a.zpre != nullptr) {
          double z[6];
          item(2 * a.nlayers, a.d[0], z);
          if (a.sel_a) {
            double c[6];
            item(2 * a.nlayers + 1, a.pac, c);
            t[6] += c[0]; t[7] += c[1]; t[9] += c[3];
          }
          const double f = (double)a.gp_f[i];
          t[8] = z[0] * f;
          t[10] = z[1] * f;
          t[11] = f * f * a.gp_q[i];
        }
        double v = t[0] * t[6] + t[3] * t[9] + t[5] * t[11] + 2.0 * (t[1] * t[7] + t[2] * t[8] + t[4] * t[10]);
        v += t[0] + 2.0 * t[1] + t[3];                     // bias: ||delta_0 + delta_1||^2
        total += v;
      }
      a.norm2[i] = total;
      float nrm = a.gscale * (float)sqrt(total);
      float c = nrm > a.clip ? a.clip / nrm : 1.f;
      if (a.norm_out) a.norm_out[i] = nrm;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a.rowc[p * a.n + i] = c;
        if (a.rowc_w) a.rowc_w[p * a.n + i] = c * a.roww[p * a.n + i];
      }
      if (a.fc_out) a.fc_out[i] = c * a.gp_f[i];
    }
  }
}

// delta / act: host arrays of nlayers device pointers (each [3n, d_l] / [3n, K_l]) with their
// leading dimensions and widths.  Every layer counts its bias.
static int persample_clip_all_impl(int nlayers, const float* const* delta, const int64_t* ldd, const int* d,
                                   const float* const* act, const int64_t* lda, const int* K, int64_t n, float clip,
                                   float gscale, double* norm2, float* rowc, const float* roww, float* rowc_w,
                                   float* norm_out, float* zpre, int64_t ldz, const float* gp_f,
                                   const double* gp_q, float* fc_out, int clear_zpre, const int32_t* sel_a,
                                   const int32_t* sel_b, int pac, void* stream) {
  DPC_CHECK_ARG(nlayers >= 1 && nlayers <= DPC_MAX_LAYERS, "persample_clip_all: 1..8 layers");
  DPC_CHECK_ARG(delta && ldd && d && act && lda && K && norm2 && rowc && n > 0, "persample_clip_all: null pointer");
  DPC_CHECK_ARG(!fc_out || gp_f, "persample_clip_all: fc_out needs gp_f");
  DPC_CHECK_ARG(clip > 0.f && gscale > 0.f, "persample_clip_all: clip bound and scale must be positive");
  DPC_CHECK_ARG(!rowc_w || roww, "persample_clip_all: row weights missing");
  GramAllArgs a;
  for (int l = 0; l < nlayers; ++l) {
    DPC_CHECK_ARG(delta[l] && act[l] && d[l] > 0 && K[l] > 0 && ldd[l] >= d[l] && lda[l] >= K[l], "persample_clip_all: bad layer");
    a.delta[l] = delta[l]; a.ldd[l] = ldd[l]; a.d[l] = d[l]; a.act[l] = act[l]; a.lda[l] = lda[l]; a.K[l] = K[l];
    a.vec_d[l] = (((uintptr_t)delta[l] & 15) == 0) && (ldd[l] % 4 == 0) && (d[l] % 4 == 0);
    a.vec_a[l] = (((uintptr_t)act[l] & 15) == 0) && (lda[l] % 4 == 0) && (K[l] % 4 == 0);
  }
  a.nlayers = nlayers; a.n = n; a.clip = clip; a.gscale = gscale; a.norm2 = norm2; a.rowc = rowc; a.roww = roww;
  a.rowc_w = rowc_w; a.norm_out = norm_out;
  a.zpre = zpre; a.ldz = ldz; a.gp_f = gp_f; a.gp_q = gp_q; a.fc_out = fc_out; a.zero = clear_zpre ? zpre : nullptr;
  a.sel_a = sel_a; a.sel_b = sel_b; a.pac = pac;
  hipLaunchKernelGGL(persample_gram_all_kernel, dim3((unsigned)(n < 16384 ? n : 16384)), dim3(64 * GA_WAVES), 0,
                     (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

extern "C" int dpc_persample_clip_all(int nlayers, const float* const* delta, const int64_t* ldd, const int* d,
                                      const float* const* act, const int64_t* lda, const int* K, int64_t n, float clip,
                                      float gscale, double* norm2, float* rowc, const float* roww, float* rowc_w,
                                      float* norm_out, void* stream) {
  return persample_clip_all_impl(nlayers, delta, ldd, d, act, lda, K, n, clip, gscale, norm2, rowc, roww, rowc_w,
                                 norm_out, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 1, stream);
}

// The same with the penalty row of layer 0 left implicit (see GramAllArgs): act[0] is [2n, K[0]] (fake and
// real rows), zpre [2n, d[0]] the first layer's plain products, gp_f / gp_q what dpc_gp_gram_rows
// left behind.  fc_out[i] (nullable) = c_i * gp_f[i], the per-row scale of the penalty's Gram-form
// weight gradient P = u^T diag(fc) u.
extern "C" int dpc_persample_clip_all_gram(int nlayers, const float* const* delta, const int64_t* ldd, const int* d,
                                           const float* const* act, const int64_t* lda, const int* K, int64_t n,
                                           float clip, float gscale, double* norm2, float* rowc, const float* roww,
                                           float* rowc_w, float* norm_out, float* zpre, int64_t ldz,
                                           const float* gp_f, const double* gp_q, float* fc_out, int clear_zpre,
                                           const int32_t* sel_a, const int32_t* sel_b, int pac, void* stream) {
  DPC_CHECK_ARG(zpre && gp_f && gp_q, "persample_clip_all_gram: null pointer");
  DPC_CHECK_ARG(!sel_a || (sel_b && pac >= 1), "persample_clip_all_gram: cond ids need both tables and pac");
  DPC_CHECK_ARG(nlayers >= 1 && d && ldz >= d[0], "persample_clip_all_gram: ldz < d[0]");
  return persample_clip_all_impl(nlayers, delta, ldd, d, act, lda, K, n, clip, gscale, norm2, rowc, roww, rowc_w,
                                 norm_out, zpre, ldz, gp_f, gp_q, fc_out, clear_zpre, sel_a, sel_b, pac, stream);
}

// c_i = min(1, clip / (gscale * sqrt(norm2[i]))); rowc[p*n + i] = c_i for the three rows of example i
// (the per-row scale handed to the weight-gradient GEMM and the bias column sums), and
// rowc_w[r] = rowc[r] * roww[r] (the head layer's row weights).  norm_out[i] = gscale*sqrt(norm2[i]).
__global__ __launch_bounds__(256) void clip_factors_kernel(const double* __restrict__ norm2, int64_t n, float clip,
                                                           float gscale, float* __restrict__ rowc,
                                                           const float* __restrict__ roww,
                                                           float* __restrict__ rowc_w, float* __restrict__ norm_out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float nrm = gscale * (float)sqrt(norm2[i]);
    float c = nrm > clip ? clip / nrm : 1.f;
    if (norm_out) norm_out[i] = nrm;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      rowc[p * n + i] = c;
      if (rowc_w) rowc_w[p * n + i] = c * roww[p * n + i];
    }
  }
}

extern "C" int dpc_clip_factors(const double* norm2, int64_t n, float clip, float gscale, float* rowc,
                                const float* roww, float* rowc_w, float* norm_out, void* stream) {
  DPC_CHECK_ARG(norm2 && rowc && n > 0, "clip_factors: bad argument");
  DPC_CHECK_ARG(clip > 0.f && gscale > 0.f, "clip_factors: clip bound and scale must be positive");
  DPC_CHECK_ARG(!rowc_w || roww, "clip_factors: row weights missing");
  hipLaunchKernelGGL(clip_factors_kernel, dim3(dpc_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, norm2, n, clip,
                     gscale, rowc, roww, rowc_w, norm_out);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// The reference's private-mode hook noise (dp_cgan.py:638-647): every hook firing adds
// one scalar N(0,1)/B to the whole gradient tensor, and after t discriminator steps a tensor
// carries t+1 hook sets, each fired `fires[s]` times per step.  The sum of those independent
// scalars is N(0, fires[s]*(t+1)) / B:  seg_add[s] = z_s * sqrt(fires[s] * (t + t_add)) * inv_B
// (t = *t_dev discriminator updates done before this step, t_add = 1 + updates done inside it).
__global__ void hook_noise_kernel(float* seg_add, int nseg, const float* fires, const int32_t* t_dev, int t_add,
                                  float inv_B, const uint64_t* rng, uint32_t site) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  float t1 = (float)(t_dev[0] + t_add);
  seg_add[s] = dpc_draw_normal(rng, site, (uint64_t)s) * sqrtf(fires[s] * t1) * inv_B;
}

extern "C" int dpc_hook_noise(float* seg_add, int nseg, const float* fires, const int32_t* t_dev, int t_add,
                              float inv_B, const uint64_t* rng, uint32_t site, void* stream) {
  DPC_CHECK_ARG(seg_add && fires && t_dev && rng && nseg > 0 && t_add >= 1, "hook_noise: bad argument");
  hipLaunchKernelGGL(hook_noise_kernel, dim3((unsigned)dpc_cdiv(nseg, 64)), dim3(64), 0, (hipStream_t)stream, seg_add,
                     nseg, fires, t_dev, t_add, inv_B, rng, site);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
