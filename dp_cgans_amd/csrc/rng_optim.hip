// Philox fills (the same idx -> draw mapping the fused kernels use) and the fused
// noise + Adam + clamp parameter update.
#include "dpc_common.h"
#include <string.h>

static thread_local char g_err[256] = "";
extern "C" void dpc_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* dpc_last_error(void) { return g_err; }
extern "C" int dpc_version(void) { return 100; }

__global__ void rng_advance_kernel(uint64_t* rng_a, uint64_t* rng_b, int32_t* t_a, int add_a, int32_t* t_b, int add_b) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (rng_a) rng_a[1] += 1;
    if (rng_b) rng_b[1] += 1;
    if (t_a) t_a[0] += add_a;
    if (t_b) t_b[0] += add_b;
  }
}

// End-of-step bookkeeping in ONE launch: bump the step counter of up to two RNG states and add to up
// to two optimiser step counters (all device-resident so that a captured graph replays with fresh
// numbers).  Any pointer may be null.  Must run after every kernel of the step that reads them.
extern "C" int dpc_state_advance(uint64_t* rng_a, uint64_t* rng_b, int32_t* t_a, int add_a, int32_t* t_b, int add_b,
                                 void* stream) {
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng_a, rng_b, t_a, add_a, t_b,
                     add_b);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// kind: 0 raw u32 (as float bits are NOT used; written to uint32 out), 1 normal f32, 2 Exp(1) f32,
//       3 uniform(0,1) f32, 4 uniform [0,1) f64 (two draws: 2*idx, 2*idx+1), 5 keep bit (u8)
template <int KIND, typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* out, int64_t ld, int64_t rows, int64_t cols, const uint64_t* rng,
                                                   uint32_t site, float a, float b) {
  // one thread per group of draws that share a Philox block: 4 consecutive flat indices (2 for the
  // fp64 uniforms, which take two words each)
  constexpr int PER = (KIND == 4) ? 2 : 4;
  const int64_t n = rows * cols, ngrp = (n + PER - 1) / PER;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < ngrp; q += (int64_t)gridDim.x * blockDim.x) {
    T v[PER];
    if (KIND == 1) {
      float z[4];
      dpc_normal4(rng, site, (uint64_t)q, z);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (T)(a + b * z[j]);
    } else {
      DpcPhilox4 p = dpc_philox(rng[0], rng[1], site, (uint64_t)q);
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (KIND == 0) v[j] = (T)p.v[j];
        else if (KIND == 2) v[j] = (T)dpc_exp1(p.v[j]);
        else if (KIND == 3) v[j] = (T)dpc_u32_to_unit(p.v[j]);
        else if (KIND == 4) v[j] = (T)dpc_u64_to_unit(p.v[2 * j], p.v[2 * j + 1]);
        else v[j] = (T)(p.v[j] >> 31);
      }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int64_t i = q * PER + j;
      if (i < n) {
        int64_t r = i / cols, c = i - r * cols;
        out[r * ld + c] = v[j];
      }
    }
  }
}

extern "C" int dpc_fill(void* out, int kind, int64_t ld, int64_t rows, int64_t cols, const uint64_t* rng,
                        uint32_t site, float a, float b, void* stream) {
  DPC_CHECK_ARG(out && rng, "fill: null pointer");
  DPC_CHECK_ARG(rows >= 0 && cols >= 0 && ld >= cols, "fill: bad shape");
  if (rows * cols == 0) return DPC_OK;
  hipStream_t st = (hipStream_t)stream;
  int grid = dpc_grid((rows * cols + 3) / 4, 256);
  switch (kind) {
    case 0: hipLaunchKernelGGL((fill_kernel<0, uint32_t>), dim3(grid), dim3(256), 0, st, (uint32_t*)out, ld, rows, cols, rng, site, a, b); break;
    case 1: hipLaunchKernelGGL((fill_kernel<1, float>), dim3(grid), dim3(256), 0, st, (float*)out, ld, rows, cols, rng, site, a, b); break;
    case 2: hipLaunchKernelGGL((fill_kernel<2, float>), dim3(grid), dim3(256), 0, st, (float*)out, ld, rows, cols, rng, site, a, b); break;
    case 3: hipLaunchKernelGGL((fill_kernel<3, float>), dim3(grid), dim3(256), 0, st, (float*)out, ld, rows, cols, rng, site, a, b); break;
    case 4: hipLaunchKernelGGL((fill_kernel<4, double>), dim3(grid), dim3(256), 0, st, (double*)out, ld, rows, cols, rng, site, a, b); break;
    case 5: hipLaunchKernelGGL((fill_kernel<5, uint8_t>), dim3(grid), dim3(256), 0, st, (uint8_t*)out, ld, rows, cols, rng, site, a, b); break;
    default: DPC_CHECK_ARG(false, "fill: unknown kind");
  }
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused gradient post-processing + Adam + clamp over one flat parameter buffer.
//   g_eff = gscale * g + seg_add[segment(i)] + noise_scale * (noise_inj ? noise_inj[i] : N(0,1)_i)
//   g_eff += wd * p;  m = m + (1-b1)(g_eff - m);  v = b2 v + (1-b2) g_eff^2
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps);  p = clamp(p, +-clamp) if clamp > 0
// Algorithmic HBM traffic: 16 B read + 12 B written per parameter (+4 with an injected noise tensor).
// ---------------------------------------------------------------------------------------------
#define DPC_ADAM_MAX_SLICES 16
#define DPC_ADAM_MAX_MATS 8
// A parameter matrix [rows][ld] at flat offset `off` whose columns c with (c mod period) >= col0 belong to cond
// slices: the flat blocks leave those elements to the tile blocks.  ld / period divisions by multiply-shift
// (x < 2^31: q = (x * m) >> s is exact for m = ceil(2^s / d), s = 31 + ceil(log2 d)).
struct AdamMat { int64_t off, end; uint32_t ld, ld_m, ld_s, period, per_m, per_s, col0; };
// One cond slice: columns [col0, col0 + ncol) of the rows of a parameter matrix.  gT (nullable): transposed gradient
// addend [ncol][ld_gT], g += gT[c][row] (the sorted row sums of dpc_hot_rows_sum), the sum written back to g;
// pT (nullable): transposed copy of the UPDATED parameters [ncol][ld_pT] (W[:, cond]^T, the row-gather table of the
// forward).  tile0: first tile block of the slice.
struct AdamSlice { int64_t off; int rows, ld, col0, ncol; const float* gT; int64_t ld_gT; float* pT; int64_t ld_pT; int tile0; };

struct AdamArgs {
  float* p; float* g; float* m; float* v; int64_t n;
  float lr, b1, b2, eps, wd, clamp, gscale;
  const int32_t* t_dev; int t_host;
  const int64_t* seg_off; const float* seg_add; int nseg;
  float noise_scale; const float* noise_inj; const uint64_t* rng; uint32_t site;
  int flat_blocks, tile_blocks, nmat, nslice;
  AdamMat mat[DPC_ADAM_MAX_MATS];
  AdamSlice sl[DPC_ADAM_MAX_SLICES];
};

__device__ __forceinline__ uint32_t adam_div(uint32_t x, uint32_t m, uint32_t s) {
  return (uint32_t)(((uint64_t)x * m) >> s);
}

// does flat element i belong to a cond slice (tile blocks update it)?
__device__ __forceinline__ bool adam_in_slice(const AdamArgs& a, int64_t i) {
  for (int k = 0; k < a.nmat; ++k) {
    const AdamMat& t = a.mat[k];
    if (i >= t.off && i < t.end) {
      const uint32_t x = (uint32_t)(i - t.off);
      const uint32_t col = x - adam_div(x, t.ld_m, t.ld_s) * t.ld;
      const uint32_t cc = col - adam_div(col, t.per_m, t.per_s) * t.period;
      return cc >= t.col0;
    }
  }
  return false;
}

// (FMA contraction off: the flat blocks and the tile blocks of one launch must give an element the same update to
// the bit, whichever of them owns it - left to the compiler, v came out one ulp apart between the two call sites)
// 0: no element of the quad i0 .. i0+3 belongs to a cond slice; 1: all four do; 2: look at each
__device__ __forceinline__ int adam_quad_class(const AdamArgs& a, int64_t i0) {
  for (int k = 0; k < a.nmat; ++k) {
    const AdamMat& t = a.mat[k];
    if (i0 + 3 >= t.off && i0 < t.end) {
      if (i0 < t.off || i0 + 3 >= t.end) return 2;
      const uint32_t x = (uint32_t)(i0 - t.off);
      const uint32_t col = x - adam_div(x, t.ld_m, t.ld_s) * t.ld;
      if (col + 3 >= t.ld) return 2;
      const uint32_t cc = col - adam_div(col, t.per_m, t.per_s) * t.period;
      if (cc + 3 >= t.period) return 2;
      if (cc >= t.col0) return 1;
      return cc + 3 < t.col0 ? 0 : 2;
    }
  }
  return 0;
}

__device__ __forceinline__ float adam_one(const AdamArgs& a, int64_t i, float g, float p, float& m, float& v,
                                          float z, float step_size, float bc2s) {
#pragma clang fp contract(off)
  g *= a.gscale;
  if (a.nseg > 0) {
    int s = 0;
    while (s + 1 < a.nseg && i >= a.seg_off[s + 1]) ++s;
    g += a.seg_add[s];
  }
  g += a.noise_scale * z;
  g += a.wd * p;
  m = m + (1.f - a.b1) * (g - m);
  v = a.b2 * v + (1.f - a.b2) * g * g;
  float denom = sqrtf(v) / bc2s + a.eps;
  p -= step_size * (m / denom);
  if (a.clamp > 0.f) p = fminf(fmaxf(p, -a.clamp), a.clamp);
  return p;
}

// (Measured in round 3 at config 5's first Linear - 76 M parameters, half of them in cond slices - tools/adam_bench.py:
// flat pass alone 393 us = 5.4 TB/s; slices without addend and copy 531 us, + addend 601, + copy 556, all 648 us =
// 3.75 TB/s.  64 x 64 tiles (256-byte runs on both sides, four rows per thread in flight) changed nothing: 648 us.
// The strided tile walk itself is what costs; the step never READS W1's cond columns in the [d0][K0] layout
// (forward and weight gradient go through W1cT / dW1cT), so the way out is to make the transposed copy the master of
// those parameters and their moments and update it flat - DESIGN.md section 10.)
// A 32 x 32 tile of a cond slice: the transposed gradient addend comes in through LDS, the elements take the same
// update as the flat blocks give (same flat index for the segment lookup and the noise draw), and the updated
// parameters leave through LDS into the transposed copy.  256 threads, four elements each per phase, every global
// access coalesced along its own buffer's rows.
typedef float adam_f4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte access at dword alignment

__device__ __forceinline__ void adam_slice_tile(const AdamArgs& a, int tb, float step_size, float bc2s, float (*t)[33]) {
  int k = 0;
  while (k + 1 < a.nslice && tb >= a.sl[k + 1].tile0) ++k;
  const AdamSlice& sl = a.sl[k];
  const int tcols = (sl.ncol + 31) >> 5;
  const int lt = tb - sl.tile0;
  const int r0 = (lt / tcols) * 32, c0 = (lt % tcols) * 32;
  const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;       // hi: 0..7
  const bool noisy = a.noise_scale != 0.f;
  float gt[4] = {0.f, 0.f, 0.f, 0.f};
  if (sl.gT) {          // (the addend tile's loads are in flight together with the update's own loads below)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = min(c0 + hi + 8 * q, sl.ncol - 1), r2 = min(r0 + lo, sl.rows - 1);
      gt[q] = sl.gT[(int64_t)cc * sl.ld_gT + r2];
    }
  }
  // the update: thread (rr = tid / 8, c4 = 4 * (tid % 8)) owns columns c4 .. c4+3 of row rr - one 16-byte access
  // per buffer when the four lie inside the slice (all but the slice's last partial group), element-wise otherwise
  const int rr = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  const bool rok = r0 + rr < sl.rows;
  const int nc = rok ? max(0, min(4, sl.ncol - (c0 + c4))) : 0;          // live columns of this thread
  const int64_t i0 = sl.off + (int64_t)min(r0 + rr, sl.rows - 1) * sl.ld + sl.col0 + c0 + c4;
  float gg[4] = {0.f, 0.f, 0.f, 0.f}, pp[4] = {0.f, 0.f, 0.f, 0.f}, mm[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
  float zz[4] = {0.f, 0.f, 0.f, 0.f};
  if (nc == 4) {
    const bool nt = a.n > (int64_t)(16 << 20);         // as in the flat blocks: stream m and v past the caches when they cannot hold them
    const adam_f4u g4 = *reinterpret_cast<const adam_f4u*>(a.g + i0), p4 = *reinterpret_cast<const adam_f4u*>(a.p + i0);
    const adam_f4u m4 = nt ? __builtin_nontemporal_load(reinterpret_cast<const adam_f4u*>(a.m + i0)) : *reinterpret_cast<const adam_f4u*>(a.m + i0);
    const adam_f4u v4 = nt ? __builtin_nontemporal_load(reinterpret_cast<const adam_f4u*>(a.v + i0)) : *reinterpret_cast<const adam_f4u*>(a.v + i0);
#pragma unroll
    for (int e = 0; e < 4; ++e) { gg[e] = g4[e]; pp[e] = p4[e]; mm[e] = m4[e]; vv[e] = v4[e]; }
    if (noisy && a.noise_inj) {
      const adam_f4u z4 = *reinterpret_cast<const adam_f4u*>(a.noise_inj + i0);
#pragma unroll
      for (int e = 0; e < 4; ++e) zz[e] = z4[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < nc) {
        gg[e] = a.g[i0 + e]; pp[e] = a.p[i0 + e]; mm[e] = a.m[i0 + e]; vv[e] = a.v[i0 + e];
        if (noisy && a.noise_inj) zz[e] = a.noise_inj[i0 + e];
      }
    }
  }
  if (noisy && !a.noise_inj && nc > 0) {
    // the four flat indices i0 .. i0+3 span at most two Philox blocks
    float za[4], zb[4];
    dpc_normal4(a.rng, a.site, (uint64_t)(i0 >> 2), za);
    dpc_normal4(a.rng, a.site, (uint64_t)((i0 + 3) >> 2), zb);
    const int sh = (int)(i0 & 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = sh + e;              // 0..6: element j of the pair of blocks
      const float lo4 = (j & 3) == 0 ? za[0] : ((j & 3) == 1 ? za[1] : ((j & 3) == 2 ? za[2] : za[3]));
      const float hi4 = (j & 3) == 0 ? zb[0] : ((j & 3) == 1 ? zb[1] : ((j & 3) == 2 ? zb[2] : zb[3]));
      zz[e] = j < 4 ? lo4 : hi4;
    }
  }
  if (sl.gT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) t[hi + 8 * q][lo] = gt[q];       // (entries beyond the slice's edge are never read)
    __syncthreads();
  }
  float pn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < nc) {
      if (sl.gT) gg[e] += t[c4 + e][rr];
      pn[e] = adam_one(a, i0 + e, gg[e], pp[e], mm[e], vv[e], zz[e], step_size, bc2s);
    }
  }
  if (nc == 4) {
    if (sl.gT) *reinterpret_cast<adam_f4u*>(a.g + i0) = adam_f4u{gg[0], gg[1], gg[2], gg[3]};
    *reinterpret_cast<adam_f4u*>(a.p + i0) = adam_f4u{pn[0], pn[1], pn[2], pn[3]};
    if (a.n > (int64_t)(16 << 20)) {
      __builtin_nontemporal_store(adam_f4u{mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<adam_f4u*>(a.m + i0));
      __builtin_nontemporal_store(adam_f4u{vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<adam_f4u*>(a.v + i0));
    } else {
      *reinterpret_cast<adam_f4u*>(a.m + i0) = adam_f4u{mm[0], mm[1], mm[2], mm[3]};
      *reinterpret_cast<adam_f4u*>(a.v + i0) = adam_f4u{vv[0], vv[1], vv[2], vv[3]};
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < nc) {
        if (sl.gT) a.g[i0 + e] = gg[e];
        a.p[i0 + e] = pn[e]; a.m[i0 + e] = mm[e]; a.v[i0 + e] = vv[e];
      }
    }
  }
  if (sl.pT) {
    __syncthreads();                 // the addend tile has been consumed
#pragma unroll
    for (int e = 0; e < 4; ++e) t[c4 + e][rr] = pn[e];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cc = hi + 8 * q, r2 = lo;
      if (c0 + cc < sl.ncol && r0 + r2 < sl.rows) sl.pT[(int64_t)(c0 + cc) * sl.ld_pT + r0 + r2] = t[cc][r2];
    }
  }
}

// 16-byte load / store of a buffer only this launch uses (g, m, v): non-temporal when the four buffers together are
// larger than the 256 MiB Infinity Cache (streamed once per step: 4.73 -> 5.04 TB/s at 76 M parameters), plain when
// they fit (at 1.3 M parameters the next step finds m and v in the cache; non-temporal there cost 3 us per launch).
__device__ __forceinline__ float4 adam_ld4(const float* p, bool nt) {
  typedef float nt4 __attribute__((ext_vector_type(4)));
  if (nt) {
    const nt4 x = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p));
    return make_float4(x.x, x.y, x.z, x.w);
  }
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void adam_st4(float* p, float4 x, bool nt) {
  typedef float nt4 __attribute__((ext_vector_type(4)));
  if (nt) __builtin_nontemporal_store(nt4{x.x, x.y, x.z, x.w}, reinterpret_cast<nt4*>(p));
  else *reinterpret_cast<float4*>(p) = x;
}

// VEC: all four buffers (and the injected noise) are 16-byte aligned; each thread owns 4 consecutive
// parameters = one Philox block of noise.
template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  __shared__ float bc[2];
  __shared__ float tile[32][33];
  if (threadIdx.x == 0) {
    // beta^t for the integer step count by repeated squaring (pow() in fp64 is some hundred dependent instructions
    // that every block of the launch would sit behind at its first barrier)
    int t = (a.t_dev ? a.t_dev[0] : 0) + a.t_host;
    double p1 = 1.0, p2 = 1.0, x1 = (double)a.b1, x2 = (double)a.b2;
    for (int e = t; e > 0; e >>= 1) {
      if (e & 1) { p1 *= x1; p2 *= x2; }
      x1 *= x1; x2 *= x2;
    }
    bc[0] = (float)(a.lr / (1.0 - p1));
    bc[1] = (float)sqrt(1.0 - p2);
  }
  __syncthreads();
  const float step_size = bc[0], bc2s = bc[1];
  // the tile blocks come FIRST in the grid: they have the longer life (three barriers, two passes through LDS) and
  // the flat blocks fill in around them (tile blocks LAST, measured: 14.8 -> 17.1 us at the T1M discriminator's size)
  if ((int)blockIdx.x < a.tile_blocks) {
    adam_slice_tile(a, (int)blockIdx.x, step_size, bc2s, tile);
    return;
  }
  // (measured and dropped: requesting every thread's first quad BEFORE that barrier, beside thread 0's step-size
  // computation - 14.8 -> 19.3 us at the T1M discriminator's size: the flat blocks then also fetch the quads that
  // belong to the tile blocks)
  const int fb = (int)blockIdx.x - a.tile_blocks;
  const bool noisy = a.noise_scale != 0.f;
  const int64_t nq = (a.n + 3) >> 2;
  const bool nt = a.n > (int64_t)(16 << 20);       // 4 buffers x 4 B x 16 Mi = the 256 MiB Infinity Cache
  for (int64_t q = (int64_t)fb * blockDim.x + threadIdx.x; q < nq; q += (int64_t)a.flat_blocks * blockDim.x) {
    const int64_t i0 = q << 2;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (noisy && !a.noise_inj) dpc_normal4(a.rng, a.site, (uint64_t)q, z);
    // elements of cond slices are the tile blocks' (block-uniform nmat == 0: nothing to test)
    bool skip[4] = {false, false, false, false};
    if (a.nmat > 0) {
      // one pair of divisions classifies a quad that lies inside one row and one period of its matrix (all but the
      // quads on a boundary): all ours, or all a tile block's
      int cls = adam_quad_class(a, i0);
      if (cls == 1) continue;
      if (cls == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) skip[j] = adam_in_slice(a, i0 + j);
        if (skip[0] && skip[1] && skip[2] && skip[3]) continue;
      }
    }
    if (VEC && i0 + 4 <= a.n) {
      // (g, m and v are this launch's alone - read once, m and v written once, next touched a step later: streamed
      // past the caches with non-temporal accesses; p stays cacheable, the next products read it)
      float4 p = *reinterpret_cast<const float4*>(a.p + i0);
      float4 g = adam_ld4(a.g + i0, nt), m = adam_ld4(a.m + i0, nt), v = adam_ld4(a.v + i0, nt);
      if (noisy && a.noise_inj) {
        float4 zi = *reinterpret_cast<const float4*>(a.noise_inj + i0);
        z[0] = zi.x; z[1] = zi.y; z[2] = zi.z; z[3] = zi.w;
      }
      p.x = adam_one(a, i0 + 0, g.x, p.x, m.x, v.x, z[0], step_size, bc2s);
      p.y = adam_one(a, i0 + 1, g.y, p.y, m.y, v.y, z[1], step_size, bc2s);
      p.z = adam_one(a, i0 + 2, g.z, p.z, m.z, v.z, z[2], step_size, bc2s);
      p.w = adam_one(a, i0 + 3, g.w, p.w, m.w, v.w, z[3], step_size, bc2s);
      if (a.nmat > 0 && (skip[0] || skip[1] || skip[2] || skip[3])) {
        // a quad that touches a cond slice: element-wise stores of the elements that are ours (the tile blocks
        // write the others in this same launch)
        const float pv[4] = {p.x, p.y, p.z, p.w}, mv[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!skip[j]) { a.p[i0 + j] = pv[j]; a.m[i0 + j] = mv[j]; a.v[i0 + j] = vv[j]; }
        }
      } else {
        *reinterpret_cast<float4*>(a.p + i0) = p;
        adam_st4(a.m + i0, m, nt);
        adam_st4(a.v + i0, v, nt);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t i = i0 + j;
        if (i < a.n && !skip[j]) {
          float m = a.m[i], v = a.v[i];
          float zz = (noisy && a.noise_inj) ? a.noise_inj[i] : z[j];
          a.p[i] = adam_one(a, i, a.g[i], a.p[i], m, v, zz, step_size, bc2s);
          a.m[i] = m; a.v[i] = v;
        }
      }
    }
  }
}

static void adam_magic(uint32_t d, uint32_t& m, uint32_t& s) {
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  s = 31 + l;
  m = (uint32_t)(((1ull << s) + d - 1) / d);
}

static int adam_launch(AdamArgs& a, void* stream) {
  const uintptr_t al = (uintptr_t)a.p | (uintptr_t)a.g | (uintptr_t)a.m | (uintptr_t)a.v | (uintptr_t)a.noise_inj;
  a.flat_blocks = dpc_grid((a.n + 3) / 4, 256);
  int tiles = 0;
  for (int k = 0; k < a.nslice; ++k) {
    a.sl[k].tile0 = tiles;
    tiles += (int)(dpc_cdiv(a.sl[k].rows, 32) * dpc_cdiv(a.sl[k].ncol, 32));
  }
  a.tile_blocks = tiles;
  const dim3 grid((unsigned)(a.flat_blocks + tiles));
  if ((al & 15) == 0) hipLaunchKernelGGL(adam_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(adam_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

static int adam_fill(AdamArgs& a, float* p, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                     float wd, float clamp, float gscale, const int32_t* t_dev, int t_host, const int64_t* seg_off,
                     const float* seg_add, int nseg, float noise_scale, const float* noise_inj, const uint64_t* rng,
                     uint32_t site) {
  DPC_CHECK_ARG(p && g && m && v, "adam: null pointer");
  DPC_CHECK_ARG(n >= 0, "adam: negative size");
  DPC_CHECK_ARG(t_host >= 1 || (t_host == 0 && t_dev), "adam: step count offset must be >= 1 (0 with a device counter that has been bumped already)");
  DPC_CHECK_ARG(nseg == 0 || (seg_off && seg_add), "adam: segment table missing");
  DPC_CHECK_ARG(noise_scale == 0.f || noise_inj || rng, "adam: noise requested without a source");
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
  a.clamp = clamp; a.gscale = gscale; a.t_dev = t_dev; a.t_host = t_host; a.seg_off = seg_off;
  a.seg_add = seg_add; a.nseg = nseg; a.noise_scale = noise_scale; a.noise_inj = noise_inj; a.rng = rng;
  a.site = site; a.nmat = 0; a.nslice = 0;
  return DPC_OK;
}

extern "C" int dpc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                             float b2, float eps, float wd, float clamp, float gscale, const int32_t* t_dev,
                             int t_host, const int64_t* seg_off, const float* seg_add, int nseg,
                             float noise_scale, const float* noise_inj, const uint64_t* rng, uint32_t site,
                             void* stream) {
  AdamArgs a;
  int rc = adam_fill(a, p, const_cast<float*>(g), m, v, n, lr, b1, b2, eps, wd, clamp, gscale, t_dev, t_host, seg_off, seg_add,
                     nseg, noise_scale, noise_inj, rng, site);
  if (rc != DPC_OK) return rc;
  if (n == 0) return DPC_OK;
  return adam_launch(a, stream);
}

// The same step over a parameter buffer that holds matrices with COND SLICES (columns that multiply a two-hot cond
// vector; the generator's Linears over cat([.., cond]), the discriminator's first Linear over pac sub-rows
// [data | cond]: dp_cgan.py:66-68,83,606,622-623).  For slice k = columns [s_col0[k], s_col0[k] + s_ncol[k]) of the
// matrix of s_rows[k] rows of length s_ld[k] at flat offset s_off[k]: the gradient takes the transposed addend
// s_gT[k] [ncol][s_ld_gT[k]] (nullable; the sum is written back to g) and the updated parameters are also written
// transposed to s_pT[k] [ncol][s_ld_pT[k]] (nullable) - what two dpc_transpose_slices launches around dpc_adam_step
// did, in the one launch.  The nmat matrices that own the slices: offset m_off, rows m_rows, row length m_ld; their
// cond columns are those with (column mod m_period) >= m_col0.  All arrays are HOST arrays.
extern "C" int dpc_adam_step_slices(float* p, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                    float eps, float wd, float clamp, float gscale, const int32_t* t_dev, int t_host,
                                    const int64_t* seg_off, const float* seg_add, int nseg, float noise_scale,
                                    const float* noise_inj, const uint64_t* rng, uint32_t site, int nmat,
                                    const int64_t* m_off, const int* m_rows, const int* m_ld, const int* m_period,
                                    const int* m_col0, int nslice, const int64_t* s_off, const int* s_rows,
                                    const int* s_ld, const int* s_col0, const int* s_ncol, const float* const* s_gT,
                                    const int64_t* s_ld_gT, float* const* s_pT, const int64_t* s_ld_pT, void* stream) {
  AdamArgs a;
  int rc = adam_fill(a, p, g, m, v, n, lr, b1, b2, eps, wd, clamp, gscale, t_dev, t_host, seg_off, seg_add, nseg, noise_scale,
                     noise_inj, rng, site);
  if (rc != DPC_OK) return rc;
  DPC_CHECK_ARG(nmat >= 0 && nmat <= DPC_ADAM_MAX_MATS && nslice >= 0 && nslice <= DPC_ADAM_MAX_SLICES, "adam_slices: at most 8 matrices and 16 slices");
  DPC_CHECK_ARG(nmat == 0 || (m_off && m_rows && m_ld && m_period && m_col0), "adam_slices: matrix table missing");
  DPC_CHECK_ARG(nslice == 0 || (s_off && s_rows && s_ld && s_col0 && s_ncol && s_gT && s_ld_gT && s_pT && s_ld_pT), "adam_slices: slice table missing");
  if (n == 0) return DPC_OK;
  for (int k = 0; k < nmat; ++k) {
    DPC_CHECK_ARG(m_off[k] >= 0 && m_rows[k] > 0 && m_ld[k] > 0 && m_period[k] > 0 && m_period[k] <= m_ld[k] && m_col0[k] >= 0 &&
                  m_col0[k] <= m_period[k] && m_off[k] + (int64_t)m_rows[k] * m_ld[k] <= n && (int64_t)m_rows[k] * m_ld[k] < (1ll << 31),
                  "adam_slices: bad matrix");
    AdamMat& t = a.mat[k];
    t.off = m_off[k]; t.end = m_off[k] + (int64_t)m_rows[k] * m_ld[k]; t.ld = (uint32_t)m_ld[k]; t.period = (uint32_t)m_period[k];
    t.col0 = (uint32_t)m_col0[k];
    adam_magic(t.ld, t.ld_m, t.ld_s);
    adam_magic(t.period, t.per_m, t.per_s);
  }
  for (int k = 0; k < nslice; ++k) {
    DPC_CHECK_ARG(s_off[k] >= 0 && s_rows[k] > 0 && s_ld[k] > 0 && s_col0[k] >= 0 && s_ncol[k] > 0 && s_col0[k] + s_ncol[k] <= s_ld[k] &&
                  s_off[k] + (int64_t)s_rows[k] * s_ld[k] <= n, "adam_slices: bad slice");
    DPC_CHECK_ARG((!s_gT[k] || s_ld_gT[k] >= s_rows[k]) && (!s_pT[k] || s_ld_pT[k] >= s_rows[k]), "adam_slices: transposed buffers need rows of at least the matrix's row count");
    // every element of the slice must be one the flat blocks skip
    bool owned = false;
    for (int q = 0; q < nmat; ++q)
      owned = owned || (s_off[k] == m_off[q] && s_rows[k] == m_rows[q] && s_ld[k] == m_ld[q] && s_col0[k] % m_period[q] >= m_col0[q] &&
                        (s_col0[k] + s_ncol[k] - 1) % m_period[q] >= m_col0[q] && s_col0[k] / m_period[q] == (s_col0[k] + s_ncol[k] - 1) / m_period[q]);
    DPC_CHECK_ARG(owned, "adam_slices: a slice must lie inside the cond columns of one of the matrices");
    AdamSlice& sl = a.sl[k];
    sl.off = s_off[k]; sl.rows = s_rows[k]; sl.ld = s_ld[k]; sl.col0 = s_col0[k]; sl.ncol = s_ncol[k]; sl.gT = s_gT[k];
    sl.ld_gT = s_ld_gT[k]; sl.pT = s_pT[k]; sl.ld_pT = s_ld_pT[k];
  }
  // ... and the converse: the flat blocks skip EVERY cond column of a listed matrix, so the slices must cover them
  // all, exactly once - a column left out would never be updated (p, m and v would silently go stale)
  for (int q = 0; q < nmat; ++q) {
    const int64_t per_row = (int64_t)(m_ld[q] / m_period[q]) * (m_period[q] - m_col0[q]) +
                            (m_ld[q] % m_period[q] > m_col0[q] ? m_ld[q] % m_period[q] - m_col0[q] : 0);
    int64_t covered = 0;
    for (int k = 0; k < nslice; ++k) {
      if (s_off[k] != m_off[q] || s_ld[k] != m_ld[q]) continue;
      covered += s_ncol[k];
      for (int j = 0; j < k; ++j)
        DPC_CHECK_ARG(s_off[j] != s_off[k] || s_col0[j] + s_ncol[j] <= s_col0[k] || s_col0[k] + s_ncol[k] <= s_col0[j],
                      "adam_slices: two slices of one matrix overlap");
    }
    DPC_CHECK_ARG(covered == per_row, "adam_slices: the slices of a matrix must cover all of its cond columns (the flat pass skips them)");
  }
  a.nmat = nmat; a.nslice = nslice;
  return adam_launch(a, stream);
}

// Standalone DP noise-add (+ optional clamp) over a flat buffer: x = clamp(x + scale * noise).
// 8 B/element of algorithmic HBM traffic (12 with an injected noise tensor).
__device__ __forceinline__ float4 noise_clamp4(float4 v, const float z[4], float scale, float clamp) {
  v.x += scale * z[0]; v.y += scale * z[1]; v.z += scale * z[2]; v.w += scale * z[3];
  if (clamp > 0.f) {
    v.x = fminf(fmaxf(v.x, -clamp), clamp); v.y = fminf(fmaxf(v.y, -clamp), clamp);
    v.z = fminf(fmaxf(v.z, -clamp), clamp); v.w = fminf(fmaxf(v.w, -clamp), clamp);
  }
  return v;
}

__global__ __launch_bounds__(256) void noise_add_clamp_kernel(float* x, int64_t n, float scale, const float* inj,
                                                              const uint64_t* rng, uint32_t site, float clamp) {
  const int64_t n4 = n >> 2;
  float4* x4 = reinterpret_cast<float4*>(x);
  const float4* j4 = reinterpret_cast<const float4*>(inj);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // Four 16-byte loads in flight per thread (quads t, t+256, t+512, t+768 of the block's 1024-quad
  // chunk: every wave instruction is one coalesced KiB), issued before any of the Philox / Box-Muller
  // work; one load per thread left the kernel bound by bytes in flight (32 KiB per CU), not by HBM
  // or by the VALU.  Noise and stores then go quad by quad to keep the register count at 8 waves/SIMD.
  for (int64_t base = (int64_t)blockIdx.x * 1024; base < n4; base += (int64_t)gridDim.x * 1024) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t i = base + j * 256 + threadIdx.x;
      if (i < n4) v[j] = x4[i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t i = base + j * 256 + threadIdx.x;
      if (i < n4) {
        float z[4];
        if (inj) {
          float4 q = j4[i];
          z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w;
        } else {
          dpc_normal4(rng, site, (uint64_t)i, z);       // normals 4i .. 4i+3 (same map as dpc_draw_normal)
        }
        x4[i] = noise_clamp4(v[j], z, scale, clamp);
      }
    }
  }
  // tail (n % 4 elements)
  for (int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    float z = inj ? inj[t] : dpc_draw_normal(rng, site, (uint64_t)t);
    float v = x[t] + scale * z;
    if (clamp > 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    x[t] = v;
  }
}

extern "C" int dpc_noise_add_clamp(float* x, int64_t n, float scale, const float* noise_inj, const uint64_t* rng,
                                   uint32_t site, float clamp, void* stream) {
  DPC_CHECK_ARG(x, "noise_add_clamp: null pointer");
  DPC_CHECK_ARG(noise_inj || rng, "noise_add_clamp: no noise source");
  DPC_CHECK_ARG(((uintptr_t)x & 15) == 0 && (!noise_inj || ((uintptr_t)noise_inj & 15) == 0), "noise_add_clamp: buffers must be 16-byte aligned");
  if (n <= 0) return n == 0 ? DPC_OK : DPC_EINVAL;
  hipLaunchKernelGGL(noise_add_clamp_kernel, dim3(dpc_grid((n + 3) / 4, 1024)), dim3(256), 0, (hipStream_t)stream,
                     x, n, scale, noise_inj, rng, site, clamp);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
