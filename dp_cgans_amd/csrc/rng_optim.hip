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
        else if (KIND == 2) v[j] = (T)(-logf(dpc_u32_to_unit(p.v[j])));
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
struct AdamArgs {
  float* p; const float* g; float* m; float* v; int64_t n;
  float lr, b1, b2, eps, wd, clamp, gscale;
  const int32_t* t_dev; int t_host;
  const int64_t* seg_off; const float* seg_add; int nseg;
  float noise_scale; const float* noise_inj; const uint64_t* rng; uint32_t site;
};

__device__ __forceinline__ float adam_one(const AdamArgs& a, int64_t i, float g, float p, float& m, float& v,
                                          float z, float step_size, float bc2s) {
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

// VEC: all four buffers (and the injected noise) are 16-byte aligned; each thread owns 4 consecutive
// parameters = one Philox block of noise.
template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    int t = (a.t_dev ? a.t_dev[0] : 0) + a.t_host;
    bc[0] = (float)(a.lr / (1.0 - pow((double)a.b1, (double)t)));
    bc[1] = (float)sqrt(1.0 - pow((double)a.b2, (double)t));
  }
  __syncthreads();
  const float step_size = bc[0], bc2s = bc[1];
  const bool noisy = a.noise_scale != 0.f;
  const int64_t nq = (a.n + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q << 2;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (noisy && !a.noise_inj) dpc_normal4(a.rng, a.site, (uint64_t)q, z);
    if (VEC && i0 + 4 <= a.n) {
      float4 p = *reinterpret_cast<const float4*>(a.p + i0), g = *reinterpret_cast<const float4*>(a.g + i0);
      float4 m = *reinterpret_cast<const float4*>(a.m + i0), v = *reinterpret_cast<const float4*>(a.v + i0);
      if (noisy && a.noise_inj) {
        float4 zi = *reinterpret_cast<const float4*>(a.noise_inj + i0);
        z[0] = zi.x; z[1] = zi.y; z[2] = zi.z; z[3] = zi.w;
      }
      p.x = adam_one(a, i0 + 0, g.x, p.x, m.x, v.x, z[0], step_size, bc2s);
      p.y = adam_one(a, i0 + 1, g.y, p.y, m.y, v.y, z[1], step_size, bc2s);
      p.z = adam_one(a, i0 + 2, g.z, p.z, m.z, v.z, z[2], step_size, bc2s);
      p.w = adam_one(a, i0 + 3, g.w, p.w, m.w, v.w, z[3], step_size, bc2s);
      *reinterpret_cast<float4*>(a.p + i0) = p;
      *reinterpret_cast<float4*>(a.m + i0) = m;
      *reinterpret_cast<float4*>(a.v + i0) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t i = i0 + j;
        if (i < a.n) {
          float m = a.m[i], v = a.v[i];
          float zz = (noisy && a.noise_inj) ? a.noise_inj[i] : z[j];
          a.p[i] = adam_one(a, i, a.g[i], a.p[i], m, v, zz, step_size, bc2s);
          a.m[i] = m; a.v[i] = v;
        }
      }
    }
  }
}

extern "C" int dpc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                             float b2, float eps, float wd, float clamp, float gscale, const int32_t* t_dev,
                             int t_host, const int64_t* seg_off, const float* seg_add, int nseg,
                             float noise_scale, const float* noise_inj, const uint64_t* rng, uint32_t site,
                             void* stream) {
  DPC_CHECK_ARG(p && g && m && v, "adam: null pointer");
  DPC_CHECK_ARG(n >= 0, "adam: negative size");
  DPC_CHECK_ARG(t_host >= 1, "adam: step count offset must be >= 1");
  DPC_CHECK_ARG(nseg == 0 || (seg_off && seg_add), "adam: segment table missing");
  DPC_CHECK_ARG(noise_scale == 0.f || noise_inj || rng, "adam: noise requested without a source");
  if (n == 0) return DPC_OK;
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd;
  a.clamp = clamp; a.gscale = gscale; a.t_dev = t_dev; a.t_host = t_host; a.seg_off = seg_off;
  a.seg_add = seg_add; a.nseg = nseg; a.noise_scale = noise_scale; a.noise_inj = noise_inj; a.rng = rng;
  a.site = site;
  const uintptr_t al = (uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)noise_inj;
  if ((al & 15) == 0) hipLaunchKernelGGL(adam_kernel<true>, dim3(dpc_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(adam_kernel<false>, dim3(dpc_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
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
