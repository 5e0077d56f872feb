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
__global__ void fill_kernel(T* out, int64_t ld, int64_t rows, int64_t cols, const uint64_t* rng, uint32_t site,
                            float a, float b) {
  int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / cols, c = i - r * cols;
    T v;
    if (KIND == 0) v = (T)dpc_draw_u32(rng, site, (uint64_t)i);
    else if (KIND == 1) v = (T)(a + b * dpc_draw_normal(rng, site, (uint64_t)i));
    else if (KIND == 2) v = (T)(-logf(dpc_u32_to_unit(dpc_draw_u32(rng, site, (uint64_t)i))));
    else if (KIND == 3) v = (T)dpc_u32_to_unit(dpc_draw_u32(rng, site, (uint64_t)i));
    else if (KIND == 4) v = (T)dpc_u64_to_unit(dpc_draw_u32(rng, site, 2 * (uint64_t)i), dpc_draw_u32(rng, site, 2 * (uint64_t)i + 1));
    else v = (T)(dpc_draw_u32(rng, site, (uint64_t)i) >> 31);
    out[r * ld + c] = v;
  }
}

extern "C" int dpc_fill(void* out, int kind, int64_t ld, int64_t rows, int64_t cols, const uint64_t* rng,
                        uint32_t site, float a, float b, void* stream) {
  DPC_CHECK_ARG(out && rng, "fill: null pointer");
  DPC_CHECK_ARG(rows >= 0 && cols >= 0 && ld >= cols, "fill: bad shape");
  if (rows * cols == 0) return DPC_OK;
  hipStream_t st = (hipStream_t)stream;
  int grid = dpc_grid(rows * cols, 256);
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
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = a.gscale * a.g[i];
    if (a.nseg > 0) {
      int s = 0;
      while (s + 1 < a.nseg && i >= a.seg_off[s + 1]) ++s;
      g += a.seg_add[s];
    }
    if (noisy) g += a.noise_scale * (a.noise_inj ? a.noise_inj[i] : dpc_draw_normal(a.rng, a.site, (uint64_t)i));
    float p = a.p[i];
    g += a.wd * p;
    float m = a.m[i];
    m = m + (1.f - a.b1) * (g - m);
    float v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
    float denom = sqrtf(v) / bc2s + a.eps;
    p -= step_size * (m / denom);
    if (a.clamp > 0.f) p = fminf(fmaxf(p, -a.clamp), a.clamp);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
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
  hipLaunchKernelGGL(adam_kernel, dim3(dpc_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// Standalone DP noise-add (+ optional clamp) over a flat buffer: x = clamp(x + scale * noise).
// 8 B/element of algorithmic HBM traffic (12 with an injected noise tensor).
__global__ __launch_bounds__(256) void noise_add_clamp_kernel(float* x, int64_t n, float scale, const float* inj,
                                                              const uint64_t* rng, uint32_t site, float clamp) {
  int64_t n4 = n >> 2;
  float4* x4 = reinterpret_cast<float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = x4[i];
    float z0, z1, z2, z3;
    if (inj) {
      float4 q = reinterpret_cast<const float4*>(inj)[i];
      z0 = q.x; z1 = q.y; z2 = q.z; z3 = q.w;
    } else {
      // normals 4i..4i+3 = Box-Muller pairs of Philox blocks 2i and 2i+1 (same map as dpc_draw_normal)
      DpcPhilox4 pa = dpc_philox(rng[0], rng[1], site, 2 * (uint64_t)i);
      DpcPhilox4 pb = dpc_philox(rng[0], rng[1], site, 2 * (uint64_t)i + 1);
      z0 = sqrtf(-2.f * logf(dpc_u32_to_unit(pa.v[0]))) * cospif(2.f * dpc_u32_to_unit(pa.v[1]));
      z1 = sqrtf(-2.f * logf(dpc_u32_to_unit(pa.v[2]))) * cospif(2.f * dpc_u32_to_unit(pa.v[3]));
      z2 = sqrtf(-2.f * logf(dpc_u32_to_unit(pb.v[0]))) * cospif(2.f * dpc_u32_to_unit(pb.v[1]));
      z3 = sqrtf(-2.f * logf(dpc_u32_to_unit(pb.v[2]))) * cospif(2.f * dpc_u32_to_unit(pb.v[3]));
    }
    v.x += scale * z0; v.y += scale * z1; v.z += scale * z2; v.w += scale * z3;
    if (clamp > 0.f) {
      v.x = fminf(fmaxf(v.x, -clamp), clamp); v.y = fminf(fmaxf(v.y, -clamp), clamp);
      v.z = fminf(fmaxf(v.z, -clamp), clamp); v.w = fminf(fmaxf(v.w, -clamp), clamp);
    }
    x4[i] = v;
  }
  // tail
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float z = inj ? inj[i] : dpc_draw_normal(rng, site, (uint64_t)i);
    float v = x[i] + scale * z;
    if (clamp > 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    x[i] = v;
  }
}

extern "C" int dpc_noise_add_clamp(float* x, int64_t n, float scale, const float* noise_inj, const uint64_t* rng,
                                   uint32_t site, float clamp, void* stream) {
  DPC_CHECK_ARG(x, "noise_add_clamp: null pointer");
  DPC_CHECK_ARG(noise_inj || rng, "noise_add_clamp: no noise source");
  DPC_CHECK_ARG(((uintptr_t)x & 15) == 0 && (!noise_inj || ((uintptr_t)noise_inj & 15) == 0), "noise_add_clamp: buffers must be 16-byte aligned");
  if (n <= 0) return n == 0 ? DPC_OK : DPC_EINVAL;
  hipLaunchKernelGGL(noise_add_clamp_kernel, dim3(dpc_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     x, n, scale, noise_inj, rng, site, clamp);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
