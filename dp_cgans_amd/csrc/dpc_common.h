// Shared device/host helpers for the DP-CGAN gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPC_OK 0
#define DPC_EINVAL (-1)   // bad argument (shape, alignment, null pointer)
#define DPC_EHIP (-2)     // HIP runtime reported an error at launch

extern "C" void dpc_set_error(const char* msg);

#define DPC_CHECK_ARG(cond, msg)            \
  do {                                      \
    if (!(cond)) {                          \
      dpc_set_error(msg);                   \
      return DPC_EINVAL;                    \
    }                                       \
  } while (0)

#define DPC_LAUNCH_CHECK()                        \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) {                      \
      dpc_set_error(hipGetErrorString(e__));      \
      return DPC_EHIP;                            \
    }                                             \
  } while (0)

static inline int64_t dpc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for grid-stride elementwise kernels: enough blocks to fill 256 CUs x 8, no more.
static inline int dpc_grid(int64_t work_items, int block) {
  int64_t g = dpc_cdiv(work_items, block);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// RNG state: two uint64 in device memory {seed, step}.  Every kernel that draws reads them from
// the device so that a captured hipGraph replays with fresh numbers; dpc_rng_advance bumps step.
// Stream layout: Philox4x32-10, key = seed, counter = {idx/4 lo, idx/4 hi, step lo, site | step hi<<16};
// draw number idx of (site, step) is lane idx%4 of that block.  dpc_fill_* expose the same
// mapping, so the draws a fused kernel used can be reproduced and handed to the oracle.
// ---------------------------------------------------------------------------------------------
struct DpcPhilox4 {
  uint32_t v[4];
};

// Philox4x32-10 (Salmon et al., SC 2011), the conservative default round count; 7 rounds measured
// no faster on the noise kernels (they are bound by bytes in flight, not by the multiplies).
// 
#define DPC_PHILOX_ROUNDS 10
__device__ __forceinline__ DpcPhilox4 dpc_philox(uint64_t seed, uint64_t step, uint32_t site, uint64_t blk) {
  uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32);
  uint32_t c2 = (uint32_t)step, c3 = (site & 0xffffu) | ((uint32_t)(step >> 32) << 16);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < DPC_PHILOX_ROUNDS; ++r) {
    // full 32x32 -> 64 products (one v_mad_u64_u32 each instead of a mul_hi + mul_lo pair)
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  DpcPhilox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

__device__ __forceinline__ uint32_t dpc_draw_u32(const uint64_t* rng, uint32_t site, uint64_t idx) {
  DpcPhilox4 p = dpc_philox(rng[0], rng[1], site, idx >> 2);
  // selects, not p.v[idx & 3]: a dynamically indexed private array lives in scratch memory
  const uint32_t l = (uint32_t)idx & 3u;
  uint32_t lo = (l & 1u) ? p.v[1] : p.v[0], hi = (l & 1u) ? p.v[3] : p.v[2];
  return (l & 2u) ? hi : lo;
}

// The same draws for a thread that walks consecutive indices: the Philox block is recomputed only
// when idx crosses into the next group of four (a 10-round Philox per element made the activation
// kernel ALU-bound).
struct DpcDrawCache {
  uint64_t seed, step, cur;
  uint32_t site;
  DpcPhilox4 p;
  __device__ __forceinline__ DpcDrawCache(const uint64_t* rng, uint32_t site_) : seed(rng[0]), step(rng[1]), cur(~0ull), site(site_) {}
  __device__ __forceinline__ uint32_t draw(uint64_t idx) {
    const uint64_t blk = idx >> 2;
    if (blk != cur) {
      p = dpc_philox(seed, step, site, blk);
      cur = blk;
    }
    const uint32_t l = (uint32_t)idx & 3u;
    uint32_t lo = (l & 1u) ? p.v[1] : p.v[0], hi = (l & 1u) ? p.v[3] : p.v[2];
    return (l & 2u) ? hi : lo;
  }
};

// (0,1) open interval, 23 bits: k + 0.5 for k < 2^23 is exact in fp32, so neither 0 nor 1 can come
// out (with 24 bits the largest value would round up to 1.0 and -log(-log(u)) would be +inf).
__device__ __forceinline__ float dpc_u32_to_unit(uint32_t x) { return ((x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// Transcendentals of the activation path on the hardware units (v_log_f32 = log2, v_exp_f32 = 2^x, v_rcp_f32; 1 ulp
// each): the library's logf / expf / tanhf / IEEE division are 15-40 VALU instructions apiece and made the
// activation kernels instruction-bound (166 VALU instructions per element at T1M, SQ counters).  Every kernel that
// draws or consumes an Exp(1) number uses THESE functions, so the draws dpc_fill exposes are the fused kernels' own.
__device__ __forceinline__ float dpc_fast_ln(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float dpc_fast_exp(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
__device__ __forceinline__ float dpc_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// Exp(1) = -ln(u), u in (0,1) (Tensor.exponential_ inside F.gumbel_softmax, dp_cgan.py:307)
__device__ __forceinline__ float dpc_exp1(uint32_t x) { return -dpc_fast_ln(dpc_u32_to_unit(x)); }
// tanh(x) = 1 - 2 / (e^{2x} + 1): absolute error ~1e-7 (the outputs are compared on the tensor's scale)
__device__ __forceinline__ float dpc_fast_tanh(float x) {
  const float e = dpc_fast_exp(2.0f * fminf(fmaxf(x, -20.f), 20.f));
  return 1.0f - 2.0f * dpc_fast_rcp(e + 1.0f);
}

// [0,1) with 53 bits from two draws (matches the resolution of numpy's random_sample).
__device__ __forceinline__ double dpc_u64_to_unit(uint32_t hi, uint32_t lo) {
  uint64_t a = hi >> 5, b = lo >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

// Standard normals: Philox block b of (site, step) yields normals 4b .. 4b+3 by Box-Muller on its
// word pairs (v0, v1) and (v2, v3), cosine and sine branch each, on the hardware transcendental
// units (v_log_f32 is log2, v_sin/v_cos take revolutions): sqrt(-2 ln u1) = sqrt(-2 ln2 * log2 u1).
// One Philox call and eight quarter-rate instructions per four normals keeps the DP noise kernels
// HBM-bound instead of VALU-bound.  |z| <= 5.77 (u1 >= 2^-24).
__device__ __forceinline__ void dpc_box_muller(uint32_t a, uint32_t b, float& zc, float& zs) {
  float u1 = dpc_u32_to_unit(a), u2 = dpc_u32_to_unit(b);
  float r = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(u1));
  zc = r * __builtin_amdgcn_cosf(u2);
  zs = r * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ void dpc_normal4(const uint64_t* rng, uint32_t site, uint64_t blk, float z[4]) {
  DpcPhilox4 p = dpc_philox(rng[0], rng[1], site, blk);
  dpc_box_muller(p.v[0], p.v[1], z[0], z[1]);
  dpc_box_muller(p.v[2], p.v[3], z[2], z[3]);
}

// Normal number idx (one element of dpc_normal4's block idx >> 2).
__device__ __forceinline__ float dpc_draw_normal(const uint64_t* rng, uint32_t site, uint64_t idx) {
  DpcPhilox4 p = dpc_philox(rng[0], rng[1], site, idx >> 2);
  const bool hi = (idx & 2) != 0;
  float zc, zs;
  dpc_box_muller(hi ? p.v[2] : p.v[0], hi ? p.v[3] : p.v[1], zc, zs);
  return (idx & 1) ? zs : zc;
}

// ---------------------------------------------------------------------------------------------
// wave64 / block reductions
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T dpc_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// fp32 sum over the 64 lanes with DPP adds only (row_shr 1/2/3, 4, 8, then row_bcast 15 and 31: the gfx9 sequence) -
// seven VALU instructions and no LDS traffic, where the __shfl_xor butterfly above issues six ds_bpermute round
// trips per value (twelve for a double).  The sum comes back in every lane (read from lane 63 into an SGPR).
#define DPC_DPP(x, ctrl, row_mask, bank_mask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (row_mask), (bank_mask), false))
__device__ __forceinline__ float dpc_wave_sum_dpp(float v) {
  float s = v + DPC_DPP(v, 0x111, 0xf, 0xf);  // row_shr:1
  s += DPC_DPP(v, 0x112, 0xf, 0xf);           // row_shr:2
  s += DPC_DPP(v, 0x113, 0xf, 0xf);           // row_shr:3  -> s[i] = v[i-3 .. i] within a row of 16
  s += DPC_DPP(s, 0x114, 0xf, 0xe);           // row_shr:4, banks 1-3
  s += DPC_DPP(s, 0x118, 0xf, 0xc);           // row_shr:8, banks 2-3  -> lane 15 of each row holds the row's sum
  s += DPC_DPP(s, 0x142, 0xa, 0xf);           // row_bcast:15 into rows 1 and 3
  s += DPC_DPP(s, 0x143, 0xc, 0xf);           // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
}

// Sum over a block of up to 1024 threads; result valid in every thread.  smem: >= 16 elements.
template <typename T>
__device__ __forceinline__ T dpc_block_sum(T v, T* smem) {
  v = dpc_wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  T t = (T)0;
  for (int i = 0; i < nw; ++i) t += smem[i];
  return t;
}
