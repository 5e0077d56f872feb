// _apply_activate (tanh / Gumbel-softmax over the span table) forward + backward, and the
// conditional cross-entropy loss (closed form) forward + gradient in one pass.
// Reference: dp_cgan.py:280-327 (activation), :330-389 (cond loss, closed form per SURVEY F6).
// All HBM-bound: 8 B per generator-output element for the activation, 4 B read (+4 written for
// the gradient) per element plus the cond/mask rows for the loss.
#include "dpc_common.h"

// One term of the conditional loss (closed form) and its gradient factor: BCE-with-logits of the raw
// output x against the cond entry c under mask m.  One function for the three kernels that take the loss.
__device__ __forceinline__ float act_loss_term(float xv, float c, float m, float scale, float& gval) {
  const float ex = dpc_fast_exp(-fabsf(xv));
  const float r = dpc_fast_rcp(1.f + ex);
  const float sig = xv >= 0.f ? r : ex * r;
  gval = scale * m * (sig - c);
  return m * (fmaxf(xv, 0.f) - xv * c + dpc_fast_ln(1.f + ex));
}

struct ActFwdArgs {
  const float* x; int64_t ldx; float* y; int64_t ldy; const int32_t* spans; int nspan; int64_t B; int Dd;
  const float* en; int64_t ld_en; const uint64_t* rng; uint32_t site; float inv_tau;
  int R;                      // rows per tile
  // conditional loss (dfake != NULL || partials != NULL): see dpc_activate_cond_loss_fwd
  const float* cond; int64_t ldc; const int32_t* mask; int64_t ldm; const int32_t* col2d; const int32_t* col2c;
  float scale; float* dfake; int64_t lddf; double* partials;
};

// Narrow rows (Adult, T1M, T10M): two phases around ONE barrier, laid out for the instruction count.  Round 3's SQ
// counters showed the three-phase form (rows staged through LDS on the way in and on the way out) VALU-bound at 166
// instructions per element - library logf / expf / division (now the hardware units, dpc_common.h) and per-element
// index bookkeeping.
//   LDS     ts[R][ldT] logits (ldT = Dd rounded up to 4, + 8 of padding) | spans[3*nspan] | LOSS: col2d[Dd] | col2c[Dd] | mask [R*nd].
//   phase 1 one thread per (row, Philox block that touches the row): a row of Dd elements starting at flat index
//           b*Dd lies in at most nqr = (Dd+3)/4 + 1 blocks; a block on a row boundary is computed by both rows
//           (2 in ~85), which buys a thread whose four elements share one row - no wrap-around bookkeeping.  The
//           four raw outputs come straight from global memory in ONE 16-byte load at 4-byte alignment (consecutive
//           threads, consecutive addresses; a block that hangs over the row reads the neighbouring row's storage)
//           while the Philox block is computed; t = (x - log E) / tau goes to LDS, all four values unpredicated
//           (the overhang lands in the row's padding).
//   phase 2 one thread per (row, span): softmax over the span's LDS values (max, exp, sum in column order, scale
//           by 1/sum); spans of up to 4 columns - all of them through the reference's public wrapper - stay in
//           registers; the result goes straight to y (consecutive threads hold consecutive spans: the wave's stores
//           are contiguous).  tanh spans and the conditional loss read their raw values again (L2-hot; the loss only
//           for the two chosen discrete columns of a row - mask bit set - so its transcendentals run for ~1 lane in
//           64), and the loss gradient is written here as well.
// Rows and items are found from flat indices by a float reciprocal + correction (exact below 2^16; a 32-bit division
// by a runtime value is ~30 instructions).
// Measured at T1M's 8192 x 335 (tools/act_bench.py, launches back to back in a graph): 20.7 -> 14.0 us, with the loss
// 37 -> 23.5 us; then 11.9 / 22.1 us once the injected-draws pointer stopped aliasing x (`er = en ? en : x` made every
// thread load its four raw values twice: 14.0 -> 12.3 us), the four clamped 4-byte loads became one 16-byte load
// (-> 12.05) and the LDS stores lost their predicates (-> 11.9).  A plain copy of the same bytes takes 4.7 us.
// What the loss adds to the 11.8 us (parts switched off one at a time, every index kept valid): the chosen spans'
// terms 3.7 us, the zeros of the other spans' gradient 1.3 us, the fp64 block sum + partials 1.1 us, the mask tile's
// loads 0.6 us, staging the two column tables and the rest of the LOSS template 3.2 us.  Tried for the 3.7 us and
// not kept: a third phase with one wave per row and its lanes across the chosen span's categories (found by a
// ballot over the row's spans) instead of the owner thread walking them alone: 22.1 -> 22.7 us - six rows on four
// waves are two rounds of the same L2 round trip.  With phases switched off one at a time: reading the
// rows into LDS alone 7.6 us, + coalesced stores 10.7 us (the load -> barrier -> store skeleton at ~20 KB in flight
// per CU is latency-bound), Philox + logs 1.1 us, the span pass 5.5 us (12.7 us with the loss: one L2 round trip per
// pass of its item loop).  Tried on top and not kept: raw rows and the tile's cond rows staged in LDS with span-owner
// threads walking the rows (no global load in phase 2: 13.7 / 27-31 us), two items per thread with both items'
// loads issued first (+-0), 16-byte staging loads with padded LDS rows (+-0).
typedef float act_f4u __attribute__((ext_vector_type(4), aligned(4)));      // four floats at any 4-byte alignment

__device__ __forceinline__ uint32_t act_div(uint32_t x, uint32_t d, float inv_d) {
  uint32_t q = (uint32_t)((float)x * inv_d);
  if ((q + 1) * d <= x) ++q;
  if (q * d > x) --q;
  return q;
}

template <bool LOSS>
__global__ __launch_bounds__(256) void activate_fwd_small_kernel(ActFwdArgs a, int nd) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ double red[16];
  const uint32_t Dd = (uint32_t)a.Dd, R = (uint32_t)a.R, nspan = (uint32_t)a.nspan;
  const uint32_t ldT = ((Dd + 3u) & ~3u) + 8u;      // 4 floats of padding in front of a row's logits, 4+ behind (phase 1)
  const uint32_t nqr = ((Dd + 3u) >> 2) + 1u;
  float* ts = tile;
  int32_t* sp = reinterpret_cast<int32_t*>(ts + R * ldT);
  int32_t* c2d = sp + 3 * nspan;
  int32_t* c2c = c2d + Dd;
  int32_t* mk = c2c + Dd;
  for (uint32_t i = threadIdx.x; i < 3 * nspan; i += 256) sp[i] = a.spans[i];
  if (LOSS) {
    for (uint32_t i = threadIdx.x; i < Dd; i += 256) {
      c2d[i] = a.col2d[i];
      c2c[i] = a.col2c[i];
    }
  }
  const float inv_nqr = 1.0f / (float)nqr, inv_ns = 1.0f / (float)nspan;
  double acc = 0.0;
  const int64_t ntiles = (a.B + R - 1) / R;
  for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const int64_t b0 = tix * R;
    const uint32_t rows = (uint32_t)min((int64_t)R, a.B - b0);
    const float* __restrict__ xg = a.x + b0 * a.ldx;
    if (LOSS) {
      const int32_t* __restrict__ mg = a.mask + b0 * a.ldm;
      for (uint32_t r = 0; r < rows; ++r)
        for (uint32_t d = threadIdx.x; d < (uint32_t)nd; d += 256) mk[r * (uint32_t)nd + d] = mg[r * a.ldm + d];
    }
    // ---- phase 1: one thread per (row, Philox block)
    const uint32_t nit = rows * nqr;
    for (uint32_t it = threadIdx.x; it < nit; it += 256) {
      const uint32_t r = act_div(it, nqr, inv_nqr), j = it - r * nqr;
      const uint64_t fr = (uint64_t)(b0 + r) * Dd;                 // flat index of the row's first element
      const int c0 = (int)(j << 2) - (int)((uint32_t)fr & 3u);     // column of the block's first element
      if (c0 >= (int)Dd) continue;                                 // (the row ends one block earlier)
      const float* __restrict__ xr = xg + r * a.ldx;
      float xv[4], ev[4];
      // The block's four raw values in ONE 16-byte load (any 4-byte alignment).  A block that hangs over the row's
      // first or last column reads the neighbouring row's storage (ldx >= Dd: it exists) - except in the matrix's very
      // first and last row, whose overhanging blocks clamp their columns instead.
      if ((c0 >= 0 || b0 + r > 0) && (c0 + 3 < (int)Dd || b0 + r + 1 < a.B)) {
        const act_f4u q = *reinterpret_cast<const act_f4u*>(xr + c0);
        xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
      } else {
#pragma unroll
        for (int l = 0; l < 4; ++l) xv[l] = xr[min(max(c0 + l, 0), (int)Dd - 1)];
      }
      if (a.en) {                                                  // injected Exp(1) draws (tests): a uniform branch
        const float* __restrict__ er = a.en + (b0 + r) * a.ld_en;
#pragma unroll
        for (int l = 0; l < 4; ++l) ev[l] = er[min(max(c0 + l, 0), (int)Dd - 1)];
      } else {
        const DpcPhilox4 ph = dpc_philox(a.rng[0], a.rng[1], a.site, (fr >> 2) + j);
#pragma unroll
        for (int l = 0; l < 4; ++l) ev[l] = dpc_exp1(ph.v[l]);
      }
      // all four values are stored: what hangs over the row (c0 in -3 .. Dd-1) lands in the row's padding
      float* tr = ts + r * ldT + 4u + c0;
#pragma unroll
      for (int l = 0; l < 4; ++l) tr[l] = (xv[l] - dpc_fast_ln(ev[l])) * a.inv_tau;
    }
    __syncthreads();
    // ---- phase 2: one thread per (row, span)
    const uint32_t nitem = rows * nspan;
    for (uint32_t it = threadIdx.x; it < nitem; it += 256) {
      const uint32_t r = act_div(it, nspan, inv_ns), s = it - r * nspan;
      const int off = sp[3 * s], dim = sp[3 * s + 1], kind = sp[3 * s + 2];
      const float* tr = ts + r * ldT + 4u + off;
      const float* __restrict__ xr = xg + r * a.ldx + off;
      float* __restrict__ yr = a.y + (b0 + r) * a.ldy + off;
      float* __restrict__ dr = (LOSS && a.dfake) ? a.dfake + (b0 + r) * a.lddf + off : nullptr;
      if (kind == 0) {
        for (int j = 0; j < dim; ++j) {
          yr[j] = dpc_fast_tanh(xr[j]);
          if (LOSS && dr) dr[j] = 0.f;
        }
        continue;
      }
      if (dim <= 4) {
        float t[4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t[j] = j < dim ? tr[j] : -INFINITY;
          mx = fmaxf(mx, t[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t[j] = dpc_fast_exp(t[j] - mx);            // (a slot beyond dim: 2^-inf = 0)
          sum += t[j];
        }
        const float inv = dpc_fast_rcp(sum);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < dim) yr[j] = t[j] * inv;
      } else {
        float mx = -INFINITY;
        for (int j = 0; j < dim; ++j) mx = fmaxf(mx, tr[j]);
        float sum = 0.f;
        for (int j = 0; j < dim; ++j) sum += dpc_fast_exp(tr[j] - mx);
        const float inv = dpc_fast_rcp(sum);
        for (int j = 0; j < dim; ++j) yr[j] = dpc_fast_exp(tr[j] - mx) * inv;
      }
      if (LOSS) {
        const int d = c2d[off];                 // >= 0: a discrete column (its span is the whole column)
        const float m = d >= 0 ? (float)mk[r * (uint32_t)nd + d] : 0.f;
        if (m != 0.f) {
          const float* __restrict__ cr = a.cond + (b0 + r) * a.ldc + c2c[off];
          for (int j = 0; j < dim; ++j) {
            float gval;
            acc += (double)act_loss_term(xr[j], cr[j], m, a.scale, gval);
            if (dr) dr[j] = gval;
          }
        } else if (dr) {
          for (int j = 0; j < dim; ++j) dr[j] = 0.f;
        }
      }
    }
    __syncthreads();
  }
  if (LOSS && a.partials) {
    const double tot = dpc_block_sum(acc, red);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
  }
}

// rows per LDS tile: as many as fit 32 KB, at most 8 (more blocks beat longer tiles: the kernel is
// latency-bound at a few blocks per CU); 0 = a row does not fit the 128 KB a block may take
static int act_tile_rows(int Dd, int64_t B, int arrays) {
  const int64_t row_bytes = (int64_t)Dd * 4 * arrays;
  int64_t R = (32 << 10) / row_bytes;
  if (R > 8) R = 8;
  if (R < 1) R = (row_bytes <= (128 << 10)) ? 1 : 0;
  if (R > B) R = B;
  return (int)R;
}

static int act_set_lds(const void* fn, size_t bytes) {
  if (bytes <= (48u << 10)) return DPC_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { dpc_set_error(hipGetErrorString(e)); return DPC_EHIP; }
  return DPC_OK;
}

struct ActBwdArgs {
  const float* y; int64_t ldy; const float* dy; int64_t lddy; float* dx; int64_t lddx;
  const int32_t* spans; int nspan; int64_t B; int Dd; float inv_tau; int accumulate; int R;
};

// Backward, same tiling: y and dy tiles in LDS (coalesced loads), one thread per (row, span) turns dy into
// the span's input gradient in place, coalesced (accumulating) stores.
template <bool SPL>      // SPL: the span table sits in LDS behind the tiles
__global__ __launch_bounds__(256) void activate_bwd_tiled_kernel(ActBwdArgs a, int rshift) {
  extern __shared__ __attribute__((aligned(16))) float tile[];       // y [R][Dd], then dy [R][Dd]
  const uint32_t Dd = (uint32_t)a.Dd, R = (uint32_t)a.R, nspan = (uint32_t)a.nspan;   // R = 1 << rshift
  const int64_t ntiles = (a.B + R - 1) / R;
  float* ty = tile;
  float* tg = tile + (size_t)R * Dd;
  int32_t* sp = reinterpret_cast<int32_t*>(tg + (size_t)R * Dd);
  if (SPL)
    for (uint32_t i = threadIdx.x; i < 3u * nspan; i += 256) sp[i] = a.spans[i];
  const int32_t* spans = SPL ? sp : a.spans;
  for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const int64_t b0 = tix * R;
    const uint32_t rows = (uint32_t)min((int64_t)R, a.B - b0);
    const float* __restrict__ yg = a.y + b0 * a.ldy;
    const float* __restrict__ gg = a.dy + b0 * a.lddy;
    for (uint32_t r = 0; r < rows; ++r)
      for (uint32_t col = threadIdx.x; col < Dd; col += 256) {
        ty[r * Dd + col] = yg[r * a.ldy + col];
        tg[r * Dd + col] = gg[r * a.lddy + col];
      }
    __syncthreads();
    const uint32_t r = threadIdx.x >> (8 - rshift), tpr = 256u >> rshift;     // 256 / R threads per row
    if (r < rows) {
      for (uint32_t s = threadIdx.x & (tpr - 1); s < nspan; s += tpr) {
        const int off = spans[3 * s], dim = spans[3 * s + 1], kind = spans[3 * s + 2];
        const float* yr = ty + r * Dd + off;
        float* gr = tg + r * Dd + off;
        if (kind == 0) {
          for (int j = 0; j < dim; ++j) gr[j] = gr[j] * (1.f - yr[j] * yr[j]);
        } else {
          float dot = 0.f;
          for (int j = 0; j < dim; ++j) dot += yr[j] * gr[j];
          for (int j = 0; j < dim; ++j) gr[j] = a.inv_tau * yr[j] * (gr[j] - dot);
        }
      }
    }
    __syncthreads();
    float* dg = a.dx + b0 * a.lddx;
    for (uint32_t r2 = 0; r2 < rows; ++r2)
      for (uint32_t col = threadIdx.x; col < Dd; col += 256) {
        float* dp = dg + r2 * a.lddx + col;
        *dp = a.accumulate ? *dp + tg[r2 * Dd + col] : tg[r2 * Dd + col];
      }
    __syncthreads();
  }
}

// span table rows: (offset, dim, kind) kind 0 = tanh, 1 = gumbel-softmax
// One (row, span) straight from global memory: the path for spans that do not fit an LDS tile.  The span's
// intermediate values live in y (a store -> load round trip per pass), its Philox blocks are recomputed per element
// group (DpcDrawCache).
template <bool LOSS>
__device__ __forceinline__ void act_span_direct(const ActFwdArgs& a, int64_t b, int off, int dim, int kind, double& acc) {
  const float* __restrict__ xr = a.x + b * a.ldx + off;
  float* __restrict__ yr = a.y + b * a.ldy + off;
  float* __restrict__ dr = (LOSS && a.dfake) ? a.dfake + b * a.lddf + off : nullptr;
  if (kind == 0) {
    for (int j = 0; j < dim; ++j) {
      yr[j] = dpc_fast_tanh(xr[j]);
      if (dr) dr[j] = 0.f;
    }
    return;
  }
  const uint64_t zero2[2] = {0, 0};
  DpcDrawCache dc(a.en ? zero2 : a.rng, a.site);
  float mx = -INFINITY;
  for (int j = 0; j < dim; ++j) {
    const float e = a.en ? a.en[b * a.ld_en + off + j] : dpc_exp1(dc.draw((uint64_t)(b * a.Dd + off + j)));
    const float t = (xr[j] - dpc_fast_ln(e)) * a.inv_tau;
    yr[j] = t;
    mx = fmaxf(mx, t);
  }
  float sum = 0.f;
  for (int j = 0; j < dim; ++j) {
    const float e = dpc_fast_exp(yr[j] - mx);
    yr[j] = e;
    sum += e;
  }
  const float inv = dpc_fast_rcp(sum);
  for (int j = 0; j < dim; ++j) yr[j] *= inv;
  if (LOSS) {
    const int d = a.col2d[off];
    const float m = d >= 0 ? (float)a.mask[b * a.ldm + d] : 0.f;
    if (m != 0.f) {
      const float* __restrict__ cr = a.cond + b * a.ldc + a.col2c[off];
      for (int j = 0; j < dim; ++j) {
        float gval;
        acc += (double)act_loss_term(xr[j], cr[j], m, a.scale, gval);
        if (dr) dr[j] = gval;
      }
    } else if (dr) {
      for (int j = 0; j < dim; ++j) dr[j] = 0.f;
    }
  }
}

// Wide rows (TWIDE: 19 000 columns in 9 800 spans - a row does not fit the narrow kernel's LDS budget, and the
// one-row-per-block tile that round 2 used for it ran at 0.12 of the HBM roofline: 80 KB of LDS per block, 38 serial
// span passes per row).  Same two phases as the narrow kernel, per WORK ITEM = (tile of R rows) x (chunk of ACT_CHUNK
// consecutive spans): the chunk's columns [cA, cB) are contiguous, phase 1 walks the Philox blocks that cover them
// in each row (the draw of flat index b*Dd + col does not care how rows are cut), phase 2 holds one span per thread
// and row.  Chunks are the fast index: neighbouring blocks work on neighbouring columns of the same rows.  A chunk
// whose R rows exceed the LDS tile (spans hundreds of columns wide) takes its spans straight from global memory
// (act_span_direct) - always correct, fast for the tables the reference's wrapper produces (2-column spans).
// Phase 2 here: thread s owns span s of the chunk (at most 256 = the block) and walks the tile's rows, so the span's
// table entry, its loss-table entries and the row-independent address parts are computed once per work item.
// TWIDE (8192 x 19 000, tools/act_bench.py 5 TWIDE): 489 -> 388 us = 3.2 TB/s (0.41 of HBM; the 16-byte load did it:
// 460 -> 386), with the loss 1113 -> 818 us (span-owner threads: 968 -> 818).  The SQ counters of the 386 us form: 195 M
// wave-level VALU instructions per launch, VALU-active 207 M quad-cycles over 1024 SIMDs = 338 us - the kernel is
// bound by its arithmetic, and ~60 % of that is the draw: Philox4x32-10's twenty quarter-rate 32 x 32 multiplies per
// four elements and two logarithms per element (ln(-ln u)).  Rows per item 1 / 2 / 4 / 8 / 12 / 16 -> 628 / 470 / 388 /
// 392 / 414 / 480 us; an 8-byte store for two-column spans +-0.
constexpr int ACT_CHUNK = 256;

template <bool LOSS>
__global__ __launch_bounds__(256) void activate_fwd_chunk_kernel(ActFwdArgs a, int nd, uint32_t lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ double red[16];
  __shared__ int32_t sp[3 * ACT_CHUNK];
  const uint32_t Dd = (uint32_t)a.Dd, R = (uint32_t)a.R, nspan = (uint32_t)a.nspan;
  const uint32_t nchunk = (nspan + ACT_CHUNK - 1) / ACT_CHUNK;
  const int64_t ntiles = (a.B + R - 1) / R, nwork = ntiles * nchunk;
  float* ts = tile;
  double acc = 0.0;
  for (int64_t wix = blockIdx.x; wix < nwork; wix += gridDim.x) {
    const int64_t tix = wix / nchunk;
    const uint32_t s0 = (uint32_t)(wix - tix * nchunk) * ACT_CHUNK, ns = min((uint32_t)ACT_CHUNK, nspan - s0);
    for (uint32_t i = threadIdx.x; i < 3 * ns; i += 256) sp[i] = a.spans[3 * s0 + i];
    __syncthreads();
    const uint32_t cA = (uint32_t)sp[0], w = (uint32_t)(sp[3 * (ns - 1)] + sp[3 * (ns - 1) + 1]) - cA;
    const int64_t b0 = tix * R;
    const uint32_t rows = (uint32_t)min((int64_t)R, a.B - b0);
    const uint32_t nitem = rows * ns;
    const float inv_ns = 1.0f / (float)ns;
    const uint32_t ldw = w + 8u;                   // LDS row: 4 floats of padding in front, 4+ behind (see phase 1)
    if ((uint64_t)R * ldw <= lds_floats && w < 32768u) {
      // ---- phase 1: one thread per (row, Philox block that touches the chunk's columns of the row)
      const uint32_t nqr = ((w + 3u) >> 2) + 1u, nit = rows * nqr;
      const float inv_nqr = 1.0f / (float)nqr;
      for (uint32_t it = threadIdx.x; it < nit; it += 256) {
        const uint32_t r = act_div(it, nqr, inv_nqr), j = it - r * nqr;
        const uint64_t fr = (uint64_t)(b0 + r) * Dd + cA;            // flat index of the chunk's first element in the row
        const int c0 = (int)(j << 2) - (int)((uint32_t)fr & 3u);     // chunk-relative column of the block's first element
        if (c0 >= (int)w) continue;
        const float* __restrict__ xr = a.x + (b0 + r) * a.ldx + cA;
        float xv[4], ev[4];
        // (one 16-byte load; only the matrix's first and last element have no neighbour to read over: see the narrow kernel)
        if (((int)cA + c0 >= 0 || b0 + r > 0) && (cA + (uint32_t)(c0 + 3) < Dd || b0 + r + 1 < a.B)) {
          const act_f4u q = *reinterpret_cast<const act_f4u*>(xr + c0);
          xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
        } else {
#pragma unroll
          for (int l = 0; l < 4; ++l) xv[l] = xr[min(max(c0 + l, 0), (int)w - 1)];
        }
        if (a.en) {
          const float* __restrict__ er = a.en + (b0 + r) * a.ld_en + cA;
#pragma unroll
          for (int l = 0; l < 4; ++l) ev[l] = er[min(max(c0 + l, 0), (int)w - 1)];
        } else {
          const DpcPhilox4 ph = dpc_philox(a.rng[0], a.rng[1], a.site, (fr >> 2) + j);
#pragma unroll
          for (int l = 0; l < 4; ++l) ev[l] = dpc_exp1(ph.v[l]);
        }
        // all four values are stored: what hangs over the chunk's columns (c0 in -3 .. w-1) lands in the row's padding
        float* tr = ts + r * ldw + 4u + c0;
#pragma unroll
        for (int l = 0; l < 4; ++l) tr[l] = (xv[l] - dpc_fast_ln(ev[l])) * a.inv_tau;
      }
      __syncthreads();
      // ---- phase 2: thread s holds span s of the chunk (ns <= 256 = the block) and walks the tile's rows: the span's
      // table entry, its column and every row-independent address part are computed once; consecutive threads hold
      // consecutive spans, so each row's stores are contiguous
      if (threadIdx.x < ns) {
        const uint32_t s = threadIdx.x;
        const int off = sp[3 * s], dim = sp[3 * s + 1], kind = sp[3 * s + 2];
        int d = -1, cc = 0;
        if (LOSS) {
          d = a.col2d[off];
          cc = a.col2c[off];
        }
        for (uint32_t r = 0; r < rows; ++r) {
          const float* tr = ts + r * ldw + 4u + ((uint32_t)off - cA);
          const float* __restrict__ xr = a.x + (b0 + r) * a.ldx + off;
          float* __restrict__ yr = a.y + (b0 + r) * a.ldy + off;
          float* __restrict__ dr = (LOSS && a.dfake) ? a.dfake + (b0 + r) * a.lddf + off : nullptr;
          if (kind == 0) {
            for (int j = 0; j < dim; ++j) {
              yr[j] = dpc_fast_tanh(xr[j]);
              if (LOSS && dr) dr[j] = 0.f;
            }
            continue;
          }
          if (dim <= 4) {
            float t[4];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              t[j] = j < dim ? tr[j] : -INFINITY;
              mx = fmaxf(mx, t[j]);
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              t[j] = dpc_fast_exp(t[j] - mx);
              sum += t[j];
            }
            const float inv = dpc_fast_rcp(sum);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < dim) yr[j] = t[j] * inv;
          } else {
            float mx = -INFINITY;
            for (int j = 0; j < dim; ++j) mx = fmaxf(mx, tr[j]);
            float sum = 0.f;
            for (int j = 0; j < dim; ++j) sum += dpc_fast_exp(tr[j] - mx);
            const float inv = dpc_fast_rcp(sum);
            for (int j = 0; j < dim; ++j) yr[j] = dpc_fast_exp(tr[j] - mx) * inv;
          }
          if (LOSS) {
            const float m = d >= 0 ? (float)a.mask[(b0 + r) * a.ldm + d] : 0.f;
            if (m != 0.f) {
              const float* __restrict__ cr = a.cond + (b0 + r) * a.ldc + cc;
              for (int j = 0; j < dim; ++j) {
                float gval;
                acc += (double)act_loss_term(xr[j], cr[j], m, a.scale, gval);
                if (dr) dr[j] = gval;
              }
            } else if (dr) {
              for (int j = 0; j < dim; ++j) dr[j] = 0.f;
            }
          }
        }
      }
    } else {
      for (uint32_t it = threadIdx.x; it < nitem; it += 256) {
        const uint32_t r = act_div(it, ns, inv_ns), s = it - r * ns;
        act_span_direct<LOSS>(a, b0 + r, sp[3 * s], sp[3 * s + 1], sp[3 * s + 2], acc);
      }
    }
    __syncthreads();
  }
  if (LOSS && a.partials) {
    const double tot = dpc_block_sum(acc, red);
    if (threadIdx.x == 0) a.partials[blockIdx.x] = tot;
  }
}

static int activate_fwd_impl(ActFwdArgs a, bool loss, int nd, int n_partials, int* n_used, float tau, hipStream_t st) {
  a.inv_tau = 1.0f / tau;
  // narrow rows: the logits of R rows and every table in LDS (activate_fwd_small_kernel).  R: among 1..16 rows
  // (<= 40 KB of LDS) the smallest count whose phase-1 items - R rows x the nqr Philox blocks of a row - fill rounds
  // of 256 threads best, counting a grid beyond 2048 blocks (8 resident blocks per CU) as a quarter worse
  {
    const int64_t ldT = (a.Dd + 3) / 4 * 4 + 8, nqr = (a.Dd + 3) / 4 + 1;
    auto lds_of = [&](int64_t R) {
      return ((size_t)R * ldT + 3 * (size_t)a.nspan + (loss ? 2 * (size_t)a.Dd + (size_t)R * nd : 0)) * 4;
    };
    int64_t best = 0;
    double best_u = 0.0;
    for (int64_t R = 1; R <= 16 && R <= a.B; ++R) {
      const int64_t items = R * (nqr > a.nspan ? nqr : (int64_t)a.nspan);
      if (lds_of(R) > (40u << 10) || items >= 65536) break;
      double u = (double)(R * nqr) / (256.0 * (double)dpc_cdiv(R * nqr, 256));
      if (dpc_cdiv(a.B, R) > 2048) u *= 0.75;
      if (u > best_u + 0.02) { best_u = u; best = R; }
    }
    if (best > 0) {
      const int64_t R = best;
      a.R = (int)R;
      const int64_t ntiles = dpc_cdiv(a.B, R);
      int64_t grid = ntiles < 8192 ? ntiles : 8192;
      if (loss && grid > n_partials) grid = n_partials;
      if (loss) hipLaunchKernelGGL(activate_fwd_small_kernel<true>, dim3((unsigned)grid), dim3(256), lds_of(R), st, a, nd);
      else hipLaunchKernelGGL(activate_fwd_small_kernel<false>, dim3((unsigned)grid), dim3(256), lds_of(R), st, a, nd);
      if (n_used) *n_used = (int)grid;
      DPC_LAUNCH_CHECK();
      return DPC_OK;
    }
  }
  // wide rows: work items of R rows x ACT_CHUNK spans (activate_fwd_chunk_kernel).  LDS for R rows of the average
  // chunk width and half as much again, at most 48 KB (asking for 48 KB always left 3 blocks per CU); a chunk that
  // still does not fit goes the direct way, in the kernel
  {
    const int64_t nchunk = dpc_cdiv((int64_t)a.nspan, ACT_CHUNK);
    const int64_t guess = (a.Dd / nchunk) * 3 / 2 + 12;      // (+ 8 floats of padding per LDS row)
    int64_t R = 12288 / guess;
    R = R < 1 ? 1 : (R > 8 ? 8 : R);                   // (sweep: see the kernel's comment)
    if (R > a.B) R = a.B;
    const uint32_t lds_floats = (uint32_t)(R * guess < 12288 ? R * guess : 12288);
    a.R = (int)R;
    const int64_t nwork = dpc_cdiv(a.B, R) * nchunk;
    int64_t grid = nwork < 65536 ? nwork : 65536;
    if (loss && grid > n_partials) grid = n_partials;
    if (loss) hipLaunchKernelGGL(activate_fwd_chunk_kernel<true>, dim3((unsigned)grid), dim3(256), lds_floats * 4, st, a, nd, lds_floats);
    else hipLaunchKernelGGL(activate_fwd_chunk_kernel<false>, dim3((unsigned)grid), dim3(256), lds_floats * 4, st, a, nd, lds_floats);
    if (n_used) *n_used = (int)grid;
  }
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

extern "C" int dpc_apply_activate_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const int32_t* spans,
                                      int nspan, int64_t B, int Dd, const float* exp_noise, int64_t ld_noise,
                                      const uint64_t* rng, uint32_t site, float tau, void* stream) {
  DPC_CHECK_ARG(x && y && spans, "apply_activate_fwd: null pointer");
  DPC_CHECK_ARG(exp_noise || rng, "apply_activate_fwd: no noise source");
  DPC_CHECK_ARG(B >= 0 && nspan >= 0 && ldx >= Dd && ldy >= Dd && tau > 0.f, "apply_activate_fwd: bad shape");
  if (B == 0 || nspan == 0) return DPC_OK;
  ActFwdArgs a = {};
  a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.spans = spans; a.nspan = nspan; a.B = B; a.Dd = Dd; a.en = exp_noise;
  a.ld_en = ld_noise; a.rng = rng; a.site = site;
  return activate_fwd_impl(a, false, 0, 0, nullptr, tau, (hipStream_t)stream);
}

// _apply_activate and _cond_loss_pair (closed form, see dpc_cond_loss) of the generator update in ONE pass
// over the raw generator output: y = activation(x), dfake = d cond-loss / d x (written, zero outside the
// discrete columns), partials[0 : *n_used] = block sums of the un-scaled loss.
extern "C" int dpc_activate_cond_loss_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const int32_t* spans,
                                          int nspan, int64_t B, int Dd, const float* exp_noise, int64_t ld_noise,
                                          const uint64_t* rng, uint32_t site, float tau, const float* cond,
                                          int64_t ldc, const int32_t* mask, int64_t ldm, const int32_t* col2d,
                                          const int32_t* col2c, int nd, int64_t B_total, float* dfake, int64_t lddf,
                                          double* partials, int n_partials, int* n_used, void* stream) {
  DPC_CHECK_ARG(x && y && spans && cond && mask && col2d && col2c && partials && n_used, "activate_cond_loss_fwd: null pointer");
  DPC_CHECK_ARG(exp_noise || rng, "activate_cond_loss_fwd: no noise source");
  DPC_CHECK_ARG(B > 0 && nspan > 0 && nd > 0 && ldx >= Dd && ldy >= Dd && tau > 0.f && n_partials >= 1,
                "activate_cond_loss_fwd: bad shape");
  DPC_CHECK_ARG(!dfake || lddf >= Dd, "activate_cond_loss_fwd: lddf < Dd");
  if (B_total <= 0) B_total = B;
  ActFwdArgs a = {};
  a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.spans = spans; a.nspan = nspan; a.B = B; a.Dd = Dd; a.en = exp_noise;
  a.ld_en = ld_noise; a.rng = rng; a.site = site;
  a.cond = cond; a.ldc = ldc; a.mask = mask; a.ldm = ldm; a.col2d = col2d; a.col2c = col2c;
  a.scale = (float)(2.0 / ((double)B_total * (double)nd)); a.dfake = dfake; a.lddf = lddf; a.partials = partials;
  return activate_fwd_impl(a, true, nd, n_partials, n_used, tau, (hipStream_t)stream);
}

// dx = dL/dx given y (saved activations) and dy.  tanh: dy (1 - y^2); softmax((x+g)/tau):
// (1/tau) y (dy - sum_j y_j dy_j).  accumulate != 0 adds into dx.
template <typename IT>
__global__ __launch_bounds__(256) void activate_bwd_wide_kernel(const float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ dy, int64_t lddy,
                                                           float* __restrict__ dx, int64_t lddx,
                                                           const int32_t* __restrict__ spans, int nspan, int64_t B,
                                                           float inv_tau, int accumulate) {
  const IT total = (IT)B * (IT)nspan, stride = (IT)gridDim.x * blockDim.x;
  for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const IT bq = idx / (IT)nspan;
    const int64_t b = (int64_t)bq;
    int s = (int)(idx - bq * (IT)nspan);
    int off = spans[3 * s], dim = spans[3 * s + 1], kind = spans[3 * s + 2];
    const float* yr = y + b * ldy + off;
    const float* gr = dy + b * lddy + off;
    float* dr = dx + b * lddx + off;
    if (kind == 0) {
      for (int j = 0; j < dim; ++j) {
        float v = gr[j] * (1.f - yr[j] * yr[j]);
        dr[j] = accumulate ? dr[j] + v : v;
      }
    } else if (dim <= 16) {
      float yv[16], gv[16];
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < dim) {
          yv[j] = yr[j];
          gv[j] = gr[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < dim) dot += yv[j] * gv[j];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < dim) {
          float v = inv_tau * yv[j] * (gv[j] - dot);
          dr[j] = accumulate ? dr[j] + v : v;
        }
      }
    } else {
      float dot = 0.f;
      for (int j = 0; j < dim; ++j) dot += yr[j] * gr[j];
      for (int j = 0; j < dim; ++j) {
        float v = inv_tau * yr[j] * (gr[j] - dot);
        dr[j] = accumulate ? dr[j] + v : v;
      }
    }
  }
}

extern "C" int dpc_apply_activate_bwd(const float* y, int64_t ldy, const float* dy, int64_t lddy, float* dx,
                                      int64_t lddx, const int32_t* spans, int nspan, int64_t B, int Dd, float tau,
                                      int accumulate, void* stream) {
  DPC_CHECK_ARG(y && dy && dx && spans, "apply_activate_bwd: null pointer");
  DPC_CHECK_ARG(B >= 0 && nspan >= 0 && tau > 0.f && Dd > 0 && ldy >= Dd && lddy >= Dd && lddx >= Dd, "apply_activate_bwd: bad shape");
  if (B == 0 || nspan == 0) return DPC_OK;
  hipStream_t st = (hipStream_t)stream;
  int R = act_tile_rows(Dd, B, 2);
  if (R > 0) {
    // narrow rows: 4-row tiles and the span table in LDS (8 blocks per CU in flight)
    const bool spl = (size_t)(2 * 4 * Dd + 3 * nspan) * 4 <= (20u << 10);
    if (spl && R > 4) R = 4;
    int rshift = 0;
    while ((2 << rshift) <= R) ++rshift;
    R = 1 << rshift;                       // a power of two: 256 / R threads per row
    ActBwdArgs a;
    a.y = y; a.ldy = ldy; a.dy = dy; a.lddy = lddy; a.dx = dx; a.lddx = lddx; a.spans = spans; a.nspan = nspan; a.B = B;
    a.Dd = Dd; a.inv_tau = 1.0f / tau; a.accumulate = accumulate; a.R = R;
    const int64_t ntiles = dpc_cdiv(B, R);
    const size_t lds = ((size_t)2 * R * Dd + (spl ? 3 * (size_t)nspan : 0)) * sizeof(float);
    const unsigned grid = (unsigned)(ntiles < 8192 ? ntiles : 8192);
    if (spl) {
      hipLaunchKernelGGL(activate_bwd_tiled_kernel<true>, dim3(grid), dim3(256), lds, st, a, rshift);
    } else {
      int rc = act_set_lds((const void*)activate_bwd_tiled_kernel<false>, lds);
      if (rc != DPC_OK) return rc;
      hipLaunchKernelGGL(activate_bwd_tiled_kernel<false>, dim3(grid), dim3(256), lds, st, a, rshift);
    }
  } else if (B * nspan < (1LL << 31)) {
    hipLaunchKernelGGL(activate_bwd_wide_kernel<uint32_t>, dim3(dpc_grid(B * nspan, 256)), dim3(256), 0, st, y,
                       ldy, dy, lddy, dx, lddx, spans, nspan, B, 1.0f / tau, accumulate);
  } else {
    hipLaunchKernelGGL(activate_bwd_wide_kernel<int64_t>, dim3(dpc_grid(B * nspan, 256)), dim3(256), 0, st, y,
                       ldy, dy, lddy, dx, lddx, spans, nspan, B, 1.0f / tau, accumulate);
  }
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// ---------------------------------------------------------------------------------------------
// Conditional loss.  col2d[j] = discrete-column index of generator-output column j (-1 if j is not
// inside a discrete column), col2c[j] = matching column of the cond vector.
//   loss = 2/(B*nd) * sum_b sum_j m[b,col2d[j]] * (max(x,0) - x c + log1p(exp(-|x|)))
//   dfake[b,j] = 2/(B*nd) * m * (sigmoid(x) - c)     (0 where col2d[j] < 0)
// Block partial sums go to `partials[gridDim.x]`; cond_loss_finish adds them in index order
// (bit-reproducible) and scales.
// ---------------------------------------------------------------------------------------------
template <typename IT>
__global__ __launch_bounds__(256) void cond_loss_kernel(const float* __restrict__ fake, int64_t ldf,
                                                        const float* __restrict__ cond, int64_t ldc,
                                                        const int32_t* __restrict__ mask, int64_t ldm,
                                                        const int32_t* __restrict__ col2d,
                                                        const int32_t* __restrict__ col2c, int64_t B, int Dd,
                                                        float scale, float* __restrict__ dfake, int64_t lddf,
                                                        int accumulate, double* __restrict__ partials) {
  __shared__ double red[16];
  double acc = 0.0;
  const IT total = (IT)B * (IT)Dd, stride = (IT)gridDim.x * blockDim.x;
  for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const IT bq = idx / (IT)Dd;
    const int64_t b = (int64_t)bq;
    int j = (int)(idx - bq * (IT)Dd);
    int d = col2d[j];
    float gval = 0.f;
    if (d >= 0) {
      float m = (float)mask[b * ldm + d];
      if (m != 0.f) {
        float x = fake[b * ldf + j];
        float c = cond[b * ldc + col2c[j]];
        acc += (double)act_loss_term(x, c, m, scale, gval);
      }
    }
    if (dfake) {
      float* dp = dfake + b * lddf + j;
      *dp = accumulate ? *dp + gval : gval;
    }
  }
  double tot = dpc_block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void sum_partials_kernel(const double* __restrict__ partials, int n, double scale, float* out) {
  __shared__ double red[16];
  // fixed-order: each thread sums a strided slice, then the block tree; same grid -> same bits.
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
  double tot = dpc_block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = (float)(tot * scale);
}

extern "C" int dpc_cond_loss(const float* fake, int64_t ldf, const float* cond, int64_t ldc, const int32_t* mask,
                             int64_t ldm, const int32_t* col2d, const int32_t* col2c, int64_t B, int Dd, int nd,
                             float* loss_out, float* dfake, int64_t lddf, int accumulate, double* partials,
                             int n_partials, int64_t B_total, int* n_used, void* stream) {
  DPC_CHECK_ARG(fake && cond && mask && col2d && col2c && partials, "cond_loss: null pointer");
  DPC_CHECK_ARG(loss_out || n_used, "cond_loss: neither a loss output nor a partial count requested");
  DPC_CHECK_ARG(B > 0 && Dd > 0 && nd > 0, "cond_loss: bad shape");
  DPC_CHECK_ARG(n_partials >= 1, "cond_loss: workspace too small");
  int grid = dpc_grid(B * Dd, 256);
  if (grid > n_partials) grid = n_partials;
  if (B_total <= 0) B_total = B;      // data parallel: B rows of a global batch of B_total
  float scale = (float)(2.0 / ((double)B_total * (double)nd));
  hipStream_t st = (hipStream_t)stream;
  if (B * Dd < (1LL << 31))
    hipLaunchKernelGGL(cond_loss_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, fake, ldf, cond, ldc, mask, ldm, col2d,
                       col2c, B, Dd, scale, dfake, lddf, accumulate, partials);
  else
    hipLaunchKernelGGL(cond_loss_kernel<int64_t>, dim3(grid), dim3(256), 0, st, fake, ldf, cond, ldc, mask, ldm, col2d,
                       col2c, B, Dd, scale, dfake, lddf, accumulate, partials);
  if (n_used) *n_used = grid;          // the caller may sum partials[0 : grid] * 2/(B_total*nd) later (dpc_sum_f64)
  if (loss_out)
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partials, grid, 2.0 / ((double)B_total * (double)nd), loss_out);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
