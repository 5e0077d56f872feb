// _apply_activate (tanh / Gumbel-softmax over the span table) forward + backward, and the
// conditional cross-entropy loss (closed form) forward + gradient in one pass.
// Reference: dp_cgan.py:280-327 (activation), :330-389 (cond loss, closed form per SURVEY F6).
// All HBM-bound: 8 B per generator-output element for the activation, 4 B read (+4 written for
// the gradient) per element plus the cond/mask rows for the loss.
#include "dpc_common.h"

// span table rows: (offset, dim, kind) kind 0 = tanh, 1 = gumbel-softmax
// IT: index type of the flattened (row, span) loop; uint32_t whenever it fits (a 64-bit division per
// thread is ~100 instructions, which showed in a kernel that does ~15 elements per row and span)
template <typename IT>
__global__ __launch_bounds__(256) void activate_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                           int64_t ldy, const int32_t* __restrict__ spans, int nspan,
                                                           int64_t B, int Dd, const float* __restrict__ en, int64_t ld_en,
                                                           const uint64_t* rng, uint32_t site, float inv_tau) {
  const IT total = (IT)B * (IT)nspan, stride = (IT)gridDim.x * blockDim.x;
  for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const IT bq = idx / (IT)nspan;
    const int64_t b = (int64_t)bq;
    int s = (int)(idx - bq * (IT)nspan);
    int off = spans[3 * s], dim = spans[3 * s + 1], kind = spans[3 * s + 2];
    const float* xr = x + b * ldx + off;
    float* yr = y + b * ldy + off;
    if (kind == 0) {
      for (int j = 0; j < dim; ++j) yr[j] = tanhf(xr[j]);
    } else if (dim <= 16) {
      // the span lives in registers (same operations in the same order as the general path below,
      // which keeps its intermediate values in y and pays a store -> load round trip per pass)
      float t[16];
      float mx = -INFINITY;
      const uint64_t zero2[2] = {0, 0};
      DpcDrawCache dc(en ? zero2 : rng, site);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < dim) {
          float e = en ? en[b * ld_en + off + j] : -logf(dpc_u32_to_unit(dc.draw((uint64_t)(b * Dd + off + j))));
          t[j] = (xr[j] - logf(e)) * inv_tau;
          mx = fmaxf(mx, t[j]);
        }
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < dim) {
          t[j] = expf(t[j] - mx);
          sum += t[j];
        }
      }
      float inv = 1.f / sum;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < dim) yr[j] = t[j] * inv;
    } else {
      float mx = -INFINITY;
      for (int j = 0; j < dim; ++j) {
        float e = en ? en[b * ld_en + off + j]
                     : -logf(dpc_u32_to_unit(dpc_draw_u32(rng, site, (uint64_t)(b * Dd + off + j))));
        float t = (xr[j] - logf(e)) * inv_tau;
        yr[j] = t;
        mx = fmaxf(mx, t);
      }
      float sum = 0.f;
      for (int j = 0; j < dim; ++j) {
        float e = expf(yr[j] - mx);
        yr[j] = e;
        sum += e;
      }
      float inv = 1.f / sum;
      for (int j = 0; j < dim; ++j) yr[j] *= inv;
    }
  }
}

extern "C" int dpc_apply_activate_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const int32_t* spans,
                                      int nspan, int64_t B, int Dd, const float* exp_noise, int64_t ld_noise,
                                      const uint64_t* rng, uint32_t site, float tau, void* stream) {
  DPC_CHECK_ARG(x && y && spans, "apply_activate_fwd: null pointer");
  DPC_CHECK_ARG(exp_noise || rng, "apply_activate_fwd: no noise source");
  DPC_CHECK_ARG(B >= 0 && nspan >= 0 && ldx >= Dd && ldy >= Dd && tau > 0.f, "apply_activate_fwd: bad shape");
  if (B == 0 || nspan == 0) return DPC_OK;
  if (B * nspan < (1LL << 31))
    hipLaunchKernelGGL(activate_fwd_kernel<uint32_t>, dim3(dpc_grid(B * nspan, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, y, ldy, spans, nspan, B, Dd, exp_noise, ld_noise, rng, site, 1.0f / tau);
  else
    hipLaunchKernelGGL(activate_fwd_kernel<int64_t>, dim3(dpc_grid(B * nspan, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, y, ldy, spans, nspan, B, Dd, exp_noise, ld_noise, rng, site, 1.0f / tau);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// dx = dL/dx given y (saved activations) and dy.  tanh: dy (1 - y^2); softmax((x+g)/tau):
// (1/tau) y (dy - sum_j y_j dy_j).  accumulate != 0 adds into dx.
template <typename IT>
__global__ __launch_bounds__(256) void activate_bwd_kernel(const float* __restrict__ y, int64_t ldy,
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
                                      int64_t lddx, const int32_t* spans, int nspan, int64_t B, float tau,
                                      int accumulate, void* stream) {
  DPC_CHECK_ARG(y && dy && dx && spans, "apply_activate_bwd: null pointer");
  DPC_CHECK_ARG(B >= 0 && nspan >= 0 && tau > 0.f, "apply_activate_bwd: bad shape");
  if (B == 0 || nspan == 0) return DPC_OK;
  if (B * nspan < (1LL << 31))
    hipLaunchKernelGGL(activate_bwd_kernel<uint32_t>, dim3(dpc_grid(B * nspan, 256)), dim3(256), 0, (hipStream_t)stream, y,
                       ldy, dy, lddy, dx, lddx, spans, nspan, B, 1.0f / tau, accumulate);
  else
    hipLaunchKernelGGL(activate_bwd_kernel<int64_t>, dim3(dpc_grid(B * nspan, 256)), dim3(256), 0, (hipStream_t)stream, y,
                       ldy, dy, lddy, dx, lddx, spans, nspan, B, 1.0f / tau, accumulate);
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
        float e = expf(-fabsf(x));
        acc += (double)(m * (fmaxf(x, 0.f) - x * c + log1pf(e)));
        float sig = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        gval = scale * m * (sig - c);
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
