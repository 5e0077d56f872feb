// DataTransformer.fit for continuous columns: the variational Bayesian Gaussian
// mixture of /root/reference/src/dp_cgans/synthesizers/data_transformer.py:36-57 (scikit-learn's
// BayesianGaussianMixture: one feature, 'full' covariance, dirichlet_process weights, random initial
// responsibilities) for F independent fits (columns x initialisations) side by side.
//
// One iteration = two launches.
//   gmm_estep_kernel  grid (R, F): streams the column once (8 B per row), evaluates the K weighted
//     log-probabilities A_k - .5 ((x - mu_k) pc_k)^2, the responsibilities, and leaves per block the
//     sufficient statistics {sum r, sum r d, sum r d^2}_k (d = x - mu_k of the CURRENT parameters, so
//     the variance update does not cancel) and sum r log r.  HBM-bound by construction: 8 B per row
//     against ~6 fp64 transcendentals per row.
//   gmm_mstep_kernel  grid F: adds the R partials in a fixed order (deterministic), then one thread
//     restates sklearn's _m_step / _compute_lower_bound in fp64 and decides convergence
//     (|lower bound - previous| < tol).  A finished fit sets done = 1; both kernels return at once
//     for it, so the host can queue iterations blindly and poll `info` every few dozen.
#include "dpc_common.h"

#define GMM_MAXK 16
#define GMM_ROWS 8         // params rows: mu, pc, A, cov, a, b, mean_precision, dof
#define GMM_INFO 4         // info: lower bound, n_iter, converged, done

// psi(x), x > 0: recurrence up to x >= 10, then the asymptotic series (error < 1e-16 there).
__device__ double gmm_digamma(double x) {
  double r = 0.0;
  while (x < 10.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 +
                  f * (1.0 / 120.0 +
                       f * (-1.0 / 252.0 +
                            f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

template <int KT>
__global__ __launch_bounds__(256) void gmm_estep_kernel(const double* __restrict__ x, int64_t col_stride, int64_t N,
                                                        const int32_t* __restrict__ fit_col,
                                                        const double* __restrict__ params, int K,
                                                        const double* __restrict__ prior,
                                                        const double* __restrict__ info, int init,
                                                        const double* __restrict__ resp_inj, const uint64_t* rng,
                                                        uint32_t site, double* __restrict__ part) {
  const int f = blockIdx.y, R = gridDim.x;
  if (info[(int64_t)f * GMM_INFO + 3] != 0.0) return;
  const double* xc = x + (int64_t)fit_col[f] * col_stride;
  const double* P = params + (int64_t)f * GMM_ROWS * K;
  double mu[KT], pc[KT], A[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const bool live = k < K;
    mu[k] = init ? prior[2 * f] : (live ? P[k] : 0.0);
    pc[k] = (live && !init) ? P[K + k] : 0.0;
    A[k] = (live && !init) ? P[2 * K + k] : (live ? 0.0 : -INFINITY);
  }
  double s0[KT], s1[KT], s2[KT], ent = 0.0;
#pragma unroll
  for (int k = 0; k < KT; ++k) s0[k] = s1[k] = s2[k] = 0.0;

  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)R * blockDim.x) {
    const double xv = xc[i];
    double d[KT], r[KT];
    if (init) {
      // init_params='random' (sklearn _base.py:121-125): uniforms normalised over the components
      double tot = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        double u = 0.0;
        if (k < K) {
          if (resp_inj) {
            u = resp_inj[((int64_t)f * N + i) * K + k];
          } else {
            uint64_t idx = 2 * ((uint64_t)i * K + k);
            u = dpc_u64_to_unit(dpc_draw_u32(rng, site + f, idx), dpc_draw_u32(rng, site + f, idx + 1));
          }
        }
        r[k] = u;
        tot += u;
        d[k] = xv - mu[k];
      }
      if (!resp_inj) {
#pragma unroll
        for (int k = 0; k < KT; ++k) r[k] = r[k] / tot;
      }
    } else {
      double w[KT], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        d[k] = xv - mu[k];
        double y = d[k] * pc[k];
        w[k] = A[k] - 0.5 * (y * y);
        mx = fmax(mx, w[k]);
      }
      double se = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) se += exp(w[k] - mx);
      const double lse = mx + log(se);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        double lr = w[k] - lse;
        r[k] = exp(lr);
        ent += (r[k] > 0.0) ? r[k] * lr : 0.0;
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      s0[k] += r[k];
      double rd = r[k] * d[k];
      s1[k] += rd;
      s2[k] += rd * d[k];
    }
  }

  __shared__ double red[16];
  double* out = part + ((int64_t)f * R + blockIdx.x) * (3 * K + 1);
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < K) {                                            // K is uniform: no divergence around the barriers
      double a = dpc_block_sum(s0[k], red);
      double b = dpc_block_sum(s1[k], red);
      double c = dpc_block_sum(s2[k], red);
      if (threadIdx.x == 0) {
        out[k] = a;
        out[K + k] = b;
        out[2 * K + k] = c;
      }
    }
  }
  double e = dpc_block_sum(ent, red);
  if (threadIdx.x == 0) out[3 * K] = e;
}

struct GmmHyper {
  double wcp, mpp, dofp, reg_covar, tol;
  int max_iter;
};

__global__ __launch_bounds__(64) void gmm_mstep_kernel(const double* __restrict__ part, int R, int K,
                                                       const double* __restrict__ prior, double* __restrict__ params,
                                                       double* __restrict__ info, int init, GmmHyper h) {
  const int f = blockIdx.x;
  double* I = info + (int64_t)f * GMM_INFO;
  if (I[3] != 0.0) return;
  __shared__ double S[3 * GMM_MAXK + 1];
  const int nstat = 3 * K + 1;
  if ((int)threadIdx.x < nstat) {
    const double* p = part + (int64_t)f * R * nstat + threadIdx.x;
    double t = 0.0;
    for (int r = 0; r < R; ++r) t += p[(int64_t)r * nstat];
    S[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;

  double* P = params + (int64_t)f * GMM_ROWS * K;
  const double mean_prior = prior[2 * f], cov_prior = prior[2 * f + 1];
  double nk[GMM_MAXK], tail[GMM_MAXK];
  for (int k = 0; k < K; ++k) nk[k] = S[k] + 10.0 * 2.220446049250313e-16;
  {                                                         // cumsum(nk[::-1])[-2::-1], 0  (:546)
    double t = 0.0;
    tail[K - 1] = 0.0;
    for (int k = K - 1; k >= 1; --k) {
      t += nk[k];
      tail[k - 1] = t;
    }
  }
  double cum = 0.0;                                          // running sum of psi(b) - psi(a + b)  (:755)
  double log_wishart = 0.0, log_norm_weight = 0.0, sum_log_mp = 0.0;
  for (int k = 0; k < K; ++k) {
    const double shift = init ? mean_prior : P[k];
    const double m1 = S[K + k] / nk[k], m2 = S[2 * K + k] / nk[k];
    const double xk = shift + m1;
    const double sk = fmax(m2 - m1 * m1, 0.0) + h.reg_covar;
    const double a = 1.0 + nk[k], b = h.wcp + tail[k];
    const double mp = h.mpp + nk[k];
    const double mean = (h.mpp * mean_prior + nk[k] * xk) / mp;
    const double dof = h.dofp + nk[k];
    const double dx = xk - mean_prior;
    const double cov = (cov_prior + nk[k] * sk + nk[k] * h.mpp / mp * (dx * dx)) / dof;
    const double pc = 1.0 / sqrt(cov);
    const double dsum = gmm_digamma(a + b);
    const double logw = gmm_digamma(a) - dsum + cum;
    cum += gmm_digamma(b) - dsum;
    const double Ak = logw + log(pc) - 0.5 * log(2.0 * 3.141592653589793) - 0.5 * log(dof) +
                      0.5 * (log(2.0) + gmm_digamma(0.5 * dof) - 1.0 / mp);
    P[k] = mean;
    P[K + k] = pc;
    P[2 * K + k] = Ak;
    P[3 * K + k] = cov;
    P[4 * K + k] = a;
    P[5 * K + k] = b;
    P[6 * K + k] = mp;
    P[7 * K + k] = dof;
    const double ldpc = log(pc) - 0.5 * log(dof);
    log_wishart += -(dof * ldpc + dof * 0.5 * log(2.0) + lgamma(0.5 * dof));
    log_norm_weight += -(lgamma(a) + lgamma(b) - lgamma(a + b));
    sum_log_mp += log(mp);
  }
  if (init) {
    I[0] = -INFINITY;
    I[1] = 0.0;
    I[2] = 0.0;
    return;
  }
  const double lb = -S[3 * K] - log_wishart - log_norm_weight - 0.5 * sum_log_mp;
  const double change = lb - I[0];
  const double n_iter = I[1] + 1.0;
  I[0] = lb;
  I[1] = n_iter;
  if (fabs(change) < h.tol) {
    I[2] = 1.0;
    I[3] = 1.0;
  } else if (n_iter >= (double)h.max_iter) {
    I[3] = 1.0;
  }
}

static int gmm_check(int64_t N, int F, int K, int R) {
  DPC_CHECK_ARG(N >= 2 && F >= 1 && F <= 65535, "gmm: need N >= 2 rows and 1 <= F <= 65535 fits");
  DPC_CHECK_ARG(K >= 1 && K <= GMM_MAXK, "gmm: 1 <= K <= 16 components");
  DPC_CHECK_ARG(R >= 1 && R <= 4096, "gmm: 1 <= R <= 4096 row blocks");
  return DPC_OK;
}

extern "C" int dpc_gmm_estep(const double* x, int64_t col_stride, int64_t N, const int32_t* fit_col, int F,
                             const double* params, int K, const double* prior, const double* info, int init,
                             const double* resp_inj, const uint64_t* rng, uint32_t site, double* part, int R,
                             void* stream) {
  DPC_CHECK_ARG(x && fit_col && params && prior && info && part, "gmm_estep: null pointer");
  if (gmm_check(N, F, K, R) != DPC_OK) return DPC_EINVAL;
  DPC_CHECK_ARG(!init || resp_inj || rng, "gmm_estep: no random source for the initial responsibilities");
  dim3 grid((unsigned)R, (unsigned)F), block(256);
#define GMM_LAUNCH(KT)                                                                                              \
  hipLaunchKernelGGL(gmm_estep_kernel<KT>, grid, block, 0, (hipStream_t)stream, x, col_stride, N, fit_col, params, \
                     K, prior, info, init, resp_inj, rng, site, part)
  if (K <= 5) GMM_LAUNCH(5);
  else if (K <= 8) GMM_LAUNCH(8);
  else GMM_LAUNCH(16);
#undef GMM_LAUNCH
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

extern "C" int dpc_gmm_mstep(const double* part, int R, int F, int K, const double* prior, double* params,
                             double* info, int init, double weight_concentration_prior, double mean_precision_prior,
                             double dof_prior, double reg_covar, double tol, int max_iter, void* stream) {
  DPC_CHECK_ARG(part && prior && params && info, "gmm_mstep: null pointer");
  if (gmm_check(2, F, K, R) != DPC_OK) return DPC_EINVAL;
  DPC_CHECK_ARG(weight_concentration_prior > 0 && mean_precision_prior > 0 && dof_prior > 0 && reg_covar >= 0 &&
                    max_iter >= 1,
                "gmm_mstep: bad hyper-parameter");
  GmmHyper h = {weight_concentration_prior, mean_precision_prior, dof_prior, reg_covar, tol, max_iter};
  hipLaunchKernelGGL(gmm_mstep_kernel, dim3((unsigned)F), dim3(64), 0, (hipStream_t)stream, part, R, K, prior, params,
                     info, init, h);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}
