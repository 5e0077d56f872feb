/* C-ABI of libdpcgan_hip.so: the gfx950 kernels under the DP-CGAN training step.
 *
 * The reference (sunchang0124/dp_cgans) is pure Python and has no FFI; the boundary it offers is
 * the Python class API.  These entry points are what a maintainer would bind with
 * ctypes from the reference's own modules to replace the stock-PyTorch / NumPy code cited beside
 * each one (paths relative to /root/reference/src/dp_cgans/synthesizers/).  INTEGRATION.md shows
 * the binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says host; the caller owns all buffers;
 *    nothing here allocates, frees or synchronises;
 *  - `stream` is a hipStream_t; launches are asynchronous on it;
 *  - return 0 on success, DPC_EINVAL (-1) for a rejected argument, DPC_EHIP (-2) when the HIP
 *    runtime refused the launch; dpc_last_error() (host, thread-local) holds the message;
 *  - matrices are row-major fp32 with an explicit leading dimension (elements, not bytes);
 *  - randomness: `rng` points at two device uint64 {seed, step}.  Draw number idx of call site
 *    `site` is Philox4x32-10(key = seed, counter = {idx/4, step, site}) lane idx%4, so a captured
 *    graph replays with fresh numbers once dpc_state_advance has bumped `step`.  Every kernel
 *    that draws also accepts an injected tensor instead (non-null pointer wins); that is how the
 *    parity tests feed the reference's recorded draws.  dpc_fill exposes the same idx -> draw map.
 */
#ifndef DPCGAN_HIP_H
#define DPCGAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DPC_OK 0
#define DPC_EINVAL (-1)
#define DPC_EHIP (-2)

int dpc_version(void);
const char* dpc_last_error(void);

/* ---- random state / fills -------------------------------------------------------------- */
/* End-of-step bookkeeping in one launch: rng_a[1] += 1, rng_b[1] += 1, *t_a += add_a, *t_b += add_b
 * (any pointer may be NULL).  No reference counterpart: the reference uses the global torch / numpy
 * generators (dp_cgan.py:581, data_sampler.py:211,269) and torch.optim's host step counters. */
int dpc_state_advance(uint64_t* rng_a, uint64_t* rng_b, int32_t* t_a, int add_a, int32_t* t_b, int add_b,
                      void* stream);
/* The same bookkeeping as an argument of another launch (dpc_colsum_multi_adv): one node less per step. */
typedef struct dpc_step_advance {
  uint64_t* rng_a; uint64_t* rng_b;
  int32_t* t_a; int32_t* t_b;
  int add_a, add_b;
} dpc_step_advance;

/* out[r*ld + c] = draw(r*cols + c).  kind: 0 u32, 1 normal(a, b) f32 (torch.normal,
 * dp_cgan.py:581,673), 2 Exp(1) f32 (Tensor.exponential_ inside F.gumbel_softmax, :307),
 * 3 uniform(0,1) f32 (torch.rand, :47), 4 uniform[0,1) f64 (np.random.rand, data_sampler.py:211),
 * 5 keep bit u8 (Dropout(0.5), :40). */
int dpc_fill(void* out, int kind, int64_t ld, int64_t rows, int64_t cols, const uint64_t* rng,
             uint32_t site, float a, float b, void* stream);

/* ---- fp32 GEMM on the matrix cores (replaces every Linear forward/backward: dp_cgan.py:40,43,
 * 75,95 and their autograd) --------------------------------------------------------------- */
/* C = epi(alpha * A.B + bias) + beta*C.  A(m,k) = A[m*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs];
 * one stride of each must be 1.  epi: 0 none; 1 bias+LeakyReLU(leak)+Dropout(0.5): writes
 * slope = (pre>0 ? 1 : leak)*keep*2 to aux_out and C = pre*slope (keep from `keep` u8 [M,N] or
 * Philox idx m*N+n); 2 multiply by aux_in[m,n]; 3 ReLU.  splitk 0 = automatic (fp32 atomics; beta in
 * {0,1}), 1 = never (bit-reproducible).  kscale (nullable, needs an m-contiguous A, i.e. the
 * weight-gradient product dW = dY^T X): A(m,k) is multiplied by kscale[k] while it is staged, which
 * fuses the DP-SGD clip factor of row k into the gradient sum (clip-scale-sum). */
int dpc_gemm_f32(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                 const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                 const float* bias, int epi, float* aux_out, int64_t ld_aux_out, const float* aux_in,
                 int64_t ld_aux_in, const uint8_t* keep, int64_t ld_keep, const uint64_t* rng,
                 uint32_t site, float leak, int splitk, const float* kscale, void* stream);

/* Linear over an input whose trailing columns are a one-/two-hot conditional vector (the generator's
 * cat([z, cond]) input, dp_cgan.py:606, and Residual's cat([out, input]), :83): A and B cover the dense
 * columns only; the hot columns' share of x W^T - the sum of those columns of W - is added in the
 * epilogue from gat = W[:, cond columns]^T (fp32 [C][ldg]) by row ids gidx int32 [M][2] (negative =
 * none).  C = epi(alpha * A.B + bias + gat[gidx[m][0]] + gat[gidx[m][1]]), epi 0 or 3 (ReLU). */
int dpc_gemm_f32_gather(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                        const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc,
                        const float* bias, int epi, const float* gat, int64_t ldg, const int32_t* gidx,
                        void* stream);

/* One GEMM in full: what dpc_gemm_f32 / dpc_gemm_f32_gather take, plus the per-call choices that used to
 * be library state and the segment maps.  Zero-fill it, then set what the product needs.
 *   splitk  0 automatic, 1 never (no fp32 atomics: bit-reproducible results), n > 1 forced;
 *   tile    0 automatic, 64 (64x64 block tile), 128 (128x128) or 32 (32x32 tiles whose four waves split K:
 *           the small-product kernel; 324 = the same with groups of 4 K-tiles);
 *   gat / ldg / gidx  the row gather of dpc_gemm_f32_gather;
 *   seg_*   segment map: index x of a compact axis made of segments of seg_pad entries addresses element
 *           (x / seg_pad) * seg_stride + x % seg_pad of the stored axis, the first seg_len entries of a
 *           segment being meaningful.  It runs a product over the DATA columns of the discriminator's
 *           pac-row layout [pac][Dd data | C cond] (seg_stride = Dd + C; dp_cgan.py:66-68, 622-623) in place:
 *           seg_b 1 = along K of a k-contiguous B (x W1^T; seg_pad % 32 == 0), seg_b 2 = along N of an
 *           n-contiguous B (dy W1; seg_pad % 4 == 0), seg_c 1 = along N of C (dy^T x accumulated into W1's
 *           layout, beta == 1; columns beyond seg_len of a segment are not stored);
 *   colstat (nullable, fp64 [2][N]) += column sums and column sums of squares of the result (epi 0, no
 *           split-K): BatchNorm1d's batch statistics (dp_cgan.py:75-76) taken in the Linear's epilogue with
 *           fp64 atomics; the caller zeroes it.  stat_slots S > 1: colstat is [S][2][N] and the tiles of tile row t add
 *           into slot t mod S (S-fold fewer atomics per address); the consumer sums the S slots (they have the layout
 *           of dpc_colreduce_partial's row-chunk partials: dpc_bn_relu_*_fused / dpc_colreduce_fold take them as R = S);
 *   bn_y    (nullable; with colstat, bn_z, bn_mean, bn_invstd, stat_n) the statistics take BatchNorm's BACKWARD form
 *           over the first stat_n columns of the final result dy (after beta * C): colstat[n] += sum dyh,
 *           colstat[stat_n + n] += sum dyh * xhat, dyh = dy where the ReLU output bn_y > 0, xhat = (bn_z - mean) *
 *           invstd - the two column sums of autograd's batch_norm backward (dp_cgan.py:76-77), taken by the product
 *           that completes the block's output gradient;
 *   atomic  != 0: the product is ADDED to C with fp32 atomics whether or not K is split (beta == 1, plain
 *           product): for products that accumulate into one C concurrently on different streams. */
typedef struct dpc_gemm_desc {
  int32_t M, N, K;
  float alpha, beta;
  const float* A; int64_t a_rs, a_cs;
  const float* B; int64_t b_rs, b_cs;
  float* C; int64_t ldc;
  const float* bias;
  int32_t epi;
  float leak;
  float* aux_out; int64_t ld_aux_out;
  const float* aux_in; int64_t ld_aux_in;
  const uint8_t* keep; int64_t ld_keep;
  const uint64_t* rng; uint32_t site;
  int32_t splitk, tile;
  const float* kscale;
  const float* gat; int64_t ldg; const int32_t* gidx;
  int32_t seg_b, seg_c, seg_len, seg_pad; int64_t seg_stride;
  double* colstat;
  int32_t atomic;
  const float* bn_y; int64_t ld_bn_y; const float* bn_z; int64_t ld_bn_z; const float* bn_mean; const float* bn_invstd;
  int32_t stat_n;
  int32_t stat_slots;
  /* epi 1 only, optional: seed_out[m,n] = (seed_row[m] * seed_col[n]) * slope[m,n] - for the discriminator's LAST
   * hidden layer the seed of the backward chain, rowscale * w_head * slope (what dpc_disc_out_fwd's `delta` is):
   * the chain's first product can then follow the forward product directly (dp_cgan.py:629-631,652-655) */
  float* seed_out; int64_t ld_seed; const float* seed_row; const float* seed_col;
} dpc_gemm_desc;
/* `count` (1..64) products, descriptors in host memory, issued in order on `stream`.  The library keeps no
 * GEMM state: two callers in one process cannot disturb each other's tiling or reproducibility. */
int dpc_gemm_f32_ex(const dpc_gemm_desc* d, int count, void* stream);
/* `count` (1..64) INDEPENDENT products - none reads what another of the set writes; products that add into one C
 * ask for `atomic` - in as few launches as their kernels allow: those that run on the 64x64 tile kernel with the
 * same operand layouts (e.g. the weight gradients dy^T x of every layer of an update: autograd's per-layer
 * addmm calls of loss.backward(), dp_cgan.py:652-655,720-721) share ONE grid, four at a time; the others are issued
 * one by one in the order given.  Same results as dpc_gemm_f32_ex on each descriptor. */
int dpc_gemm_f32_group(const dpc_gemm_desc* d, int count, void* stream);

/* ---- generator output activation (DPCGANSynthesizer._apply_activate, dp_cgan.py:309-327) -- */
/* spans: int32 [nspan][3] = (offset, dim, kind) kind 0 tanh, 1 softmax((x - log E)/tau), E~Exp(1)
 * from exp_noise[b*ld_noise + col] or Philox idx b*Dd + col. */
int dpc_apply_activate_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const int32_t* spans,
                           int nspan, int64_t B, int Dd, const float* exp_noise, int64_t ld_noise,
                           const uint64_t* rng, uint32_t site, float tau, void* stream);
/* autograd of the above: dx (+)= J^T dy given the saved output y (rows of Dd columns). */
int dpc_apply_activate_bwd(const float* y, int64_t ldy, const float* dy, int64_t lddy, float* dx,
                           int64_t lddx, const int32_t* spans, int nspan, int64_t B, int Dd, float tau,
                           int accumulate, void* stream);
/* The generator update's two passes over its raw output in one launch: y = _apply_activate(x)
 * (dp_cgan.py:697, as dpc_apply_activate_fwd) and the conditional loss on the RAW x (dp_cgan.py:711, as
 * dpc_cond_loss: closed form of _cond_loss_pair): dfake (nullable) is WRITTEN with d loss / d x (zero
 * outside the discrete columns), partials[0 : *n_used] (fp64, *n_used <= n_partials, host int*) receive
 * block sums of the un-scaled loss: loss = 2/(B_total*nd) * sum(partials[0 : *n_used]) (dpc_sum_f64). */
int dpc_activate_cond_loss_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, const int32_t* spans,
                               int nspan, int64_t B, int Dd, const float* exp_noise, int64_t ld_noise,
                               const uint64_t* rng, uint32_t site, float tau, const float* cond, int64_t ldc,
                               const int32_t* mask, int64_t ldm, const int32_t* col2d, const int32_t* col2c,
                               int nd, int64_t B_total, float* dfake, int64_t lddf, double* partials,
                               int n_partials, int* n_used, void* stream);

/* ---- conditional loss (DPCGANSynthesizer._cond_loss_pair, dp_cgan.py:330-389), closed form --- */
/* loss_out[0] = 2/(B*nd) sum_b sum_j mask[b,col2d[j]] * BCEWithLogits(fake[b,j], cond[b,col2c[j]]);
 * dfake (nullable) gets d loss / d fake (zero outside discrete columns).  partials: fp64 workspace
 * of n_partials (>= 1, 2048 is enough) elements.  B_total > 0: these B rows are one rank's shard of a
 * global batch of B_total rows (the 1/B of the mean uses B_total, so shares add up over ranks).
 * loss_out may be NULL when n_used (host int*) is given: the loss is then
 * 2/(B_total*nd) * sum(partials[0 : *n_used]), to be taken later with dpc_sum_f64 (the step does not
 * need the scalar; the reference reads it once per epoch, :735-736). */
int dpc_cond_loss(const float* fake, int64_t ldf, const float* cond, int64_t ldc, const int32_t* mask,
                  int64_t ldm, const int32_t* col2d, const int32_t* col2c, int64_t B, int Dd, int nd,
                  float* loss_out, float* dfake, int64_t lddf, int accumulate, double* partials,
                  int n_partials, int64_t B_total, int* n_used, void* stream);

/* ---- DataSampler (data_sampler.py) ---------------------------------------------------------- */
/* Tables (replace DataSampler.__init__, data_sampler.py:9-198): bitmaps[C][W] row sets per
 * category, cnt[C][C] co-occurrence counts.  cat_col[a] = encoded-matrix column of category a. */
int dpc_build_bitmaps(const float* table, int64_t ld, int64_t N, const int32_t* cat_col, int C,
                      uint64_t* bitmaps, int64_t W, void* stream);
/* cntblk (nullable) int32 [C][C][S]: cnt[a][b] split over S (1..64) superblocks of ceil(W/S) bitmap
 * words; it lets dpc_row_select scan one superblock instead of the whole bitmaps. */
int dpc_cooccurrence(const uint64_t* bitmaps, int64_t W, int C, int32_t* cnt, int32_t* cntblk, int S,
                     void* stream);
/* sample_condvec_pair (data_sampler.py:245-343).  pairs_inj int32 [B][2] = the ordered draw of
 * np.random.choice(nd, 2, replace=False); r_inj fp64 [B] = np.random.rand(B).  Writes cond to up
 * to three destinations, mask int32, pair_id / val_idx int64, and for the row pick: sel int32
 * [B][2] (the two chosen category ids) and gcount int64 [B] (size of the joint-value group). */
int dpc_condvec_pair_sample(const int32_t* cnt, const int32_t* cond_st, const int32_t* dims, int nd,
                            int C, int64_t N, const int32_t* pairs_inj, const double* r_inj,
                            const uint64_t* rng, uint32_t site, int64_t B, float* cond0, int64_t ld0,
                            float* cond1, int64_t ld1, float* cond2, int64_t ld2, int32_t* mask,
                            int64_t ldm, int64_t* pair_id, int64_t* val_idx, int32_t* sel,
                            int64_t* gcount, void* stream);
/* The inputs of one network update in ONE launch (three kernel nodes less at the top of the step): the cond
 * vectors exactly as dpc_condvec_pair_sample draws them from `rng` / `site` (data_sampler.py:245-343), the latent
 * batch z [zrows][zcols] ~ N(0,1) (fakez = torch.normal(mean, std), dp_cgan.py:598,681; the same draws as dpc_fill
 * kind 1 at site_z; z nullable) and the zero fill of zero_bytes bytes at `zero` (the update's accumulator pool;
 * 16-byte aligned, a multiple of 16; nullable with 0 bytes). */
int dpc_step_inputs(const int32_t* cnt, const int32_t* cond_st, const int32_t* dims, int nd, int C, int64_t N,
                    const uint64_t* rng, uint32_t site, int64_t B, float* cond0, int64_t ld0, float* cond1,
                    int64_t ld1, float* cond2, int64_t ld2, int32_t* mask, int64_t ldm, int64_t* pair_id,
                    int64_t* val_idx, int32_t* sel, int64_t* gcount, float* z, int64_t ldz, int64_t zrows,
                    int64_t zcols, uint32_t site_z, void* zero, int64_t zero_bytes, void* stream);
/* sample_data_pair (data_sampler.py:382-396): row_id[b] = k-th smallest row id whose two
 * categories are sel[perm ? perm[b] : b]; k = k_inj[b] or floor(u * gcount).  -1 for an empty group. */
int dpc_row_select(const uint64_t* bitmaps, int64_t W, const int32_t* sel, const int64_t* gcount,
                   const int64_t* perm, const int64_t* k_inj, const uint64_t* rng, uint32_t site,
                   int64_t B, int64_t* row_id, const int32_t* cntblk, int C, int S, void* stream);
/* tables without discrete columns (data_sampler.py:388-390): row_id[b] = floor(u_b * N) */
int dpc_row_uniform(int64_t N, const double* u_inj, const uint64_t* rng, uint32_t site, int64_t B,
                    int64_t* row_id, void* stream);
/* self._data[idx] (data_sampler.py:396) + the cond vector that travels with it (c_pair_1[perm],
 * dp_cgan.py:607): with sel != NULL its C columns are written behind the row (C may be 0) and / or its two
 * hot ids to sel_out int32 [B][2] (nullable). */
int dpc_row_gather(const float* table, int64_t ld, const int64_t* row_id, int64_t B, int Dd,
                   float* out, int64_t ldo, const int32_t* sel, const int64_t* perm, int C,
                   int32_t* sel_out, void* stream);
/* np.random.shuffle(arange(B)) (dp_cgan.py:594-595) in distribution: a keyed Feistel bijection of
 * [0, 2^bits) cycle-walked into [0, B); keys_ws is unused (kept for ABI stability, may be NULL). */
int dpc_random_perm(const uint64_t* rng, uint32_t site, int64_t B, int64_t* perm, uint64_t* keys_ws,
                    void* stream);

/* ---- the conditional vector as one / two hot columns per row -------------------------------- */
/* sel[b][0..1] = ascending column ids of the nonzero entries of cond[b, :] (what sample_condvec_pair,
 * data_sampler.py:284-330, and sample_original_condvec, :348-363, set), -1 padded.  *bad (device
 * int32, nullable) counts rows that are not one-/two-hot {0,1} vectors. */
int dpc_cond_to_sel(const float* cond, int64_t ld, int64_t B, int C, int32_t* sel, int32_t* bad,
                    void* stream);
/* Up to 16 transposes in one launch, dst_k[c][r] (+)= src_k[r][c]; all array arguments are HOST arrays of
 * nslices entries.  Keeps W[:, cond columns]^T of the generator layers and of the discriminator's first layer
 * current (gat of dpc_gemm_f32_gather / dpc_disc_first_layer_act) and writes the cond-column gradients
 * (dpc_hot_rows_sum) back in dW's layout (accumulate != 0: added to what dW holds). */
int dpc_transpose_slices(int nslices, const float* const* src, const int64_t* ld_src, const int* rows,
                         const int* cols, float* const* dst, const int64_t* ld_dst, int accumulate, void* stream);
/* Counting sort of (row, hot id) pairs by id: order[p] = row, sid[p] = id for p < *n_items (device int32), ids
 * ascending; order / sid hold 2B (4B with sel_b) entries.  Pairs come from sel_a (and sel_b, nullable) int32
 * [B][2], negative = none.  pac > 1 groups the B sub-rows pac to a row as Discriminator.forward does
 * (dp_cgan.py:66-68): sub-row b of table t gives row t*(B/pac) + b/pac and id (b % pac)*C + sel[b][h], the
 * cond column inside the pac row.  Any number of ids (multi-block). */
int dpc_hot_sort(const int32_t* sel_a, const int32_t* sel_b, int64_t B, int pac, int C, int32_t* order,
                 int32_t* sid, int32_t* n_items, void* stream);
/* acc[id][j] += sum of rowscale[row] * dy[row][j] over the sorted pairs (rowscale nullable = 1): the cond
 * columns of dW = dY^T X (autograd of the Linear layers above), transposed [ids][ld_acc].  One fp32 atomic per
 * id change and block: not bit-reproducible.  max_items = capacity of order / sid. */
int dpc_hot_rows_sum(const float* dy, int64_t ld, int F, const int32_t* order, const int32_t* sid,
                     const int32_t* n_items, int64_t max_items, float* acc, int64_t ld_acc,
                     const float* rowscale, void* stream);

/* ---- DataTransformer.transform (data_transformer.py:102-150) -------------------------------- */
/* tab fp64 [4][K] = mu, sigma, precision_cholesky, A (A = the x-independent part of sklearn's
 * weighted log-probability, see oracle/encode.py).  u_inj fp64 [N] = the uniform behind
 * np.random.choice(p=).  Writes value at col_off and the one-hot of the picked component after it. */
int dpc_encode_continuous(const double* x, int64_t x_stride, int64_t N, const double* tab, int K,
                          const int32_t* valid_idx, int nv, const double* u_inj, const uint64_t* rng,
                          uint32_t site, double* out64, float* out32, int64_t ldo, int col_off,
                          int32_t* comp_out, void* stream);
int dpc_encode_onehot(const int32_t* codes, int64_t N, int dim, double* out64, float* out32,
                      int64_t ldo, int col_off, void* stream);

/* ---- DataTransformer._fit_continuous (data_transformer.py:36-57) ---------------------------- */
/* sklearn's BayesianGaussianMixture (one feature, 'full' covariance, dirichlet_process weights,
 * init_params='random') for F independent fits side by side; fit f reads column fit_col[f] of the
 * column-major fp64 table x[col*col_stride + row].  One iteration = dpc_gmm_estep + dpc_gmm_mstep.
 *   params fp64 [F][8][K]: mu, precision_cholesky, A (see dpc_encode_continuous), covariance,
 *                          stick a, stick b, mean_precision, degrees_of_freedom;
 *   prior  fp64 [F][2]:    mean_prior (column mean), covariance_prior (column variance, ddof 1);
 *   info   fp64 [F][4]:    lower bound, n_iter, converged, done - zero it before the init pass; once
 *                          done != 0 both calls leave that fit alone, so iterations can be queued
 *                          blindly and info polled now and then;
 *   part   fp64 [F][R][3K+1]: per-row-block sufficient statistics (sum r, sum r d, sum r d^2 per
 *                          component with d = x - mu_k, then sum r log r), added in a fixed order.
 * init != 0: the pass over the initial responsibilities (_base.py:121-125, _bayesian_mixture.py:
 * 515-530): resp_inj fp64 [F][N][K] (already normalised rows) or Philox uniforms of site + f. */
int dpc_gmm_estep(const double* x, int64_t col_stride, int64_t N, const int32_t* fit_col, int F,
                  const double* params, int K, const double* prior, const double* info, int init,
                  const double* resp_inj, const uint64_t* rng, uint32_t site, double* part, int R,
                  void* stream);
/* _m_step + _compute_lower_bound + the convergence test of fit_predict (_bayesian_mixture.py:
 * 532-628,725-838; _base.py:247-260) for every fit that is not done. */
int dpc_gmm_mstep(const double* part, int R, int F, int K, const double* prior, double* params,
                  double* info, int init, double weight_concentration_prior,
                  double mean_precision_prior, double dof_prior, double reg_covar, double tol,
                  int max_iter, void* stream);

/* ---- sample() path (DPCGANSynthesizer.sample, dp_cgan.py:838-888) ----------------------------- */
/* sample_original_condvec (data_sampler.py:348-363): one one per row of cond at cond_st[col] +
 * argmax(table[row, st:st+dims[col]]), st = matrix_st ? matrix_st[col] : 0 (the reference's offsets
 * are never filled, i.e. all 0).  rows_inj / cols_inj: np.random.randint's draws (int64, nullable). */
int dpc_condvec_original(const float* table, int64_t ld, int64_t N, const int32_t* cond_st,
                         const int32_t* dims, const int32_t* matrix_st, int nd, int C,
                         const int64_t* rows_inj, const int64_t* cols_inj, const uint64_t* rng,
                         uint32_t site, int64_t B, float* cond, int64_t ldc, void* stream);
/* DataTransformer.inverse_transform (data_transformer.py:152-213) of generated rows.  desc int32
 * [ncol][4] = (kind 0 continuous / 1 discrete, data offset, width, parameter offset); params fp64:
 * per continuous column [mean(Kc) | std(Kc) | valid component indices (as doubles)].  out fp64
 * [n][ncol]: de-normalised value, or the category index of a discrete column. */
int dpc_inverse_transform(const float* data, int64_t ld, int64_t n, const int32_t* desc, int ncol,
                          const double* params, int Kc, double* out, int64_t ldo, void* stream);

/* ---- generator / discriminator glue (dp_cgan.py:31-100) ------------------------------------- */
/* Column reductions, two steps so that a data-parallel run can all-reduce `part` in between.
 * mode 0: sum x, sum x^2 (BatchNorm1d batch statistics, :76); 1: sum w[r]*x (bias gradients);
 * 2: sum dyh, sum dyh*xhat (BatchNorm backward).  part: R*nout*F fp64, R returned in *R_out (host). */
int dpc_colreduce_partial(int mode, const float* x, int64_t ldx, int64_t rows, int F, const float* w,
                          const float* y, int64_t ldy, const float* z, int64_t ldz, const float* mean,
                          const float* invstd, double* part, int max_R, int* R_out, void* stream);
/* kind 0: mean, invstd = f(part, count, eps) and the running-statistics update (momentum);
 * kind 1/2: plain sums to out0 (and out1). */
int dpc_colreduce_finish(const double* part, int R, int nout, int F, int kind, double count, float eps,
                         float momentum, float* out0, float* out1, float* run_mean, float* run_var,
                         int accumulate, void* stream);
/* out[set*out_stride + j] = sum_r part[set*set_stride + r*width + j], j < width = nout*F, in dpc_colreduce_finish's
 * summation order: the ONE fp64 vector per BatchNorm layer that a data-parallel step all-reduces (sync BatchNorm:
 * the batch statistics of dp_cgan.py:76 over the global batch) instead of the R row-chunk partials; a consumer
 * given it with R = 1 computes bit-identical sums.  out may be part (in place). */
int dpc_colreduce_fold(const double* part, int R, int width, int nsets, int64_t set_stride, double* out,
                       int64_t out_stride, void* stream);
/* n (<= 8) weighted column sums out_p[f] (+)= sum_r w_p[r] * x_p[r*ldx_p + f] in two launches for all of them:
 * the bias gradients and the head layer's weight gradient of one backward pass (autograd of the Linear
 * layers, dp_cgan.py:40,43,75,95).  All array arguments are HOST arrays of n entries, w[p] nullable; part:
 * fp64 workspace of part_cap >= 64 * sum(F) elements.  Bit-identical to dpc_colreduce_partial(mode 1) +
 * dpc_colreduce_finish(kind 1) per sum.  accumulate 0: out = sums, 1: out += sums, 2: ONE launch - the row chunks'
 * fp64 sums are added to out with fp32 atomics (out is accumulated into: the caller zeroes it; part may be NULL;
 * summation order not fixed). */
int dpc_colsum_multi(int n, const float* const* x, const int64_t* ldx, const int64_t* rows, const int* F,
                     const float* const* w, float* const* out, int accumulate, double* part, int64_t part_cap,
                     void* stream);
/* dpc_colsum_multi that also does dpc_state_advance's bookkeeping (adv nullable): its first block bumps the counters
 * when it starts.  For the LAST column-sum launch of a training step, where no kernel that runs beside or behind it
 * draws random numbers or reads the step counters - except an optimiser launch told so (t_host = 0: the counter it
 * reads has been bumped already).  No reference counterpart (host-side counters there). */
int dpc_colsum_multi_adv(int n, const float* const* x, const int64_t* ldx, const int64_t* rows, const int* F,
                     const float* const* w, float* const* out, int accumulate, double* part, int64_t part_cap,
                     const dpc_step_advance* adv, void* stream);
/* out[f] (+)= sum_r w[r] * x[r*ldx + f] (w nullable = all ones): dpc_colreduce_partial(mode 1, max_R 64) +
 * dpc_colreduce_finish(kind 1) in one call that ALLOCATES NOTHING and takes no workspace: the row-chunk
 * partial sums live in a 4 MiB file-scope device array of the library (one copy per GPU, held until the
 * process ends).  No counterpart in the reference; it is for callers that cannot allocate (HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  Bit-identical to that pair of calls.
 * F <= 8192 (DPC_EINVAL beyond).  Calls that overlap on one GPU share the array and race.
 * x, out and w must live on the CURRENT device (DPC_EINVAL otherwise: the current device's copy of the array is the
 * one used).  The first call on a device looks the array's address up (not a capturable call): make one warm-up
 * call per device outside stream capture; later calls there are two launches and nothing else. */
int dpc_colsum_noalloc(const float* x, int64_t ldx, int64_t rows, int F, const float* w, float* out,
                       int accumulate, void* stream);
int dpc_bn_relu_fwd(const float* z, int64_t ldz, float* y, int64_t ldy, int64_t B, int F,
                    const float* mean, const float* invstd, const float* gamma, const float* beta,
                    void* stream);
int dpc_bn_relu_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* z,
                    int64_t ldz, int64_t B, int F, const float* mean, const float* invstd,
                    const float* gamma, const float* dbeta, const float* dgamma, double count,
                    float* dz, int64_t lddz, void* stream);
/* The two calls above in one launch (a kernel node costs ~5 us on the step's critical path whatever it
 * does): every block first finishes the reduction for its 64 columns from `part` (the R x 2 x F
 * sums of dpc_colreduce_partial mode 0, all-reduced over ranks if any), then normalises its rows;
 * mean_out / invstd_out and the running statistics (nullable) are written as dpc_colreduce_finish
 * kind 0 does. */
int dpc_bn_relu_fwd_fused(const float* z, int64_t ldz, float* y, int64_t ldy, int64_t B, int F,
                          const double* part, int R, double count, float eps, float momentum,
                          const float* gamma, const float* beta, float* mean_out, float* invstd_out,
                          float* run_mean, float* run_var, void* stream);
/* Likewise for the backward: part = mode-2 sums; writes dbeta, dgamma and dz.  grad_scale multiplies
 * the dbeta / dgamma written (not the ones dz uses): 1/world when a sum over ranks follows. */
int dpc_bn_relu_bwd_fused(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* z,
                          int64_t ldz, int64_t B, int F, const double* part, int R, double count,
                          const float* mean, const float* invstd, const float* gamma, float* dbeta,
                          float* dgamma, float* dz, int64_t lddz, float grad_scale, void* stream);
/* last Linear(d, 1) of the discriminator (:43); when delta != NULL the same pass writes the seed
 * delta[r,j] = rowscale[r] * w[j] * slope[r,j] of the backward / input-gradient chain */
int dpc_disc_out_fwd(const float* h, int64_t ldh, const float* w, const float* b, int64_t rows, int d,
                     float* y, const float* rowscale, const float* slope, int64_t lds, float* delta,
                     int64_t ldd, void* stream);
/* delta[r,j] = rowscale[r] * w[j] * slope[r,j]: seed of the backward / input-gradient chain */
int dpc_disc_seed(const float* rowscale, const float* w, const float* slope, int64_t lds, int64_t rows,
                  int d, float* delta, int64_t ldd, void* stream);
/* interpolates of calc_gradient_penalty (:47-51): one alpha per pac group of group_len elements */
int dpc_interpolate(const float* real, const float* fake, float* out, int64_t n, int64_t group_len,
                    const float* alpha_inj, const uint64_t* rng, uint32_t site, float* alpha_out,
                    void* stream);
/* (:61-62) in place g0 -> coef*(||g0||-1)/||g0|| * g0, pen_rows[r] = (||g0||-1)^2 */
int dpc_gp_norm(float* g0, int64_t ld, int64_t n, int K0, float coef, double* pen_rows, float* nrm_out,
                void* stream);
/* Discriminator update, first hidden layer over the stacked rows [fake | real | interpolates]
 * (dp_cgan.py:629-630 and :47-53) without forming the interpolates: Linear is linear, so their
 * pre-activations are alpha*(real W^T) + (1-alpha)*(fake W^T) with one alpha per pac group.  zpre
 * [2n, d] = plain products of the fake rows then the real rows (dpc_gemm_f32, no bias); writes
 * H = pre*slope and S = slope = (pre > 0 ? 1 : leak)*keep*2 for all 3n rows, the arithmetic of
 * dpc_gemm_f32's epilogue 1 (keep u8 [3n, d] or Philox draw row*d + col of site_keep; alpha_inj [n]
 * or Philox draw r of site_alpha, as dpc_interpolate).  stacked == 0: zpre [n, d] -> H, S [n, d], the
 * plain Linear + LeakyReLU + Dropout of n rows.  clear_zpre != 0: every product is set to 0 once read,
 * so that the next split-K product can add into the buffer without a memset in front of it.
 * gat != NULL: the cond columns of the input (cat([data, cond]), dp_cgan.py:622-623) stayed off the matrix
 * cores - zpre holds the products over the data columns only (segment-mapped GEMM) and every row adds, for
 * each of its pac sub-rows s and each of that sub-row's (up to two) hot cond ids c, row s*C + c of
 * gat = W1[:, cond columns]^T [pac*C][ldg]; sel_a int32 [n*pac][2] ids of the first block of rows (fake),
 * sel_b of the second (real; stacked only), negative = none.  With clear_zpre == 0 the gathered share is
 * also added to the stored zpre, which then holds x W1^T over all columns (dpc_persample_clip_all_gram reads it). */
int dpc_disc_first_layer_act(float* zpre, int64_t ldz, const float* bias, int64_t n, int d,
                             const float* alpha_inj, const uint8_t* keep, int64_t ld_keep,
                             const uint64_t* rng, uint32_t site_alpha, uint32_t site_keep, float leak,
                             float* H, int64_t ldh, float* S, int64_t lds, float* alpha_out, int stacked,
                             int clear_zpre, const float* gat, int64_t ldg, const int32_t* sel_a,
                             const int32_t* sel_b, int pac, int C, void* stream);
/* calc_gradient_penalty (:55-62) through the first layer's Gram matrix G = W1 W1^T: with u [n, d] the
 * penalty rows of the first layer's backward delta and T = u G,  ||grad_xhat D||^2 = rowsum(T*u)
 * and the penalty rows of the next v-chain link are v1 = f*T*slope, f = coef*(nrm-1)/nrm.  Writes
 * pen_rows[r] = (nrm-1)^2, f_out[r] = f, q_out[r] = ||grad||^2 (fp64, nullable), nrm_out (nullable). */
int dpc_gp_gram_rows(const float* T, int64_t ldt, const float* u, int64_t ldu, const float* slope,
                     int64_t lds, int64_t n, int d, float coef, double* pen_rows, float* f_out,
                     double* q_out, float* nrm_out, float* v1, int64_t ldv, void* stream);
int dpc_sum_f64(const double* v, int64_t n, double scale, float* out, int accumulate, void* stream);
int dpc_mean_pair(const float* y, int64_t n, float sa, float sb, float* out, int accumulate,
                  void* stream);

/* ---- gradient post-processing + optimiser (dp_cgan.py:510-518, 638-661) ---------------------- */
/* One launch over a flat parameter buffer: g_eff = gscale*g + seg_add[segment] + noise_scale*noise,
 * coupled-L2 Adam, optional clamp to +-clamp.  Step count t = (t_dev ? *t_dev : 0) + t_host, t_host >= 1.
 * seg_off int64 [nseg+1] device, seg_add fp32 [nseg] device (the reference's per-tensor scalar
 * hook noise); noise_inj fp32 [n] or Philox normal idx i. */
int dpc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                  float eps, float wd, float clamp, float gscale, const int32_t* t_dev, int t_host,
                  const int64_t* seg_off, const float* seg_add, int nseg, float noise_scale,
                  const float* noise_inj, const uint64_t* rng, uint32_t site, void* stream);
/* The same step over a parameter buffer whose matrices hold COND SLICES - columns that multiply a two-hot cond vector
 * (the generator's Linears over cat([.., cond]), dp_cgan.py:83,606; the discriminator's first Linear over pac
 * sub-rows [data | cond], :66-68,622-623) - in ONE launch instead of dpc_transpose_slices + dpc_adam_step +
 * dpc_transpose_slices.  Slice k = columns [s_col0[k], s_col0[k] + s_ncol[k]) of the matrix of s_rows[k] rows of
 * length s_ld[k] at flat offset s_off[k]: its gradient takes the transposed addend s_gT[k] [ncol][s_ld_gT[k]]
 * (nullable: the sorted row sums of dpc_hot_rows_sum; the sum is written back to g) and its updated parameters are
 * also written transposed to s_pT[k] [ncol][s_ld_pT[k]] (nullable: W[:, cond]^T, the row-gather table of the forward).
 * The nmat (<= 8) matrices that own the nslice (<= 16) slices: flat offset m_off, m_rows rows of length m_ld; their
 * cond columns are those with (column mod m_period) >= m_col0.  All table arguments are HOST arrays.  Same
 * arithmetic, segment lookup and noise draw per element as dpc_adam_step. */
int dpc_adam_step_slices(float* p, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                         float eps, float wd, float clamp, float gscale, const int32_t* t_dev, int t_host,
                         const int64_t* seg_off, const float* seg_add, int nseg, float noise_scale,
                         const float* noise_inj, const uint64_t* rng, uint32_t site, int nmat,
                         const int64_t* m_off, const int* m_rows, const int* m_ld, const int* m_period,
                         const int* m_col0, int nslice, const int64_t* s_off, const int* s_rows,
                         const int* s_ld, const int* s_col0, const int* s_ncol, const float* const* s_gT,
                         const int64_t* s_ld_gT, float* const* s_pT, const int64_t* s_ld_pT, void* stream);
/* x = clamp(x + scale * noise): the stand-alone DP noise-add (+ weight clamp) kernel */
int dpc_noise_add_clamp(float* x, int64_t n, float scale, const float* noise_inj, const uint64_t* rng,
                        uint32_t site, float clamp, void* stream);

/* ---- DP-SGD mode (north_star; the reference has no per-example clipping: SURVEY F2) ------------- */
/* An example is one pac group i = rows i, n+i, 2n+i (fake, real, penalty) of the stacked
 * discriminator batch.  norm2[i] (+)= ||gradient of example i w.r.t. one Linear layer||_F^2 computed
 * from the layer's backward deltas [3n, d] and input activations [3n, K] through two 3x3 Gram
 * matrices (no per-example gradient is formed); with_bias adds ||delta_i + delta_{n+i}||^2. */
int dpc_persample_gram(const float* delta, int64_t ldd, int d, const float* act, int64_t lda, int K,
                       int64_t n, int with_bias, int accumulate, double* norm2, void* stream);
/* All layers + clip factors in one launch: delta / act / ldd / lda / d / K are HOST arrays of nlayers
 * (<= 8) entries; every layer counts its bias.  Outputs as dpc_persample_gram + dpc_clip_factors. */
int dpc_persample_clip_all(int nlayers, const float* const* delta, const int64_t* ldd, const int* d,
                           const float* const* act, const int64_t* lda, const int* K, int64_t n, float clip,
                           float gscale, double* norm2, float* rowc, const float* roww, float* rowc_w,
                           float* norm_out, void* stream);
/* The same when the penalty's first-layer input row v0_i = f_i * (u_i W1) is not materialised
 * (dpc_gp_gram_rows): act[0] is [2n, K[0]] (fake and real rows only), zpre [2n, d[0]] the first
 * layer's plain products x W1^T, gp_f / gp_q the outputs of dpc_gp_gram_rows; the Gram entries of the
 * missing row are f_i*(u_i . zpre_i), f_i*(u_i . zpre_{n+i}) and f_i^2 * q_i.  fc_out[i] (nullable)
 * = c_i * gp_f[i].  clear_zpre != 0: zpre is set to 0 once read (see dpc_disc_first_layer_act).
 * sel_a != NULL: act[0] holds the DATA columns of the pac-row input only (zero padded) and the inner
 * products of the rows' two-hot cond parts are counted from the hot ids, sel_a / sel_b int32 [n*pac][2] of
 * the fake / real sub-rows (as in dpc_disc_first_layer_act). */
int dpc_persample_clip_all_gram(int nlayers, const float* const* delta, const int64_t* ldd, const int* d,
                                const float* const* act, const int64_t* lda, const int* K, int64_t n,
                                float clip, float gscale, double* norm2, float* rowc, const float* roww,
                                float* rowc_w, float* norm_out, float* zpre, int64_t ldz,
                                const float* gp_f, const double* gp_q, float* fc_out, int clear_zpre,
                                const int32_t* sel_a, const int32_t* sel_b, int pac, void* stream);
/* c_i = min(1, clip / (gscale*sqrt(norm2[i]))) -> rowc[p*n+i] (p = 0..2), rowc_w = rowc * roww
 * (nullable), norm_out[i] = gscale*sqrt(norm2[i]) (nullable). */
int dpc_clip_factors(const double* norm2, int64_t n, float clip, float gscale, float* rowc,
                     const float* roww, float* rowc_w, float* norm_out, void* stream);
/* Reference private mode (dp_cgan.py:638-647): seg_add[s] = N(0,1)_s * sqrt(fires[s]*(*t_dev + t_add)) * inv_B,
 * the sum of the scalar noises that the accumulated hook sets add to parameter tensor s. */
int dpc_hook_noise(float* seg_add, int nseg, const float* fires, const int32_t* t_dev, int t_add, float inv_B,
                   const uint64_t* rng, uint32_t site, void* stream);

#ifdef __cplusplus
}
#endif
#endif
