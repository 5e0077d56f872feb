// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   C[M,N] = epi( alpha * sum_k A(m,k) * B(k,n) + bias[n] ) (+ beta * C)
//
// A(m,k) and B(k,n) are addressed with two strides each, one of which must be 1; that covers the
// three products the DP-CGAN step needs without any transposed copies:
//   forward  y  = x W^T   A = x  (k contiguous)   B = W viewed (k, n) -> W[n,k]  (k contiguous)
//   input    dx = dy W    A = dy (k contiguous)   B = W[k,n]                      (n contiguous)
//   weight   dW = dy^T x  A = dy viewed (m, k) -> dy[k,m] (m contiguous)   B = x[k,n] (n contiguous)
//
// Structure: 256 threads = 4 waves in a 2x2 grid over a BMxBN block tile (128x128 or 64x64), BK = 32,
// operands staged global -> registers -> LDS (image per operand layout: see Stage), two LDS buffers,
// the global loads of tile t+PF issued before the MFMAs of tile t (one barrier per K-tile).  Split-K over
// gridDim.y with fp32 atomics for the weight-gradient shapes (small MxN, K = batch).  Block ids
// are remapped so that the blocks sharing an A row-panel run on one XCD (private L2).
#include "dpc_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector type for the staging registers: HIP's float4 is a struct whose copies lower to
// llvm.memcpy between address spaces, which kept the prefetch ring in scratch memory
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define EPI_NONE 0
#define EPI_LEAKY_DROPOUT 1
#define EPI_MUL 2
#define EPI_RELU 3

struct GemmArgs {
  const float* A; int64_t a_rs, a_cs;
  const float* B; int64_t b_rs, b_cs;
  float* C; int64_t ldc;
  int M, N, K;
  float alpha, beta;
  const float* bias;
  int epi;
  float* aux_out; int64_t ld_aux_out;
  const float* aux_in; int64_t ld_aux_in;
  const uint8_t* keep; int64_t ld_keep;
  const uint64_t* rng; uint32_t site;
  float leak;
  int splitk, kchunk;
  int tiles_m, tiles_n;
  int a_vec, b_vec;
  const float* kscale;   // optional: A(m,k) *= kscale[k] (fused clip-scale-sum; m-contiguous A only)
  // optional row gather added before the epilogue: C[m,n] += gat[gidx[m][0]][n] + gat[gidx[m][1]][n]
  // (negative ids skipped).  It stands for the columns of A that hold a one-/two-hot conditional
  // vector: their product with W^T is the sum of the hot columns of W, i.e. rows of gat = W^T slice.
  const float* gat; int64_t ldg; const int32_t* gidx;
};

__device__ __forceinline__ float gemm_gather2(const GemmArgs& g, int row, int col) {
  const int i0 = g.gidx[2 * row], i1 = g.gidx[2 * row + 1];
  float v = 0.f;
  if (i0 >= 0) v += g.gat[(int64_t)i0 * g.ldg + col];
  if (i1 >= 0) v += g.gat[(int64_t)i1 * g.ldg + col];
  return v;
}

constexpr int BK = 32;

// One operand's staging of a BMN x BK tile: global -> registers (float4) -> LDS.
//   KC  (k contiguous in global: stride_mn = leading dimension): LDS image [mn][BK + 4].  A thread
//       moves 4 consecutive k of one row with one 16-byte load and one ds_write_b128; a lane reads
//       its MFMA operands for FOUR k-steps with one ds_read_b128.  Row stride 36 floats: any 16
//       rows the hardware groups for a b128 access land on 16 different 4-bank sets (conflict-free).
//   !KC (m/n contiguous: stride_k = leading dimension): LDS image [BK][mn + 4]; 16-byte stores,
//       ds_read_b32 fragments (32 consecutive m/n per half-wave: conflict-free).
template <int BMN, bool KC, bool KS = false>
struct Stage {
  static constexpr int PASSES = BMN / 32;
  static constexpr int LD = KC ? (BK + 4) : (BMN + 4);
  static constexpr int SIZE = KC ? BMN * (BK + 4) : BK * (BMN + 4);    // floats per LDS buffer
  f32x4 r[PASSES];
  float sc[KS ? PASSES : 1];   // per-k scale of this thread's k-row (KS: m/n-contiguous operand only)

  // Fast path (block-uniform choice, see `fast` below): straight-line 16-byte loads with no
  // per-lane guard.  Rows/columns beyond the matrix edge are clamped to a valid address instead of
  // zero-filled: they only feed accumulators whose outputs the epilogue never stores.  (Per-lane
  // guarded loads make hipcc branch around every load and drain vmcnt at each join - a dependent
  // L2 round trip per element.)
  __device__ __forceinline__ void load_fast(const float* __restrict__ base, int64_t ld, int mn0, int k0, int MN,
                                            const float* __restrict__ ks = nullptr) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      if (KC) {
        int row = min(mn0 + p * 32 + (t >> 3), MN - 1);
        r[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)row * ld + k0 + 4 * (t & 7));
      } else {
        constexpr int TPR = BMN / 4;
        constexpr int RPP = 256 / TPR;
        int mn = min(mn0 + 4 * (t % TPR), ((MN + 3) & ~3) - 4);
        r[p] = *reinterpret_cast<const f32x4*>(base + (int64_t)(k0 + p * RPP + t / TPR) * ld + mn);
        if (KS) sc[p] = ks[k0 + p * RPP + t / TPR];
      }
    }
  }

  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int mn0, int k0,
                                       int MN, int Kend, bool fast, const float* __restrict__ ks = nullptr) {
    if (fast && k0 + BK <= Kend) {
      load_fast(base, ld, mn0, k0, MN, ks);
      return;
    }
    // Edge / unaligned path: scalar loads from clamped (always valid) addresses, zeroed by select;
    // still branch-free, so the loads stay in flight together.
    const int t = threadIdx.x;
    const int klast = Kend - 1, mlast = MN - 1;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      float e[4];
      if (KC) {
        int row = mn0 + p * 32 + (t >> 3);
        int k = k0 + 4 * (t & 7);
        const float* src = base + (int64_t)min(row, mlast) * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = src[min(k + j, klast)];
          e[j] = (row <= mlast && k + j <= klast) ? v : 0.f;
        }
      } else {
        constexpr int TPR = BMN / 4;            // threads per k-row
        constexpr int RPP = 256 / TPR;          // k-rows per pass
        int mn = mn0 + 4 * (t % TPR);
        int k = k0 + p * RPP + t / TPR;
        const float* src = base + (int64_t)min(k, klast) * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = src[min(mn + j, mlast)];
          e[j] = (k <= klast && mn + j <= mlast) ? v : 0.f;
        }
        if (KS) sc[p] = ks[min(k, klast)];
      }
      r[p] = f32x4{e[0], e[1], e[2], e[3]};
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ s) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      if (KC) {
        *reinterpret_cast<f32x4*>(&s[(p * 32 + (t >> 3)) * LD + 4 * (t & 7)]) = r[p];
      } else {
        constexpr int TPR = BMN / 4;
        constexpr int RPP = 256 / TPR;
        int krow = p * RPP + t / TPR;
        f32x4 v = r[p];
        if (KS) v *= sc[p];
        *reinterpret_cast<f32x4*>(&s[krow * LD + 4 * (t % TPR)]) = v;
      }
    }
  }

  // MFMA operands of the 32-row slab starting at mn for the 8 k-steps kb .. kb+7: lane half h
  // supplies k = kb + 4h + j to the j-th of four v_mfma_f32_32x32x2_f32 (any assignment of the 8
  // k's to the 4 x 2 operand slots is a valid product as long as A and B use the same one).
  __device__ static __forceinline__ void frag(const float* __restrict__ s, int mn, int kb, int li, int lk, float f[4]) {
    if (KC) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&s[(mn + li) * LD + kb + 4 * lk]);
      f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = s[(kb + 4 * lk + j) * LD + mn + li];
    }
  }
};

// One block-tile segment: C[m0.., n0..] (+)= A[m0.., k_begin..k_end) * B[k_begin..k_end, n0..].
// 256 threads = 4 waves as 2 x 2 over a BM x BN block tile; each wave owns TM x TN 32x32 MFMA tiles.
// `atomic`: add the partial sums with fp32 atomics (split-K / stream-K; epilogue and bias are then
// applied by gemm_epilogue_kernel); otherwise store with the fused epilogue, bias if `add_bias`.
template <int BM, int BN, bool A_KC, bool B_KC, int PF, bool KS>
__device__ __forceinline__ void gemm_segment(const GemmArgs& g, float* lds, const int m0, const int n0,
                                             const int k_begin, const int k_end, const bool atomic,
                                             const bool add_bias) {
  static_assert(!(KS && A_KC), "kscale needs an m-contiguous A");
  static_assert(PF % 2 == 0, "the LDS buffer parity is taken from the unrolled tile index");
  using SA = Stage<BM, A_KC, KS>;
  using SB = Stage<BN, B_KC>;
  constexpr int TM = BM / 64, TN = BN / 64;   // 32x32 MFMA tiles per wave in each direction
  float* As = lds;
  float* Bs = lds + 2 * SA::SIZE;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int64_t lda = A_KC ? g.a_rs : g.a_cs;
  const int64_t ldb = B_KC ? g.b_cs : g.b_rs;
  // Register-staged prefetch ring, PF tiles deep: with only 1-2 resident blocks per CU (these GEMMs
  // are small) nothing else hides the ~1 us L2/HBM latency, so the loads of tile t+PF are issued
  // before the MFMAs of tile t and land in LDS PF-1 iterations later.
  // 16-byte-aligned operand, and an edge tile (if any) that may be clamped: always for a
  // k-contiguous operand (row clamp); for an m/n-contiguous one only if the rows are long enough
  // (ld >= extent rounded up to 4) for the last 16-byte load to stay inside its row
  const bool fa = g.a_vec && (A_KC ? g.M >= 1 : (m0 + BM <= g.M || (((g.M + 3) & ~3) <= lda && g.M >= 4)));
  const bool fb = g.b_vec && (B_KC ? g.N >= 1 : (n0 + BN <= g.N || (((g.N + 3) & ~3) <= ldb && g.N >= 4)));
  // PF register stages per operand (a ring indexed by the unrolled tile number: compile-time indices)
  SA sa[PF];
  SB sb[PF];
  const int ntile = (k_end - k_begin + BK - 1) / BK;
  const int nfull = (fa && fb) ? (k_end - k_begin) / BK : 0;
  const int nmain = (nfull / PF) * PF;
  float* const As0 = As;
  float* const Bs0 = Bs;

  auto compute = [&](const float* as, const float* bs) {
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) SA::frag(as, wm0 + i * 32, kb, li, lk, a[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) SB::frag(bs, wn0 + j * 32, kb, li, lk, b[j]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
  };

  // Main pipeline: nmain (a multiple of PF) full K-tiles, strictly branch-free inside (loads past
  // the end are clamped to the last tile, stores are unconditional), so hipcc keeps PF-1 tiles of
  // loads in flight with counted s_waitcnt vmcnt(N) instead of draining to 0 at every control-flow
  // join.  The loads of tile t+PF are issued before the MFMAs of tile t and land in LDS PF-1
  // iterations later: with ~1 block per CU nothing else hides the 1-2 us L2/HBM latency.
  // Row gather (see GemmArgs::gat): the block's BM x BN share of gat[gidx[m][0]] + gat[gidx[m][1]] reaches
  // the epilogue through LDS, loaded with 16-byte loads along the rows of gat (the MFMA output layout
  // holds one column per lane: gathering from there would be 2 x 16 scalar loads per lane and tile).
  // For 64x64 tiles the loads are issued HERE, before the K loop, and sit in registers until the loop
  // is done: ids -> rows of gat is a chain of two dependent L2 round trips, longer than the whole
  // K loop of the generator's first layer (K = 128).
  const bool gather = (g.gat != nullptr) && !atomic;
  constexpr int GV = BN / 4;
  constexpr bool GPRE = (BM * BN <= 4096);
  constexpr int GQ = GPRE ? (BM * GV) / 256 : 1;
  const bool gvec = gather && ((g.ldg & 3) == 0) && (((uintptr_t)g.gat & 15) == 0) && (n0 + BN <= g.N);
  f32x4 gv[GQ];
  if (GPRE && gvec) {
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
      const int idx = threadIdx.x + q * 256;
      const int r = idx / GV, c4 = (idx - r * GV) * 4;
      const int row = min(m0 + r, g.M - 1);
      const int i0 = g.gidx[2 * row], i1 = g.gidx[2 * row + 1];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i0 >= 0) v += *reinterpret_cast<const f32x4*>(g.gat + (int64_t)i0 * g.ldg + n0 + c4);
      if (i1 >= 0) v += *reinterpret_cast<const f32x4*>(g.gat + (int64_t)i1 * g.ldg + n0 + c4);
      gv[q] = v;
    }
  }

  if (nfull > 0) {
    const int last = nfull - 1;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      sa[j].load_fast(g.A, lda, m0, k_begin + min(j, last) * BK, g.M, g.kscale);
      sb[j].load_fast(g.B, ldb, n0, k_begin + min(j, last) * BK, g.N);
    }
    sa[0].store(As0);
    sb[0].store(Bs0);
    __syncthreads();
    for (int t0 = 0; t0 < nmain; t0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int cur = u & 1;              // PF is even, so t & 1 == u & 1
        const int tl = min(t0 + u + PF, last);
        sa[u].load_fast(g.A, lda, m0, k_begin + tl * BK, g.M, g.kscale);
        sb[u].load_fast(g.B, ldb, n0, k_begin + tl * BK, g.N);
        compute(As + cur * SA::SIZE, Bs + cur * SB::SIZE);
        sa[(u + 1) % PF].store(As + (cur ^ 1) * SA::SIZE);
        sb[(u + 1) % PF].store(Bs + (cur ^ 1) * SB::SIZE);
        __syncthreads();
      }
    }
    // The nfull - nmain (< PF) full tiles left over are in the ring already - the last pass above (or
    // the prologue, when there was none) loaded them, and tile nmain sits in LDS buffer 0: they are
    // consumed from there, still pipelined (they used to take the serial path below, ~3 us a tile).
    const int rem = nfull - nmain;
#pragma unroll
    for (int u = 0; u + 1 < PF; ++u) {
      if (u < rem) {
        const int cur = u & 1;
        compute(As + cur * SA::SIZE, Bs + cur * SB::SIZE);
        if (u + 1 < rem) {
          sa[u + 1].store(As + (cur ^ 1) * SA::SIZE);
          sb[u + 1].store(Bs + (cur ^ 1) * SB::SIZE);
        }
        __syncthreads();
      }
    }
  }
  // Remainder (a partial last tile, or operands the fast path cannot take): one tile at a time.
  for (int t = nfull; t < ntile; ++t) {
    sa[0].load(g.A, lda, m0, k_begin + t * BK, g.M, k_end, fa, g.kscale);
    sb[0].load(g.B, ldb, n0, k_begin + t * BK, g.N, k_end, fb);
    sa[0].store(As0);
    sb[0].store(Bs0);
    __syncthreads();
    compute(As0, Bs0);
    __syncthreads();
  }

  if (gather) {
    static_assert(BM * BN <= 2 * SA::SIZE + 2 * SB::SIZE, "gather image must fit the operand buffers");
    if (GPRE && gvec) {
#pragma unroll
      for (int q = 0; q < GQ; ++q) {
        const int idx = threadIdx.x + q * 256;
        const int r = idx / GV, c4 = (idx - r * GV) * 4;
        *reinterpret_cast<f32x4*>(&lds[r * BN + c4]) = gv[q];
      }
    } else {
      for (int idx = threadIdx.x; idx < BM * GV; idx += 256) {
        const int r = idx / GV, c4 = (idx - r * GV) * 4;
        const int row = m0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < g.M) {
          const int i0 = g.gidx[2 * row], i1 = g.gidx[2 * row + 1];
          if (gvec) {
            if (i0 >= 0) v += *reinterpret_cast<const f32x4*>(g.gat + (int64_t)i0 * g.ldg + n0 + c4);
            if (i1 >= 0) v += *reinterpret_cast<const f32x4*>(g.gat + (int64_t)i1 * g.ldg + n0 + c4);
          } else {
            float e4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int col = n0 + c4 + e;
              float x = 0.f;
              if (col < g.N) {
                if (i0 >= 0) x += g.gat[(int64_t)i0 * g.ldg + col];
                if (i1 >= 0) x += g.gat[(int64_t)i1 * g.ldg + col];
              }
              e4[e] = x;
            }
            v = f32x4{e4[0], e4[1], e4[2], e4[3]};
          }
        }
        *reinterpret_cast<f32x4*>(&lds[r * BN + c4]) = v;
      }
    }
    __syncthreads();
  }

  // Epilogue.  C/D map of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
  // (One straight-line body per MFMA tile, no `continue`: a loop over the tiles that hipcc does not
  // fully unroll indexes `acc` dynamically and moves all accumulators to scratch memory.)
  auto tile_out = [&](const f32x16& av, const int row0, const int col) {
    if (col < g.N) {
      const float bv = (g.bias && add_bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (row < g.M) {
          float v = g.alpha * av[e] + bv;
          float* cp = g.C + (int64_t)row * g.ldc + col;
          if (atomic) {
            atomicAdd(cp, v);
          } else {
            if (gather) v += lds[(row - m0) * BN + (col - n0)];
            if (g.epi == EPI_LEAKY_DROPOUT) {
              float keep;
              if (g.keep) {
                keep = g.keep[(int64_t)row * g.ld_keep + col] ? 1.f : 0.f;
              } else {
                keep = (dpc_draw_u32(g.rng, g.site, (uint64_t)row * g.N + col) >> 31) ? 1.f : 0.f;
              }
              float slope = (v > 0.f ? 1.f : g.leak) * keep * 2.f;
              g.aux_out[(int64_t)row * g.ld_aux_out + col] = slope;
              v *= slope;
            } else if (g.epi == EPI_MUL) {
              v *= g.aux_in[(int64_t)row * g.ld_aux_in + col];
            } else if (g.epi == EPI_RELU) {
              v = fmaxf(v, 0.f);
            }
            if (g.beta != 0.f) v += g.beta * (*cp);
            *cp = v;
          }
        }
      }
    }
  };
  tile_out(acc[0][0], m0 + wm0, n0 + wn0 + li);
  if constexpr (TN > 1) tile_out(acc[0][1], m0 + wm0, n0 + wn0 + 32 + li);
  if constexpr (TM > 1) {
    tile_out(acc[1][0], m0 + wm0 + 32, n0 + wn0 + li);
    if constexpr (TN > 1) tile_out(acc[1][1], m0 + wm0 + 32, n0 + wn0 + 32 + li);
  }
  static_assert(TM <= 2 && TN <= 2, "epilogue is written out for up to 2 x 2 MFMA tiles per wave");
}

// Tile index -> tile coordinates: consecutive ids (= one XCD's run) walk the SHORTER tile axis first,
// so the tiles that share a panel of the larger operand sit in one L2 (a weight gradient with 4 x 77
// tiles used to fetch its 60 MB activation operand once per tile row: 2x the algorithmic HBM/MALL
// traffic by FETCH_SIZE).
__device__ __forceinline__ void gemm_tile_coords(const GemmArgs& g, int wg, int& tile_m, int& tile_n) {
  if (g.tiles_m < g.tiles_n) { tile_n = wg / g.tiles_m; tile_m = wg - tile_n * g.tiles_m; }
  else { tile_m = wg / g.tiles_n; tile_n = wg - tile_m * g.tiles_n; }
}

// Slot of block `bid` among `n` blocks such that the blocks of one XCD (ids congruent mod 8) hold a
// contiguous run of slots (bijective for any n).
__device__ __forceinline__ int gemm_xcd_slot(int bid, int n) {
  int q = n >> 3, r = n & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Data-parallel kernel: one block per tile (gridDim.y > 1: per K chunk of a split-K product).
template <int BM, int BN, bool A_KC, bool B_KC, int PF, bool KS = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Stage<BM, A_KC, KS>::SIZE + 2 * Stage<BN, B_KC>::SIZE];
  // 1-D grid of tiles x splitk blocks.  An XCD's contiguous run of slots is taken in split-major order
  // (slot = split * tiles + tile): with 8 or more K chunks an XCD works on ONE chunk of K for all
  // tiles, so each XCD's L2 fetches its slice of A and B once - the whole product then reads A and B
  // about once from HBM instead of once per tile row / column and XCD.
  const int tiles = g.tiles_m * g.tiles_n;
  const int slot = gemm_xcd_slot(blockIdx.x, tiles * g.splitk);
  const int split = slot / tiles;
  int tile_m, tile_n;
  gemm_tile_coords(g, slot - split * tiles, tile_m, tile_n);
  int k_begin = split * g.kchunk;
  int k_end = min(g.K, k_begin + g.kchunk);
  if (k_begin >= k_end && split > 0) return;
  gemm_segment<BM, BN, A_KC, B_KC, PF, KS>(g, lds, tile_m * BM, tile_n * BN, k_begin, k_end, g.splitk > 1, split == 0);
}

// Stream-K kernel (64x64 tiles): the tiles x K-tiles iteration space is cut into gridDim.x equal
// contiguous runs, one per block, whatever the tile boundaries; a block adds each (partial) tile of
// its run to C with fp32 atomics.  Every CU gets the same amount of MFMA work and pays the pipeline
// fill once per block instead of once per tile - the shapes of the training step (48 x 4 tiles of
// K = 4888, 4 x 11 tiles of K = 8192 ...) never divide evenly over 256 CUs otherwise.
template <bool A_KC, bool B_KC, int PF, bool KS = false>
__global__ __launch_bounds__(256) void gemm_streamk_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Stage<64, A_KC, KS>::SIZE + 2 * Stage<64, B_KC>::SIZE];
  const int ktiles = (g.K + BK - 1) / BK;
  const int64_t units = (int64_t)g.tiles_m * g.tiles_n * ktiles;
  const int slot = gemm_xcd_slot(blockIdx.x, gridDim.x);
  int64_t u = units * slot / gridDim.x;
  const int64_t u_end = units * (slot + 1) / gridDim.x;
  while (u < u_end) {
    const int tile = (int)(u / ktiles);
    const int kt0 = (int)(u - (int64_t)tile * ktiles);
    const int kt1 = (int)min((int64_t)ktiles, kt0 + (u_end - u));
    int tile_m, tile_n;
    gemm_tile_coords(g, tile, tile_m, tile_n);
    __syncthreads();                                        // the previous segment's LDS reads are done
    gemm_segment<64, 64, A_KC, B_KC, PF, KS>(g, lds, tile_m * 64, tile_n * 64, kt0 * BK, min(g.K, kt1 * BK), true, false);
    u += kt1 - kt0;
  }
}

// Bias + activation pass over C for products that ran split-K (the atomics carry the plain sums).
__global__ __launch_bounds__(256) void gemm_epilogue_kernel(GemmArgs g) {
  int64_t total = (int64_t)g.M * g.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int row = (int)(i / g.N), col = (int)(i - (int64_t)row * g.N);
    float* cp = g.C + (int64_t)row * g.ldc + col;
    float v = *cp + (g.bias ? g.bias[col] : 0.f);
    if (g.gat) v += gemm_gather2(g, row, col);
    if (g.epi == EPI_LEAKY_DROPOUT) {
      float keep;
      if (g.keep) keep = g.keep[(int64_t)row * g.ld_keep + col] ? 1.f : 0.f;
      else keep = (dpc_draw_u32(g.rng, g.site, (uint64_t)row * g.N + col) >> 31) ? 1.f : 0.f;
      float slope = (v > 0.f ? 1.f : g.leak) * keep * 2.f;
      g.aux_out[(int64_t)row * g.ld_aux_out + col] = slope;
      v *= slope;
    } else if (g.epi == EPI_MUL) {
      v *= g.aux_in[(int64_t)row * g.ld_aux_in + col];
    } else if (g.epi == EPI_RELU) {
      v = fmaxf(v, 0.f);
    }
    *cp = v;
  }
}

static int g_force_tile = 0;
static int g_deterministic = 0;
// Reproducible mode: never split K automatically (split sums are fp32 atomics, whose arrival order
// changes the last bits from run to run).  Host-side state.
extern "C" int dpc_set_deterministic(int on) {
  g_deterministic = on ? 1 : 0;
  return DPC_OK;
}

// Tuning / test hook: force the block tile: 0 = automatic, 64 = 64x64, 128 = 128x128,
// 12864 = 128x64, 64128 = 64x128.
extern "C" int dpc_gemm_config(int force_tile) {
  if (force_tile != 0 && force_tile != 64 && force_tile != 128 && force_tile != 12864 && force_tile != 64128)
    return DPC_EINVAL;
  g_force_tile = force_tile;
  return DPC_OK;
}

static int g_prefetch = 0;
// Tuning hook: depth of the register prefetch ring of the 64x64 tile: 0 = automatic, 2 or 4.  Host-side state.
extern "C" int dpc_gemm_prefetch(int depth) {
  if (depth != 0 && depth != 2 && depth != 4) return DPC_EINVAL;
  g_prefetch = depth;
  return DPC_OK;
}

template <int BM, int BN, int PF>
static void launch_tile(const GemmArgs& g, bool a_kc, bool b_kc, hipStream_t st) {
  dim3 grid(g.tiles_m * g.tiles_n * g.splitk), block(256);
  if (g.kscale) {     // validated by the caller: A is m-contiguous
    if (b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, PF, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, PF, true>), grid, block, 0, st, g);
    return;
  }
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, PF>), grid, block, 0, st, g);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false, PF>), grid, block, 0, st, g);
  else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, PF>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, PF>), grid, block, 0, st, g);
}

template <int PF>
static void launch_streamk(const GemmArgs& g, bool a_kc, bool b_kc, int nblocks, hipStream_t st) {
  dim3 grid(nblocks), block(256);
  if (g.kscale) {
    if (b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<false, true, PF, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_streamk_kernel<false, false, PF, true>), grid, block, 0, st, g);
    return;
  }
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<true, true, PF>), grid, block, 0, st, g);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<true, false, PF>), grid, block, 0, st, g);
  else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_streamk_kernel<false, true, PF>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_streamk_kernel<false, false, PF>), grid, block, 0, st, g);
}

// Stream-K is an option, not the default: on the step's shapes it is ~4 % faster than the XCD-aware
// split-K below (perfect balance) but fetches 2.7x the bytes from beyond L2 - its runs put different
// K ranges of a panel on an XCD at the same time (G dWout: 268 MB against 59 MB, 41 MB algorithmic).
static int g_streamk_blocks = -1;    // -1 = never (split-K), 0 = automatic, n = use n blocks whenever allowed
extern "C" int dpc_gemm_streamk_config(int blocks) {
  if (blocks < -1 || blocks > 4096) return DPC_EINVAL;
  g_streamk_blocks = blocks;
  return DPC_OK;
}

static int gemm_impl(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                     const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                     const float* bias, int epi, float* aux_out, int64_t ld_aux_out,
                     const float* aux_in, int64_t ld_aux_in, const uint8_t* keep, int64_t ld_keep,
                     const uint64_t* rng, uint32_t site, float leak, int splitk, const float* kscale,
                     const float* gat, int64_t ldg, const int32_t* gidx, void* stream) {
  DPC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  if (M == 0 || N == 0) return DPC_OK;
  DPC_CHECK_ARG(!gat || (gidx && ldg >= N && beta == 0.f), "gemm: row gather needs ids, ldg >= N and beta == 0");
  DPC_CHECK_ARG(A && B && C, "gemm: null operand");
  DPC_CHECK_ARG(a_rs == 1 || a_cs == 1, "gemm: A needs one unit stride");
  DPC_CHECK_ARG(b_rs == 1 || b_cs == 1, "gemm: B needs one unit stride");
  DPC_CHECK_ARG(ldc >= N, "gemm: ldc < N");
  DPC_CHECK_ARG(epi >= EPI_NONE && epi <= EPI_RELU, "gemm: unknown epilogue");
  DPC_CHECK_ARG(epi != EPI_LEAKY_DROPOUT || (aux_out && (keep || rng)), "gemm: dropout epilogue needs slope output and a mask source");
  DPC_CHECK_ARG(epi != EPI_MUL || aux_in, "gemm: multiply epilogue needs aux_in");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g;
  g.A = A; g.a_rs = a_rs; g.a_cs = a_cs;
  g.B = B; g.b_rs = b_rs; g.b_cs = b_cs;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.beta = beta; g.bias = bias; g.epi = epi;
  g.aux_out = aux_out; g.ld_aux_out = ld_aux_out; g.aux_in = aux_in; g.ld_aux_in = ld_aux_in;
  g.keep = keep; g.ld_keep = ld_keep; g.rng = rng; g.site = site; g.leak = leak;
  // k-contiguous when the k stride is 1 (ties, e.g. K == 1, go to the k-contiguous path)
  bool a_kc = (a_cs == 1), b_kc = (b_rs == 1);
  DPC_CHECK_ARG(!kscale || !a_kc, "gemm: kscale needs an m-contiguous A (the weight-gradient product)");
  g.kscale = kscale;
  g.gat = gat; g.ldg = ldg; g.gidx = gidx;
  int64_t lda = a_kc ? a_rs : a_cs, ldb = b_kc ? b_cs : b_rs;
  g.a_vec = (((uintptr_t)A & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
  g.b_vec = (((uintptr_t)B & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
  // Tile choice (measured per shape with `tools/gemm_bench.py tiles`): the 128x128 tile reaches
  // 120-130 TF/s (~80 % of the fp32 MFMA peak) once the grid holds >= 2 blocks per CU, but every
  // GEMM of the T1M training step (M*N <= 8192*512) is faster on 64x64 tiles, whose 4x as many
  // blocks fill the 256 CUs; 128x64 / 64x128 never won there and are kept for forced use only.
  int bm = 64, bn = 64;
  if (g_force_tile) {
    bm = (g_force_tile == 128 || g_force_tile == 12864) ? 128 : 64;
    bn = (g_force_tile == 128 || g_force_tile == 64128) ? 128 : 64;
  } else if (dpc_cdiv(M, 128) * dpc_cdiv(N, 128) >= 512) {
    bm = 128; bn = 128;
  }
  g.tiles_m = (int)dpc_cdiv(M, bm);
  g.tiles_n = (int)dpc_cdiv(N, bn);
  int64_t blocks = (int64_t)g.tiles_m * g.tiles_n;
  bool deferred_epi = false;
  int streamk = 0;                  // > 0: number of blocks of a stream-K launch
  if (splitk <= 0) {
    // auto.  Cost model fitted to MI355X measurements of the 64x64 tile (tools/gemm_occupancy.py,
    // tools/gemm_split_sweep.py): a CU works through its resident blocks at 17.5 ns per k and block
    // (21 ns when it holds only one: a single wave per SIMD hides less latency).  One block per tile
    // spreads `blocks` tiles over 256 CUs - 270 tiles take as long as 512 - while stream-K cuts
    // tiles x K into equal runs and pays fp32 atomics (~2.5 TB/s, mostly hidden behind other
    // blocks' MFMAs), a memset, and a separate pass for a bias / activation epilogue.
    splitk = 1;
    const bool can_split = !g_deterministic && (beta == 0.f || (beta == 1.f && epi == EPI_NONE && !bias)) && bm == 64 && bn == 64 && K >= 512;
    if (can_split && g_streamk_blocks >= 0) {
      const bool deferred = (epi != EPI_NONE || bias || gat);
      const double nb = (double)blocks;
      const double c = nb <= 256.0 ? 1.0 : (nb <= 768.0 ? (double)dpc_cdiv(blocks, 256) : nb / 256.0 + 0.3);
      const double t_plain = c * ((double)K * (nb <= 256.0 ? 0.021 : 0.0175) + 1.0) + 6.0;
      const double kblocks = nb * (double)dpc_cdiv(K, BK) * BK;          // k per block summed over all tiles
      // measured (tools/gemm_streamk_sweep.py): few tiles -> one block per CU, else 2-3 per CU
      const int sk_blocks = blocks < 64 ? 256 : (kblocks >= 768.0 * 256.0 ? 768 : (kblocks >= 512.0 * 128.0 ? 512 : 256));
      const double t_sk = kblocks / 256.0 * (sk_blocks > 256 ? 0.0175 : 0.021) + 3.0 + 6.0 +
                          (double)M * N * 4.0 * (1.0 + sk_blocks / nb) / 2.5e6 + (beta == 0.f ? 4.0 : 0.0) + (deferred ? 6.0 : 0.0);
      if (g_streamk_blocks > 0) streamk = g_streamk_blocks;
      else if (t_sk < 0.95 * t_plain) streamk = sk_blocks;
    } else if (can_split) {
      // split-K: tiles x sp blocks, each tile's K range cut into sp equal chunks
      const bool deferred = (epi != EPI_NONE || bias || gat);
      double best = 1e30;
      int64_t maxs = K / 256;
      if (maxs > 32) maxs = 32;
      for (int64_t sp = 1; sp <= maxs; ++sp) {
        const double nb = (double)(blocks * sp);
        const double c = nb <= 256.0 ? 1.0 : (nb <= 768.0 ? (double)dpc_cdiv((int64_t)nb, 256) : nb / 256.0 + 0.3);
        const double rate = nb <= 256.0 ? 0.021 : 0.0175;
        double t = c * ((double)K / sp * rate + 1.0) + 6.0;
        if (sp > 1) t += (double)M * N * 4.0 * sp / 2.5e6 + (beta == 0.f ? 4.0 : 0.0) + (deferred ? 6.0 : 0.0);
        if (t < best * 0.97) { best = t; splitk = (int)sp; }    // prefer the smaller split on near ties
      }
    }
  }
  if (splitk > 1 && (epi != EPI_NONE || bias || gat)) {
    DPC_CHECK_ARG(beta == 0.f, "gemm: split-K with an epilogue needs beta == 0");
    deferred_epi = true;
  }
  if (streamk > 0 && (epi != EPI_NONE || bias || gat)) deferred_epi = true;
  DPC_CHECK_ARG(splitk == 1 || beta == 0.f || beta == 1.f, "gemm: split-K needs beta in {0,1}");
  g.splitk = splitk;
  int ktiles = (int)dpc_cdiv(K, BK);
  g.kchunk = (int)dpc_cdiv(ktiles, splitk) * BK;
  if (K == 0) g.kchunk = BK;
  if ((splitk > 1 || streamk > 0) && beta == 0.f) {
    hipError_t e = hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, st);
    if (e != hipSuccess) { dpc_set_error(hipGetErrorString(e)); return DPC_EHIP; }
    g.beta = 1.f;   // accumulate into the zeroed C
  }
  GemmArgs full = g;
  if (deferred_epi) { g.epi = EPI_NONE; g.bias = nullptr; g.gat = nullptr; }
  if (streamk > 0) {
    int64_t units = blocks * ktiles;
    if (streamk > units) streamk = (int)units;
    launch_streamk<2>(g, a_kc, b_kc, streamk, st);
  }
  else if (bm == 128 && bn == 128) launch_tile<128, 128, 2>(g, a_kc, b_kc, st);
  else if (bm == 128) launch_tile<128, 64, 2>(g, a_kc, b_kc, st);
  else if (bn == 128) launch_tile<64, 128, 2>(g, a_kc, b_kc, st);
  else if (g_prefetch == 4) launch_tile<64, 64, 4>(g, a_kc, b_kc, st);
  else launch_tile<64, 64, 2>(g, a_kc, b_kc, st);
  if (deferred_epi)
    hipLaunchKernelGGL(gemm_epilogue_kernel, dim3(dpc_grid((int64_t)M * N, 256)), dim3(256), 0, st, full);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

extern "C" int dpc_gemm_f32(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                            const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                            const float* bias, int epi, float* aux_out, int64_t ld_aux_out,
                            const float* aux_in, int64_t ld_aux_in, const uint8_t* keep, int64_t ld_keep,
                            const uint64_t* rng, uint32_t site, float leak, int splitk, const float* kscale,
                            void* stream) {
  return gemm_impl(M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, beta, C, ldc, bias, epi, aux_out, ld_aux_out, aux_in,
                   ld_aux_in, keep, ld_keep, rng, site, leak, splitk, kscale, nullptr, 0, nullptr, stream);
}

// Linear over an input whose trailing columns are a one-/two-hot conditional vector (the generator's
// input cat([z, cond]) and the Residual blocks' cat([out, input]), dp_cgan.py:83,606): A and B cover the
// dense columns only, and the hot columns' share of x W^T - the sum of those columns of W - is added
// from gat = W[:, cond columns]^T [C, ldg] by row ids gidx int32 [M][2] (negative = none).
extern "C" int dpc_gemm_f32_gather(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                                   const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc, const float* bias,
                                   int epi, const float* gat, int64_t ldg, const int32_t* gidx, void* stream) {
  DPC_CHECK_ARG(gat && gidx, "gemm_gather: null pointer");
  DPC_CHECK_ARG(epi == EPI_NONE || epi == EPI_RELU, "gemm_gather: epilogue must be none or ReLU");
  return gemm_impl(M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, 0.f, C, ldc, bias, epi, nullptr, 0, nullptr, 0, nullptr, 0,
                   nullptr, 0, 0.2f, 0, nullptr, gat, ldg, gidx, stream);
}
