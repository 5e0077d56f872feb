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
// the global loads of tile t+PF issued before the MFMAs of tile t (one barrier per K-tile).  Split-K
// with fp32 atomics for the weight-gradient shapes (small MxN, K = batch).  Block ids
// are remapped so that the blocks sharing an A row-panel run on one XCD (private L2).
#include "dpc_common.h"
#include "../../include/dpcgan_hip.h"
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector type for the staging registers: HIP's float4 is a struct whose copies lower to
// llvm.memcpy between address spaces, which kept the prefetch ring in scratch memory
typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same with 4-byte alignment for global loads: gfx950 serves a dword-aligned global_load_dwordx4 (hipcc
// emits one for this type), which is what lets a segment-mapped operand (rows that start at odd offsets,
// see GemmArgs::seg_*) keep the 16-byte staging path
typedef f32x4 f32x4u __attribute__((aligned(4)));

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
  // Segment map: index x of a "compact" axis made of segments of seg_pad entries addresses element
  // (x / seg_pad) * seg_stride + x % seg_pad of the stored axis, of which only the first seg_len entries of a
  // segment are meaningful.  It lets a product run over the DATA columns of the discriminator's pac-row
  // layout [pac][Dd data | C cond] (seg_stride = Dd + C, dp_cgan.py:66-68,622-623) without a copy of W1:
  //   seg_b 1: along K of a k-contiguous B (first layer forward x W1^T; seg_pad % 32 == 0, so a K tile never
  //            straddles segments);  seg_b 2: along N of an n-contiguous B (input gradient dy W1);
  //   seg_c 1: along N of C (weight gradient dy^T x into W1's layout; entries beyond seg_len are not stored).
  int seg_b, seg_c, seg_len, seg_pad; int64_t seg_stride;
  // optional column statistics of the result (before beta * C): colstat[n] += sum_m v, colstat[N + n] += sum_m v^2
  // with fp64 atomics - BatchNorm1d's batch statistics taken where the Linear's output tile still sits in
  // registers (dp_cgan.py:75-76).  fp32 sums over the 32 rows of an MFMA tile, fp64 across tiles.
  double* colstat;
  // BatchNorm-backward form of the statistics (bn_y != NULL): over the first stat_n columns of the FINAL result
  // dy (after beta * C), colstat[n] += sum_m dyh, colstat[stat_n + n] += sum_m dyh * xhat with dyh = dy where the
  // block's ReLU output bn_y is positive (else 0) and xhat = (bn_z - mean[n]) * invstd[n]
  const float* bn_y; int64_t ld_bn_y; const float* bn_z; int64_t ld_bn_z; const float* bn_mean; const float* bn_invstd;
  int stat_n;
  int stat_slots;        // the accumulators are [stat_slots][2][stat_n]: tile row t adds into slot t % stat_slots (fewer atomics per address)
  int force_atomic;      // add the result to C with fp32 atomics even without split-K (another product adds to C meanwhile)
  // EPI_LEAKY_DROPOUT only, optional: seed_out[m, n] = (seed_row[m] * seed_col[n]) * slope[m, n] - the seed of the
  // backward chain of the discriminator's LAST hidden layer (delta = rowscale * w_head * slope: what the head kernel
  // writes otherwise), so that the chain's first product follows this one directly
  float* seed_out; int64_t ld_seed; const float* seed_row; const float* seed_col;
};

// stored index minus compact index
__device__ __forceinline__ int64_t gemm_seg_delta(const GemmArgs& g, int x) {
  return (int64_t)(x / g.seg_pad) * (g.seg_stride - g.seg_pad);
}

__device__ __forceinline__ float gemm_gather2(const GemmArgs& g, int row, int col) {
  const int i0 = g.gidx[2 * row], i1 = g.gidx[2 * row + 1];
  float v = 0.f;
  if (i0 >= 0) v += g.gat[(int64_t)i0 * g.ldg + col];
  if (i1 >= 0) v += g.gat[(int64_t)i1 * g.ldg + col];
  return v;
}

constexpr int BK = 32;

// Measurement build only (tools/gemm_stamp.sh compiles this file with -DDPC_GEMM_STAMP into a library of its own; the
// product library never carries it): thread 0 of every block of gemm_f32_kernel leaves the shader-cycle counter
// at five points of the block's life - and the 100 MHz wall clock at its two ends - in a caller-provided buffer.
#ifdef DPC_GEMM_STAMP
__device__ unsigned long long* dpc_stamp_buf = nullptr;
#define DPC_STAMP(i)                                                                                 \
  do {                                                                                               \
    if (threadIdx.x == 0 && dpc_stamp_buf) dpc_stamp_buf[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#define DPC_STAMP_WALL(i)                                                                            \
  do {                                                                                               \
    if (threadIdx.x == 0 && dpc_stamp_buf) dpc_stamp_buf[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
  } while (0)
extern "C" int dpc_debug_stamp_buffer(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(dpc_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
// DPC_EXP=<bits>: leave a class of instructions OUT of the K loop (results wrong; timing only): 1 global loads,
// 2 LDS stores, 4 fragment reads, 8 the barrier
#ifdef DPC_EXP
#define DPC_EXP_IF(k) if (!((DPC_EXP >> ((k) - 1)) & 1))
#else
#define DPC_EXP_IF(k)
#endif
#else
#define DPC_EXP_IF(k)
#define DPC_STAMP(i) do {} while (0)
#define DPC_STAMP_WALL(i) do {} while (0)
#endif

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
        r[p] = *reinterpret_cast<const f32x4u*>(base + (int64_t)row * ld + k0 + 4 * (t & 7));
      } else {
        constexpr int TPR = BMN / 4;
        constexpr int RPP = 256 / TPR;
        int mn = min(mn0 + 4 * (t % TPR), ((MN + 3) & ~3) - 4);
        r[p] = *reinterpret_cast<const f32x4u*>(base + (int64_t)(k0 + p * RPP + t / TPR) * ld + mn);
        if (KS) sc[p] = ks[k0 + p * RPP + t / TPR];
      }
    }
  }

  // The K loop's form of load_fast: the per-thread part of every address is loop-invariant - `off` (elements, fits 32
  // bits: checked on the host) from offsets() below - and only the block-uniform tile base moves from tile to tile,
  // so each load is one instruction with a scalar base and a 32-bit vector offset and the loop carries no 64-bit
  // vector address arithmetic (sixteen v_lshl_add_u64 per K-tile before: ~240 of a lone wave's 1450 cycles).
  //   tile_base: base + k0 (k-contiguous operand) or base + k0 * ld (m/n-contiguous);  ks_tile: ks + k0
  __device__ __forceinline__ void offsets(int64_t ld, int mn0, int MN, uint32_t off[PASSES], uint32_t koff[PASSES]) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      if (KC) {
        const int row = min(mn0 + p * 32 + (t >> 3), MN - 1);
        off[p] = (uint32_t)((int64_t)row * ld + 4 * (t & 7));
        koff[p] = 0;
      } else {
        constexpr int TPR = BMN / 4;
        constexpr int RPP = 256 / TPR;
        const int mn = min(mn0 + 4 * (t % TPR), ((MN + 3) & ~3) - 4);
        off[p] = (uint32_t)((int64_t)(p * RPP + t / TPR) * ld + mn);
        koff[p] = (uint32_t)(p * RPP + t / TPR);
      }
    }
  }
  __device__ __forceinline__ void load_tile(const float* __restrict__ tile_base, const uint32_t off[PASSES],
                                            const float* __restrict__ ks_tile, const uint32_t koff[PASSES]) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      r[p] = *reinterpret_cast<const f32x4u*>(tile_base + off[p]);
      if (KS) sc[p] = ks_tile[koff[p]];
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
          // (masked with AND, not `cond ? v : 0`: hipcc turns a select whose operand is a load into a branch around
          // the load and drains vmcnt at the join - sixteen dependent round trips for one partial tile, ~5 us)
          const float v = src[min(k + j, klast)];
          const uint32_t keep = (row <= mlast && k + j <= klast) ? 0xffffffffu : 0u;
          e[j] = __uint_as_float(__float_as_uint(v) & keep);
        }
      } else {
        constexpr int TPR = BMN / 4;            // threads per k-row
        constexpr int RPP = 256 / TPR;          // k-rows per pass
        int mn = mn0 + 4 * (t % TPR);
        int k = k0 + p * RPP + t / TPR;
        const float* src = base + (int64_t)min(k, klast) * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = src[min(mn + j, mlast)];
          const uint32_t keep = (k <= klast && mn + j <= mlast) ? 0xffffffffu : 0u;
          e[j] = __uint_as_float(__float_as_uint(v) & keep);
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

  // Accumulator chains: ONE per 32x32 tile.  (A second chain for the even / odd k-steps of a wave that owns a single
  // tile was measured - in-kernel stamps, lone block and full grids - and changed nothing: 1661 against 1656 cycles
  // per K-tile.  The SQ counters show MFMA-busy at exactly 64 cycles per instruction but ~97 wave-cycles between two
  // MFMAs of a lone wave whatever their dependence; what closes the gap is more waves per SIMD - 84 % of the MFMA rate
  // at two, 98 % at three - so registers are better spent on occupancy than on a second chain.)
  constexpr int CH = 1;
  f32x16 acc[TM][TN], acc2[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

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
  // segment-mapped B: a per-thread offset when the map runs along N (constant over the K loop), a per-K-tile
  // one (block-uniform: a tile lies inside one segment) when it runs along K
  const float* Bn = g.B;
  if (!B_KC && g.seg_b == 2) Bn += gemm_seg_delta(g, min(n0 + 4 * ((int)threadIdx.x % (BN / 4)), ((g.N + 3) & ~3) - 4));
  auto Bk = [&](int k0) -> const float* { return (B_KC && g.seg_b == 1) ? Bn + gemm_seg_delta(g, k0) : Bn; };
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
            if (CH == 2 && (s & 1))
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc2[i][j], 0, 0, 0);
            else
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
  // (only the forward form x W^T, both operands k-contiguous, ever carries a gather: the other variants do not pay the registers)
  constexpr bool GPRE = (BM * BN <= 4096) && A_KC && B_KC;
  constexpr int GQ = GPRE ? (BM * GV) / 256 : 1;
  const bool gvec = gather && ((g.ldg & 3) == 0) && (((uintptr_t)g.gat & 15) == 0) && (n0 + BN <= g.N);
  // Branch-free: all ids first, then all rows (an absent id reads row 0 and is dropped by a select after the
  // K loop).  With `if (id >= 0) v += load` hipcc branched around every load and drained the load queue at each
  // join: 3 x GQ dependent L2 round trips in front of the K loop of every gather product.
  f32x4 gw0[GQ], gw1[GQ];
  int gi0[GQ], gi1[GQ];
  if (GPRE && gvec) {
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
      const int idx = threadIdx.x + q * 256;
      const int row = min(m0 + idx / GV, g.M - 1);
      gi0[q] = g.gidx[2 * row];
      gi1[q] = g.gidx[2 * row + 1];
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
      const int idx = threadIdx.x + q * 256;
      const int r = idx / GV, c4 = (idx - r * GV) * 4;
      gw0[q] = *reinterpret_cast<const f32x4*>(g.gat + (int64_t)max(gi0[q], 0) * g.ldg + n0 + c4);
      gw1[q] = *reinterpret_cast<const f32x4*>(g.gat + (int64_t)max(gi1[q], 0) * g.ldg + n0 + c4);
    }
  }

  // Main pipeline over the nfull full K-tiles.  The MFMA chain of a wave never waits for LDS or for the barrier:
  //   * all FOUR fragment groups of a K-tile (8 k each) live in registers (fr[0..3]: 8 VGPRs per group and 32 x 32
  //     MFMA tile pair), read from LDS one half-iteration before their MFMAs issue;
  //   * the ONE barrier of an iteration sits in its middle.  Iteration t: global loads of tile t+2 -> registers;
  //     tile t+1 (loaded an iteration ago) registers -> LDS buffer (t+1)&1; MFMAs of groups 0, 1; barrier (now tile
  //     t+1 is visible and every wave holds all of tile t in registers); fragment groups 0, 1 of tile t+1 -> the
  //     registers groups 0, 1 just vacated; MFMAs of groups 2, 3; fragment groups 2, 3 of tile t+1.
  //   Buffer (t+1)&1 last held tile t-1, whose fragment reads every wave finished before the barrier of
  //   iteration t-1 - so the store at the top of iteration t is safe with two buffers.
  // (Before: fragments were read right after the barrier, group by group, and each K-tile paid one exposed LDS
  // round trip behind the barrier and another between groups - 0.61 us per K-tile for 0.49 us of MFMA issue.)
  // Strictly branch-free inside (loads past the end are clamped to the last tile, the stores and fragment reads
  // of the last iteration are redundant, not guarded), so hipcc keeps counted s_waitcnt vmcnt(N).
  struct Frag { float a[TM][4]; float b[TN][4]; };
  Frag fr[4];
  auto read_group = [&](const int gi, const float* as, const float* bs) {
#pragma unroll
    for (int i = 0; i < TM; ++i) SA::frag(as, wm0 + i * 32, gi * 8, li, lk, fr[gi].a[i]);
#pragma unroll
    for (int j = 0; j < TN; ++j) SB::frag(bs, wn0 + j * 32, gi * 8, li, lk, fr[gi].b[j]);
  };
  auto mfma_group = [&](const int gi) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (CH == 2 && (s & 1))
            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[gi].a[i][s], fr[gi].b[j][s], acc2[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[gi].a[i][s], fr[gi].b[j][s], acc[i][j], 0, 0, 0);
  };
  if (nfull > 0) {
    const int last = nfull - 1;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      sa[j].load_fast(g.A, lda, m0, k_begin + min(j, last) * BK, g.M, g.kscale);
      sb[j].load_fast(Bk(k_begin + min(j, last) * BK), ldb, n0, k_begin + min(j, last) * BK, g.N);
    }
    sa[0].store(As0);
    sb[0].store(Bs0);
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) read_group(gi, As0, Bs0);      // (group 3 at the top of the tile's own iteration)
    DPC_STAMP(1);                                   // first tile in LDS, its fragments requested
    static_assert(PF == 2, "the pipelined loop keeps two register stages (tile t+1 awaiting its LDS store, tile t+2 in flight)");
    // The tile the next iteration loads, and - for a B whose K axis is segment-mapped (Bk above) - its offset, kept
    // as running values: the division inside Bk() was a block of its own in the middle of every iteration.
    int tl = min(2, last);
    const bool seg_k = B_KC && g.seg_b == 1;
    const int seg_pad_k = seg_k ? g.seg_pad : 0x40000000;
    const int64_t seg_jump = seg_k ? g.seg_stride - g.seg_pad : 0;
    int tl_kin = seg_k ? (k_begin + tl * BK) % g.seg_pad : 0;
    // block-uniform tile bases (scalar registers) + loop-invariant per-thread offsets: see Stage::load_tile
    const int64_t a_step = A_KC ? BK : (int64_t)BK * lda, b_step = B_KC ? BK : (int64_t)BK * ldb;
    const float* a_tile = g.A + (int64_t)(k_begin + tl * BK) * (A_KC ? 1 : lda);
    const float* b_tile = g.B + (seg_k ? gemm_seg_delta(g, k_begin + tl * BK) : 0) + (int64_t)(k_begin + tl * BK) * (B_KC ? 1 : ldb);
    const float* ks_tile = KS ? g.kscale + k_begin + tl * BK : nullptr;
    uint32_t a_off[SA::PASSES], a_koff[SA::PASSES], b_off[SB::PASSES], b_koff[SB::PASSES];
    sa[0].offsets(lda, m0, g.M, a_off, a_koff);
    sb[0].offsets(ldb, n0, g.N, b_off, b_koff);
    if (!B_KC && g.seg_b == 2) {        // an N segment map shifts each thread's columns: part of its invariant offset
      const uint32_t dn = (uint32_t)(Bn - g.B);
#pragma unroll
      for (int p = 0; p < SB::PASSES; ++p) b_off[p] += dn;
    }
    // Issue order inside an iteration.  One wave per SIMD hides NOTHING behind its own MFMAs unless the other
    // instructions sit BETWEEN them: hipcc's own schedule put the sixteen MFMAs of a K-tile back to back and the
    // loads, LDS stores and fragment reads in clusters between the groups, and the SQ counters of a lone block
    // then show the issue time of those clusters added to the MFMA time (1024 MFMA-busy cycles + ~290 issuing other
    // instructions + ~180 parked = 1546 per K-tile).  The iteration is therefore written as four quarters, each the
    // four (sixteen) MFMAs of one fragment group with the memory instructions that may run beside them, fenced by
    // sched_barrier and interleaved by sched_group_barrier:
    //   Q0  MFMAs of group 0 | fragment group 3 of THIS tile <- LDS, global loads of tile t+2
    //   Q1  MFMAs of group 1 | tile t+1 registers -> LDS
    //   ---- barrier ----
    //   Q2  MFMAs of group 2 | fragment groups 0, 1 of tile t+1 <- LDS (into the registers Q0, Q1 vacated)
    //   Q3  MFMAs of group 3 | fragment group 2 of tile t+1 <- LDS
    // Every fragment group is read at least eight MFMAs (512 cycles) before its own MFMAs issue.
    constexpr int NM = 4 * TM * TN;                                   // MFMAs per quarter
    constexpr int N_VM = SA::PASSES * (KS ? 2 : 1) + SB::PASSES;      // global loads per K-tile (KS: + the k-scales)
    constexpr int N_DW = SA::PASSES + SB::PASSES;                     // LDS stores per K-tile
    constexpr int N_DRG = TM * (A_KC ? 1 : 4) + TN * (B_KC ? 1 : 4);  // LDS reads per fragment group
#define DPC_PIPE(MASK_FIRST, N_FIRST, MASK, N)                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < NM; ++i_) {                                                  \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
      if (i_ == 0 && (N_FIRST) > 0) __builtin_amdgcn_sched_group_barrier(MASK_FIRST, N_FIRST, 0);         \
      if (i_ * (((N) + NM - 1) / NM) < (N)) __builtin_amdgcn_sched_group_barrier(MASK, ((N) + NM - 1) / NM, 0); \
    }
    for (int t0 = 0; t0 + 1 < nfull; t0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int nxt = u ^ 1;                      // LDS buffer / register stage of tile t+1 (t & 1 == u)
        const float* an = As + nxt * SA::SIZE;
        const float* bn = Bs + nxt * SB::SIZE;
        // Q0
        DPC_EXP_IF(3) read_group(3, As + u * SA::SIZE, Bs + u * SB::SIZE);
        DPC_EXP_IF(1) {
        sa[u].load_tile(a_tile, a_off, ks_tile, a_koff);       // tile t+2 into the stage tile t left
        sb[u].load_tile(b_tile, b_off, nullptr, b_koff);
        }
        DPC_EXP_IF(5)
        {       // the tile after it (clamped to the last one; selects only: the iteration stays ONE basic block)
          const bool adv = tl < last;
          const bool wrap = tl_kin + BK >= seg_pad_k;
          tl += adv ? 1 : 0;
          a_tile += adv ? a_step : 0;
          b_tile += adv ? (wrap ? b_step + seg_jump : b_step) : 0;
          if (KS) ks_tile += adv ? BK : 0;
          tl_kin = adv ? (wrap ? 0 : tl_kin + BK) : tl_kin;
        }
        mfma_group(0);
        DPC_PIPE(0x100, N_DRG, 0x020, N_VM)
        __builtin_amdgcn_sched_barrier(0);
        // Q1
        DPC_EXP_IF(2) {
        sa[nxt].store(As + nxt * SA::SIZE);         // tile t+1 (its loads were issued an iteration ago)
        sb[nxt].store(Bs + nxt * SB::SIZE);
        }
        mfma_group(1);
        DPC_PIPE(0, 0, 0x200, N_DW)
        __builtin_amdgcn_sched_barrier(0);
        DPC_EXP_IF(4) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // Q2
        DPC_EXP_IF(3) {
        read_group(0, an, bn);
        read_group(1, an, bn);
        }
        mfma_group(2);
        DPC_PIPE(0, 0, 0x100, 2 * N_DRG)
        __builtin_amdgcn_sched_barrier(0);
        // Q3
        DPC_EXP_IF(3) read_group(2, an, bn);
        mfma_group(3);
        DPC_PIPE(0, 0, 0x100, N_DRG)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef DPC_PIPE
    if (nfull & 1) {                                // the last full tile: groups 0-2 sit in the fragment registers already
      read_group(3, As + ((nfull - 1) & 1) * SA::SIZE, Bs + ((nfull - 1) & 1) * SB::SIZE);
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) mfma_group(gi);
    }
    __syncthreads();                                // every wave is done with the LDS buffers (remainder / gather image next)
    DPC_STAMP(2);                                   // K loop done
  }
  // Remainder (a partial last tile, or operands the fast path cannot take): one tile at a time.
  for (int t = nfull; t < ntile; ++t) {
    sa[0].load(g.A, lda, m0, k_begin + t * BK, g.M, k_end, fa, g.kscale);
    sb[0].load(Bk(k_begin + t * BK), ldb, n0, k_begin + t * BK, g.N, k_end, fb);
    sa[0].store(As0);
    sb[0].store(Bs0);
    __syncthreads();
    compute(As0, Bs0);
    __syncthreads();
  }

  if (CH == 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += acc2[i][j];
  }
  if (gather) {
    static_assert(BM * BN <= 2 * SA::SIZE + 2 * SB::SIZE, "gather image must fit the operand buffers");
    if (GPRE && gvec) {
#pragma unroll
      for (int q = 0; q < GQ; ++q) {
        const int idx = threadIdx.x + q * 256;
        const int r = idx / GV, c4 = (idx - r * GV) * 4;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&lds[r * BN + c4]) = (gi0[q] >= 0 ? gw0[q] : zero4) + (gi1[q] >= 0 ? gw1[q] : zero4);
      }
    } else if (gvec) {
      // (128 x 128 tiles: after the K loop, four 16-byte groups per thread at a time, branch-free as above)
      static_assert((BM * GV) % 1024 == 0 || BM * GV <= 1024, "gather groups");
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      for (int base = 0; base < BM * GV; base += 1024) {
        int i0[4], i1[4];
        f32x4 w0[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 256 + (int)threadIdx.x;
          const int row = min(m0 + idx / GV, g.M - 1);
          i0[u] = g.gidx[2 * row];
          i1[u] = g.gidx[2 * row + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 256 + (int)threadIdx.x;
          const int r = idx / GV, c4 = (idx - r * GV) * 4;
          w0[u] = *reinterpret_cast<const f32x4*>(g.gat + (int64_t)max(i0[u], 0) * g.ldg + n0 + c4);
          w1[u] = *reinterpret_cast<const f32x4*>(g.gat + (int64_t)max(i1[u], 0) * g.ldg + n0 + c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * 256 + (int)threadIdx.x;
          const int r = idx / GV, c4 = (idx - r * GV) * 4;
          if (idx < BM * GV)       // (rows beyond M hold the last row's values: the epilogue never reads them)
            *reinterpret_cast<f32x4*>(&lds[r * BN + c4]) = (i0[u] >= 0 ? w0[u] : zero4) + (i1[u] >= 0 ? w1[u] : zero4);
        }
      }
    } else {
      // edge tiles (columns beyond N) and unaligned tables: element by element
      for (int idx = threadIdx.x; idx < BM * GV; idx += 256) {
        const int r = idx / GV, c4 = (idx - r * GV) * 4;
        const int row = m0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < g.M) {
          const int i0 = g.gidx[2 * row], i1 = g.gidx[2 * row + 1];
          float e4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int col = min(n0 + c4 + e, g.N - 1);
            const float x0 = g.gat[(int64_t)max(i0, 0) * g.ldg + col], x1 = g.gat[(int64_t)max(i1, 0) * g.ldg + col];
            e4[e] = (i0 >= 0 ? x0 : 0.f) + (i1 >= 0 ? x1 : 0.f);
          }
          v = f32x4{e4[0], e4[1], e4[2], e4[3]};
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
    bool live = col < g.N;
    int64_t ocol = col;                    // column of C in memory
    if (g.seg_c) {
      const int sgm = col / g.seg_pad;
      live = live && (col - sgm * g.seg_pad) < g.seg_len;
      ocol = col + (int64_t)sgm * (g.seg_stride - g.seg_pad);
    }
    float cs0 = 0.f, cs1 = 0.f;
    const bool bnb = g.bn_y != nullptr && col < g.stat_n;
    if (live) {
      const float bv = (g.bias && add_bias) ? g.bias[col] : 0.f;
      if (atomic) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
          if (row < g.M) atomicAdd(g.C + (int64_t)row * g.ldc + ocol, g.alpha * av[e] + bv);
        }
      } else {
        // Straight-line phases over eight elements at a time, every choice between phases block-uniform and the ONLY
        // per-lane guard the one around each store.  (The first form - one `if (row < M) { ... }` body per element
        // holding loads, the activation switch and the store - came out of hipcc as sixteen blocks that each begin
        // with s_waitcnt vmcnt(0): every element waited for the previous element's STORE to be acknowledged, 600-700
        // cycles each.  The in-kernel stamps of tools/gemm_stamp.py showed an 11 000-cycle epilogue, 5 us of every
        // product, for sixteen stores per lane.)
        // What the epilogue READS per element (old C for beta, the multiplier, a stored keep mask, the BatchNorm
        // block's y and z) is loaded first, from row- and column-clamped addresses.
        const float bmu = bnb ? g.bn_mean[col] : 0.f, bis = bnb ? g.bn_invstd[col] : 0.f;
        const bool ld_c = g.beta != 0.f, ld_aux = g.epi == EPI_MUL, ld_keep = g.epi == EPI_LEAKY_DROPOUT && g.keep != nullptr;
        const bool ld_bn = g.bn_y != nullptr;       // (block-uniform; lanes beyond stat_n read a clamped column and drop it)
        const int bcol = min(col, g.stat_n - 1);
        const bool fwd_stat = g.colstat != nullptr && !ld_bn;
        // (eight elements at a time: all sixteen kept the kernel above 128 registers, three waves per SIMD instead of four)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float c_old[8], aux[8], by[8], bz[8], v[8];      // (each read only under the condition it was loaded under)
          uint8_t kp[8];
          bool ok[8];
          const int rbase = row0 + 16 * h + 4 * lk;        // element e8 of this half: row rbase + (e8 & 3) + 8 * (e8 >> 2)
          const int mlast = g.M - 1;
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) ok[e8] = rbase + (e8 & 3) + 8 * (e8 >> 2) < g.M;
          // ONE block-uniform branch per kind of input (a `cond ? load : 0` per element became a branch per element:
          // forty taken branches through cold code per half - the instruction fetch of a block's once-only epilogue
          // was most of its 5 000 cycles)
          if (ld_c) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) c_old[e8] = g.C[(int64_t)min(rbase + (e8 & 3) + 8 * (e8 >> 2), mlast) * g.ldc + ocol];
          }
          if (ld_aux) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) aux[e8] = g.aux_in[(int64_t)min(rbase + (e8 & 3) + 8 * (e8 >> 2), mlast) * g.ld_aux_in + col];
          }
          if (ld_keep) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) kp[e8] = g.keep[(int64_t)min(rbase + (e8 & 3) + 8 * (e8 >> 2), mlast) * g.ld_keep + col];
          }
          if (ld_bn) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const int64_t rc = min(rbase + (e8 & 3) + 8 * (e8 >> 2), mlast);
              by[e8] = g.bn_y[rc * g.ld_bn_y + bcol];
              bz[e8] = g.bn_z[rc * g.ld_bn_z + bcol];
            }
          }
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) v[e8] = g.alpha * av[8 * h + e8] + bv;
          if (h == 0) DPC_STAMP(9);
          if (gather) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const int e = 8 * h + e8;
              v[e8] += lds[(row0 - m0 + (e & 3) + 8 * (e >> 2) + 4 * lk) * BN + (col - n0)];
            }
          }
          if (fwd_stat) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const float m = ok[e8] ? v[e8] : 0.f;
              cs0 += m;
              cs1 += m * m;
            }
          }
          if (g.epi == EPI_LEAKY_DROPOUT) {
            float slope[8];
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const int e = 8 * h + e8;
              const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
              const bool keep = g.keep ? (kp[e8] != 0) : ((dpc_draw_u32(g.rng, g.site, (uint64_t)row * g.N + col) >> 31) != 0);
              slope[e8] = keep ? (v[e8] > 0.f ? 2.f : 2.f * g.leak) : 0.f;
              v[e8] *= slope[e8];
            }
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const int e = 8 * h + e8;
              const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
              if (ok[e8]) g.aux_out[(int64_t)row * g.ld_aux_out + col] = slope[e8];
            }
            if (g.seed_out) {
              const float wc = g.seed_col[col];
#pragma unroll
              for (int e8 = 0; e8 < 8; ++e8) {
                const int row = rbase + (e8 & 3) + 8 * (e8 >> 2);
                if (ok[e8]) g.seed_out[(int64_t)row * g.ld_seed + col] = (g.seed_row[min(row, mlast)] * wc) * slope[e8];
              }
            }
          } else if (g.epi == EPI_MUL) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) v[e8] *= aux[e8];
          } else if (g.epi == EPI_RELU) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) v[e8] = fmaxf(v[e8], 0.f);
          }
          if (ld_c) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) v[e8] += g.beta * c_old[e8];
          }
          if (h == 0) DPC_STAMP(10);
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) {
            const int e = 8 * h + e8;
            const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lk;
            if (ok[e8]) g.C[(int64_t)row * g.ldc + ocol] = v[e8];
          }
          if (h == 0) DPC_STAMP(11);
          if (bnb) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
              const float dyh = (ok[e8] && by[e8] > 0.f) ? v[e8] : 0.f;
              cs0 += dyh;
              cs1 += dyh * ((bz[e8] - bmu) * bis);
            }
          }
        }
      }
    }
    if (g.colstat) {             // (block-uniform; both lane halves hold the same column)
      cs0 += __shfl_xor(cs0, 32, 64);
      cs1 += __shfl_xor(cs1, 32, 64);
      if (lk == 0 && live && col < g.stat_n) {
        double* cs = g.colstat + (int64_t)((m0 / BM) % g.stat_slots) * 2 * g.stat_n;
        atomicAdd(cs + col, (double)cs0);
        atomicAdd(cs + g.stat_n + col, (double)cs1);
      }
    }
  };
  DPC_STAMP(7);                                     // accumulator chains summed, gather image staged
  tile_out(acc[0][0], m0 + wm0, n0 + wn0 + li);
  DPC_STAMP(8);                                     // first MFMA tile's epilogue issued
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
  // (selects, not two branches writing through the references: hipcc turned those into a store + load of the pair
  // through scratch memory at the top of every product - a dependent memory round trip before the first operand load)
  const bool m_inner = g.tiles_m < g.tiles_n;
  const int inner = m_inner ? g.tiles_m : g.tiles_n;
  const int q = wg / inner, r = wg - q * inner;
  tile_m = m_inner ? r : q;
  tile_n = m_inner ? q : r;
}

// Slot of block `bid` among `n` blocks such that the blocks of one XCD (ids congruent mod 8) hold a
// contiguous run of slots (bijective for any n).
__device__ __forceinline__ int gemm_xcd_slot(int bid, int n) {
  int q = n >> 3, r = n & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Data-parallel kernel: one block per tile (gridDim.y > 1: per K chunk of a split-K product).
template <int BM, int BN, bool A_KC, bool B_KC, int PF, bool KS = false>
// (4 blocks per CU for the 64x64 tile - LDS would allow it - caps the kernel at 128 VGPRs: the k-contiguous variant
// spills 68 bytes per lane and the step goes from 0.655 to 0.674 ms; round 3)
__global__ __launch_bounds__(256, (BM == 128 ? 2 : 3)) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Stage<BM, A_KC, KS>::SIZE + 2 * Stage<BN, B_KC>::SIZE];
  // 1-D grid of tiles x splitk blocks.  An XCD's contiguous run of slots is taken in split-major order
  // (slot = split * tiles + tile): with 8 or more K chunks an XCD works on ONE chunk of K for all
  // tiles, so each XCD's L2 fetches its slice of A and B once - the whole product then reads A and B
  // about once from HBM instead of once per tile row / column and XCD.
  DPC_STAMP(0);
  DPC_STAMP_WALL(5);
  const int tiles = g.tiles_m * g.tiles_n;
  const int slot = gemm_xcd_slot(blockIdx.x, tiles * g.splitk);
  const int split = slot / tiles;
  int tile_m, tile_n;
  gemm_tile_coords(g, slot - split * tiles, tile_m, tile_n);
  int k_begin = split * g.kchunk;
  int k_end = min(g.K, k_begin + g.kchunk);
  if (k_begin >= k_end && split > 0) return;
  gemm_segment<BM, BN, A_KC, B_KC, PF, KS>(g, lds, tile_m * BM, tile_n * BN, k_begin, k_end, g.splitk > 1 || g.force_atomic, split == 0);
  DPC_STAMP(3);                                     // epilogue issued
#ifdef DPC_GEMM_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  DPC_STAMP(4);                                     // ... and its stores acknowledged
  DPC_STAMP_WALL(6);
}

// Grouped launch: up to DPC_GROUP_MAX INDEPENDENT products of one kernel variant (same operand layouts, 64x64 tiles)
// as ONE grid - block b works on the product whose block range holds b.  The weight gradients of an update are such
// a set (dy^T x per layer, all m-/n-contiguous): one launch instead of three parallel graph branches pays the
// launch + prologue + epilogue ramp once and fills the CUs that a 128- or 176-block product leaves idle.
#define DPC_GROUP_MAX 4
struct GemmGroupArgs {
  int count;
  int start[DPC_GROUP_MAX + 1];      // block range of product i: [start[i], start[i+1])
  GemmArgs g[DPC_GROUP_MAX];
};

template <bool A_KC, bool B_KC, bool KS>
__global__ __launch_bounds__(256, 3) void gemm_group_kernel(GemmGroupArgs G) {
  __shared__ __attribute__((aligned(16))) float lds[2 * Stage<64, A_KC, KS>::SIZE + 2 * Stage<64, B_KC>::SIZE];
  int i = 0;
#pragma unroll
  for (int k = 1; k < DPC_GROUP_MAX; ++k) i += (k < G.count && (int)blockIdx.x >= G.start[k]) ? 1 : 0;
  const GemmArgs& g = G.g[i];
  const int local = (int)blockIdx.x - G.start[i];
  const int tiles = g.tiles_m * g.tiles_n;
  const int slot = gemm_xcd_slot(local, tiles * g.splitk);
  const int split = slot / tiles;
  int tile_m, tile_n;
  gemm_tile_coords(g, slot - split * tiles, tile_m, tile_n);
  const int k_begin = split * g.kchunk;
  const int k_end = min(g.K, k_begin + g.kchunk);
  if (k_begin >= k_end && split > 0) return;
  gemm_segment<64, 64, A_KC, B_KC, 2, KS>(g, lds, tile_m * 64, tile_n * 64, k_begin, k_end, g.splitk > 1 || g.force_atomic, split == 0);
}

// ---------------------------------------------------------------------------------------------
// Small products (few 64x64 tiles, short K: the K = 256 links of the discriminator's chains at batch 1024-3072)
// are bound by a block's serial walk over K, not by the matrix cores: with 64-192 blocks on 256 CUs a block
// pays a load round trip + 16 MFMAs per K-tile, tile after tile.  Here a block owns a 32 x 32 tile of C (4 x as
// many blocks) and its 4 waves SPLIT K: a group of 8 K-tiles (256 k) is loaded at once - every load of the
// group in flight together - wave w takes tiles w and w + 4 of the group, and the four partial accumulators
// meet in LDS, where 256 threads x 4 consecutive columns apply the epilogue with 16-byte accesses.
// K = 256 is one round trip and 32 MFMAs per wave instead of eight of each.
// ---------------------------------------------------------------------------------------------
// KS_GROUP: K-tiles per group, 8 (2 per wave, 72 KB of LDS: two blocks per CU) or 4 (36 KB: four per CU)
template <bool A_KC, bool B_KC, bool KS, int KS_GROUP>
__global__ __launch_bounds__(256) void gemm_ksplit_kernel(GemmArgs g) {
  using SA = Stage<32, A_KC, KS>;
  using SB = Stage<32, B_KC>;
  static_assert(SA::SIZE == SB::SIZE, "32 x 32 tiles: both images are 32 x 36 floats");
  constexpr int TS = SA::SIZE;
  __shared__ __attribute__((aligned(16))) float lds[2 * KS_GROUP * TS];        // A tiles, then B tiles (72 KB)
  float* As = lds;
  float* Bs = lds + KS_GROUP * TS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  int tile_m, tile_n;
  gemm_tile_coords(g, gemm_xcd_slot(blockIdx.x, g.tiles_m * g.tiles_n), tile_m, tile_n);
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int64_t lda = A_KC ? g.a_rs : g.a_cs;
  const int64_t ldb = B_KC ? g.b_cs : g.b_rs;
  const bool fa = g.a_vec && (A_KC ? true : (m0 + 32 <= g.M || (((g.M + 3) & ~3) <= lda && g.M >= 4)));
  const bool fb = g.b_vec && (B_KC ? true : (n0 + 32 <= g.N || (((g.N + 3) & ~3) <= ldb && g.N >= 4)));
  f32x16 acc, acc2;        // two accumulator chains per wave (even / odd k-steps): see gemm_segment
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
  SA sa[KS_GROUP];
  SB sb[KS_GROUP];
  const int ntile = (g.K + BK - 1) / BK;
  for (int t0 = 0; t0 < ntile; t0 += KS_GROUP) {
#pragma unroll
    for (int j = 0; j < KS_GROUP; ++j) {
      const int k0 = min(t0 + j, ntile - 1) * BK;          // (tiles past the end re-load the last one and are not used)
      sa[j].load(g.A, lda, m0, k0, g.M, g.K, fa, g.kscale);
      sb[j].load(g.B, ldb, n0, k0, g.N, g.K, fb);
    }
    if (t0 > 0) __syncthreads();                           // the previous group's LDS reads are done
#pragma unroll
    for (int j = 0; j < KS_GROUP; ++j) {
      sa[j].store(As + j * TS);
      sb[j].store(Bs + j * TS);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < KS_GROUP / 4; ++h) {
      const int j = wave + 4 * h;
      if (t0 + j < ntile) {
#pragma unroll
        for (int kb = 0; kb < BK; kb += 8) {
          float a[4], b[4];
          SA::frag(As + j * TS, 0, kb, li, lk, a);
          SB::frag(Bs + j * TS, 0, kb, li, lk, b);
#pragma unroll
          for (int q = 0; q < 4; q += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q + 1], b[q + 1], acc2, 0, 0, 0);
          }
        }
      }
    }
  }
  // the four waves' partial tiles meet in LDS: red[wave][row][col], rows of 32 floats
  acc += acc2;
  __syncthreads();
  float* red = lds;
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc[e];
  __syncthreads();
  const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  const int row = m0 + r;
  if (row >= g.M) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(&red[r * 32 + c4]);
#pragma unroll
  for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(&red[w * 1024 + r * 32 + c4]);
  float out[4] = {v.x, v.y, v.z, v.w};
  // what the epilogue reads, for all four columns first (column-clamped addresses, block-uniform conditions: see
  // tile_out above)
  float bias4[4], aux4[4], c_old[4];
  uint8_t kp[4];
  const bool ld_keep = g.epi == EPI_LEAKY_DROPOUT && g.keep != nullptr;
  // own keep bits: the thread's four columns are one Philox block when their flat index row * N + col starts on a
  // multiple of 4 (N % 4 == 0: every layer width in practice) - one block instead of four
  uint32_t kw[4] = {0u, 0u, 0u, 0u};
  const bool quad = g.epi == EPI_LEAKY_DROPOUT && !g.keep && (g.N & 3) == 0 && n0 + c4 + 3 < g.N;
  if (quad) {
    const DpcPhilox4 ph = dpc_philox(g.rng[0], g.rng[1], g.site, ((uint64_t)row * g.N + n0 + c4) >> 2);
    kw[0] = ph.v[0]; kw[1] = ph.v[1]; kw[2] = ph.v[2]; kw[3] = ph.v[3];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int colc = min(n0 + c4 + q, g.N - 1);
    bias4[q] = g.bias ? g.bias[colc] : 0.f;
    aux4[q] = g.epi == EPI_MUL ? g.aux_in[(int64_t)row * g.ld_aux_in + colc] : 1.f;
    c_old[q] = g.beta != 0.f ? g.C[(int64_t)row * g.ldc + colc] : 0.f;
    kp[q] = ld_keep ? g.keep[(int64_t)row * g.ld_keep + colc] : (uint8_t)0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = n0 + c4 + q;
    if (col >= g.N) continue;
    float x = g.alpha * out[q] + bias4[q];
    float* cp = g.C + (int64_t)row * g.ldc + col;
    if (g.epi == EPI_LEAKY_DROPOUT) {
      float keep;
      if (g.keep) keep = kp[q] ? 1.f : 0.f;
      else if (quad) keep = (kw[q] >> 31) ? 1.f : 0.f;
      else keep = (dpc_draw_u32(g.rng, g.site, (uint64_t)row * g.N + col) >> 31) ? 1.f : 0.f;
      const float slope = (x > 0.f ? 1.f : g.leak) * keep * 2.f;
      g.aux_out[(int64_t)row * g.ld_aux_out + col] = slope;
      if (g.seed_out) g.seed_out[(int64_t)row * g.ld_seed + col] = (g.seed_row[row] * g.seed_col[col]) * slope;
      x *= slope;
    } else if (g.epi == EPI_MUL) {
      x *= aux4[q];
    } else if (g.epi == EPI_RELU) {
      x = fmaxf(x, 0.f);
    }
    if (g.beta != 0.f) x += g.beta * c_old[q];
    *cp = x;
  }
}

template <int GRP>
static void launch_ksplit(const GemmArgs& g, bool a_kc, bool b_kc, hipStream_t st) {
  dim3 grid(g.tiles_m * g.tiles_n), block(256);
  if (g.kscale) {
    if (b_kc) hipLaunchKernelGGL((gemm_ksplit_kernel<false, true, true, GRP>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_ksplit_kernel<false, false, true, GRP>), grid, block, 0, st, g);
    return;
  }
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_ksplit_kernel<true, true, false, GRP>), grid, block, 0, st, g);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_ksplit_kernel<true, false, false, GRP>), grid, block, 0, st, g);
  else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_ksplit_kernel<false, true, false, GRP>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_ksplit_kernel<false, false, false, GRP>), grid, block, 0, st, g);
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
      if (g.seed_out) g.seed_out[(int64_t)row * g.ld_seed + col] = (g.seed_row[row] * g.seed_col[col]) * slope;
      v *= slope;
    } else if (g.epi == EPI_MUL) {
      v *= g.aux_in[(int64_t)row * g.ld_aux_in + col];
    } else if (g.epi == EPI_RELU) {
      v = fmaxf(v, 0.f);
    }
    *cp = v;
  }
}

template <int BM, int BN>
static void launch_tile(const GemmArgs& g, bool a_kc, bool b_kc, hipStream_t st) {
  dim3 grid(g.tiles_m * g.tiles_n * g.splitk), block(256);
  if (g.kscale) {     // validated by the caller: A is m-contiguous
    if (b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, 2, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, 2, true>), grid, block, 0, st, g);
    return;
  }
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true, 2>), grid, block, 0, st, g);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false, 2>), grid, block, 0, st, g);
  else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true, 2>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false, 2>), grid, block, 0, st, g);
}

// What gemm_one decided for a product that a grouped launch can take (64x64 tile kernel, nothing to do before or
// after the launch): the argument block and the kernel variant.
struct GemmPlan {
  bool groupable;
  bool a_kc, b_kc, ks;
  GemmArgs g;
};

// One product.  Every tuning choice travels in the descriptor (splitk, tile): the library keeps no GEMM state.
// With `plan`: a product the grouped launch can take is NOT launched - its plan is handed back instead.
static int gemm_one(const dpc_gemm_desc& d, hipStream_t st, GemmPlan* plan = nullptr) {
  if (plan) plan->groupable = false;
  const int M = d.M, N = d.N, K = d.K;
  const float beta = d.beta;
  const int epi = d.epi;
  DPC_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  if (M == 0 || N == 0) return DPC_OK;
  DPC_CHECK_ARG(!d.gat || (d.gidx && d.ldg >= N && beta == 0.f), "gemm: row gather needs ids, ldg >= N and beta == 0");
  DPC_CHECK_ARG(!d.gat || epi == EPI_NONE || epi == EPI_RELU, "gemm: row gather goes with epilogue none or ReLU");
  DPC_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand");
  DPC_CHECK_ARG(d.a_rs == 1 || d.a_cs == 1, "gemm: A needs one unit stride");
  DPC_CHECK_ARG(d.b_rs == 1 || d.b_cs == 1, "gemm: B needs one unit stride");
  DPC_CHECK_ARG(epi >= EPI_NONE && epi <= EPI_RELU, "gemm: unknown epilogue");
  DPC_CHECK_ARG(epi != EPI_LEAKY_DROPOUT || (d.aux_out && (d.keep || d.rng)), "gemm: dropout epilogue needs slope output and a mask source");
  DPC_CHECK_ARG(epi != EPI_MUL || d.aux_in, "gemm: multiply epilogue needs aux_in");
  DPC_CHECK_ARG(d.tile == 0 || d.tile == 32 || d.tile == 324 || d.tile == 64 || d.tile == 128, "gemm: tile must be 0 (automatic), 32, 64 or 128");
  GemmArgs g;
  g.A = d.A; g.a_rs = d.a_rs; g.a_cs = d.a_cs;
  g.B = d.B; g.b_rs = d.b_rs; g.b_cs = d.b_cs;
  g.C = d.C; g.ldc = d.ldc; g.M = M; g.N = N; g.K = K;
  g.alpha = d.alpha; g.beta = beta; g.bias = d.bias; g.epi = epi;
  g.aux_out = d.aux_out; g.ld_aux_out = d.ld_aux_out; g.aux_in = d.aux_in; g.ld_aux_in = d.ld_aux_in;
  g.keep = d.keep; g.ld_keep = d.ld_keep; g.rng = d.rng; g.site = d.site; g.leak = d.leak;
  // k-contiguous when the k stride is 1 (ties, e.g. K == 1, go to the k-contiguous path)
  const bool a_kc = (d.a_cs == 1), b_kc = (d.b_rs == 1);
  DPC_CHECK_ARG(!d.kscale || !a_kc, "gemm: kscale needs an m-contiguous A (the weight-gradient product)");
  g.kscale = d.kscale;
  g.gat = d.gat; g.ldg = d.ldg; g.gidx = d.gidx;
  const int64_t ldb = b_kc ? d.b_cs : d.b_rs;
  // the 16-byte staging loads are typed dword-aligned (f32x4u): any fp32 operand takes them, whatever its row length
  // (a cond vector of C % 4 != 0 columns would otherwise push every product of the generator onto the element-wise
  // edge path).  The flags stay in the argument block for the tile kernels' edge logic.
  g.a_vec = 1;
  g.b_vec = 1;
  // segment maps (GemmArgs::seg_*)
  g.seg_b = d.seg_b; g.seg_c = d.seg_c; g.seg_len = d.seg_len; g.seg_pad = d.seg_pad; g.seg_stride = d.seg_stride;
  g.colstat = d.colstat;
  g.bn_y = d.bn_y; g.ld_bn_y = d.ld_bn_y; g.bn_z = d.bn_z; g.ld_bn_z = d.ld_bn_z; g.bn_mean = d.bn_mean; g.bn_invstd = d.bn_invstd;
  g.stat_n = d.bn_y ? d.stat_n : N;
  g.stat_slots = d.stat_slots > 1 ? d.stat_slots : 1;
  DPC_CHECK_ARG(d.stat_slots >= 0 && d.stat_slots <= 64, "gemm: stat_slots must be 0..64");
  DPC_CHECK_ARG(!d.bn_y || (d.colstat && d.bn_z && d.bn_mean && d.bn_invstd && d.stat_n >= 1 && d.stat_n <= N &&
                            d.ld_bn_y >= d.stat_n && d.ld_bn_z >= d.stat_n && (beta == 0.f || beta == 1.f)),
                "gemm: BatchNorm-backward statistics need colstat, y, z, mean, invstd and 1 <= stat_n <= N");
  g.force_atomic = d.atomic ? 1 : 0;
  g.seed_out = d.seed_out; g.ld_seed = d.ld_seed; g.seed_row = d.seed_row; g.seed_col = d.seed_col;
  DPC_CHECK_ARG(!d.seed_out || (epi == EPI_LEAKY_DROPOUT && d.seed_row && d.seed_col && d.ld_seed >= N),
                "gemm: the chain seed goes with the LeakyReLU + Dropout epilogue and needs seed_row, seed_col and ld_seed >= N");
  DPC_CHECK_ARG(!d.atomic || (beta == 1.f && epi == EPI_NONE && !d.bias && !d.gat && !d.colstat && d.tile != 32 && d.tile != 324),
                "gemm: atomic accumulation needs beta == 1 and a plain product");
  DPC_CHECK_ARG(!d.colstat || (epi == EPI_NONE && !d.seg_c && (d.splitk == 0 || d.splitk == 1)),
                "gemm: column statistics go with epilogue none, a plain C and no split-K");
  if (d.seg_b || d.seg_c) {
    DPC_CHECK_ARG(d.seg_b >= 0 && d.seg_b <= 2 && (d.seg_c == 0 || d.seg_c == 1), "gemm: unknown segment map");
    DPC_CHECK_ARG(d.seg_pad > 0 && d.seg_len > 0 && d.seg_len <= d.seg_pad && d.seg_stride >= d.seg_len, "gemm: bad segment map");
    if (d.seg_b == 1) {
      DPC_CHECK_ARG(b_kc && d.seg_pad % BK == 0 && K % d.seg_pad == 0, "gemm: a K segment map needs a k-contiguous B, seg_pad % 32 == 0 and K a multiple of seg_pad");
      DPC_CHECK_ARG((int64_t)(K / d.seg_pad - 1) * d.seg_stride + d.seg_pad <= ldb, "gemm: K segment map runs past B's row");
      g.b_vec = 1;      // dword-aligned 16-byte loads (f32x4u)
    } else if (d.seg_b == 2) {
      DPC_CHECK_ARG(!b_kc && d.seg_pad % 4 == 0 && N % d.seg_pad == 0, "gemm: an N segment map needs an n-contiguous B, seg_pad % 4 == 0 and N a multiple of seg_pad");
      DPC_CHECK_ARG((int64_t)(N / d.seg_pad - 1) * d.seg_stride + d.seg_pad <= ldb, "gemm: N segment map runs past B's row");
      g.b_vec = 1;
    }
    if (d.seg_c) {
      DPC_CHECK_ARG(epi == EPI_NONE && !d.bias && !d.gat && N % d.seg_pad == 0, "gemm: a segment-mapped C takes no epilogue, and N a multiple of seg_pad");
      DPC_CHECK_ARG((int64_t)(N / d.seg_pad - 1) * d.seg_stride + d.seg_len <= d.ldc, "gemm: C segment map runs past ldc");
      DPC_CHECK_ARG(beta == 1.f, "gemm: a segment-mapped C is accumulated into (beta == 1)");
    }
  }
  DPC_CHECK_ARG(d.seg_c || d.ldc >= N, "gemm: ldc < N");
  // Tile choice (measured per shape with `tools/gemm_bench.py tiles`): the 128x128 tile reaches
  // 120-130 TF/s (~80 % of the fp32 MFMA peak) once the grid holds >= 2 blocks per CU, but every
  // GEMM of the T1M training step (M*N <= 8192*512) is faster on 64x64 tiles, whose 4x as many
  // blocks fill the 256 CUs.
  int bm = 64, bn = 64;
  if (d.tile == 128 || (d.tile == 0 && dpc_cdiv(M, 128) * dpc_cdiv(N, 128) >= 512)) { bm = 128; bn = 128; }
  // small product (fewer 64x64 tiles than CUs, short K): 32x32 tiles whose waves split K (gemm_ksplit_kernel)
  const bool plain = !d.gat && !d.colstat && !d.seg_b && !d.seg_c && !d.atomic;
  if (d.tile == 32 || d.tile == 324 || (d.tile == 0 && plain && d.splitk <= 1 && K <= 1024 && K >= 64 &&
                       dpc_cdiv(M, 64) * dpc_cdiv(N, 64) < 256 && dpc_cdiv(M, 32) * dpc_cdiv(N, 32) >= 64)) {
    DPC_CHECK_ARG(plain && d.splitk <= 1, "gemm: the 32x32 K-split tile takes no row gather, column statistics, segment map or split-K");
    g.tiles_m = (int)dpc_cdiv(M, 32);
    g.tiles_n = (int)dpc_cdiv(N, 32);
    g.splitk = 1; g.kchunk = K;
    // up to two blocks per CU: groups of 8 K-tiles (72 KB of LDS); more: groups of 4 (36 KB, four blocks per CU)
    if ((int64_t)g.tiles_m * g.tiles_n > 512 || d.tile == 324) launch_ksplit<4>(g, a_kc, b_kc, st);
    else launch_ksplit<8>(g, a_kc, b_kc, st);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
  }
  g.tiles_m = (int)dpc_cdiv(M, bm);
  g.tiles_n = (int)dpc_cdiv(N, bn);
  const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n;
  bool deferred_epi = false;
  int splitk = d.colstat ? 1 : d.splitk;
  if (splitk <= 0) {
    // auto.  Cost model fitted to MI355X measurements of the 64x64 tile (tools/gemm_occupancy.py,
    // tools/gemm_split_sweep.py): a CU works through its resident blocks at 17.5 ns per k and block
    // (21 ns when it holds only one: a single wave per SIMD hides less latency).  One block per tile
    // spreads `blocks` tiles over 256 CUs - 270 tiles take as long as 512 - while split-K cuts each tile's
    // K range into sp chunks and pays fp32 atomics (~2.5 TB/s, mostly hidden behind other
    // blocks' MFMAs), a memset, and a separate pass for a bias / activation epilogue.
    splitk = 1;
    const bool can_split = (beta == 0.f || (beta == 1.f && epi == EPI_NONE && !d.bias)) && bm == 64 && bn == 64 && K >= 512;
    if (can_split) {
      const bool deferred = (epi != EPI_NONE || d.bias || d.gat);
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
  if (splitk > 1) {
    DPC_CHECK_ARG(bm == 64 && bn == 64, "gemm: split-K runs on the 64x64 tile");
    DPC_CHECK_ARG(beta == 0.f || beta == 1.f, "gemm: split-K needs beta in {0,1}");
    if (epi != EPI_NONE || d.bias || d.gat) {
      DPC_CHECK_ARG(beta == 0.f, "gemm: split-K with an epilogue needs beta == 0");
      deferred_epi = true;
    }
  }
  g.splitk = splitk;
  const int ktiles = (int)dpc_cdiv(K, BK);
  g.kchunk = (int)dpc_cdiv(ktiles, splitk) * BK;
  if (K == 0) g.kchunk = BK;
  if (splitk > 1 && beta == 0.f) {
    hipError_t e = hipMemset2DAsync(d.C, d.ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, st);
    if (e != hipSuccess) { dpc_set_error(hipGetErrorString(e)); return DPC_EHIP; }
    g.beta = 1.f;   // accumulate into the zeroed C
  }
  GemmArgs full = g;
  if (plan && bm == 64 && !deferred_epi) {       // (a split product's zero fill of C, if any, is on the stream already)
    plan->groupable = true;
    plan->a_kc = a_kc; plan->b_kc = b_kc; plan->ks = d.kscale != nullptr;
    plan->g = g;
    return DPC_OK;
  }
  if (deferred_epi) { g.epi = EPI_NONE; g.bias = nullptr; g.gat = nullptr; }
  if (bm == 128) launch_tile<128, 128>(g, a_kc, b_kc, st);
  else launch_tile<64, 64>(g, a_kc, b_kc, st);
  if (deferred_epi)
    hipLaunchKernelGGL(gemm_epilogue_kernel, dim3(dpc_grid((int64_t)M * N, 256)), dim3(256), 0, st, full);
  DPC_LAUNCH_CHECK();
  return DPC_OK;
}

// The GEMM entry point: `count` products described by `d[0 .. count)` (host memory), issued in order on `stream`.
extern "C" int dpc_gemm_f32_ex(const dpc_gemm_desc* d, int count, void* stream) {
  DPC_CHECK_ARG(d && count >= 1 && count <= 64, "gemm_ex: 1..64 descriptors");
  for (int i = 0; i < count; ++i) {
    int rc = gemm_one(d[i], (hipStream_t)stream);
    if (rc != DPC_OK) return rc;
  }
  return DPC_OK;
}

template <bool A_KC, bool B_KC, bool KS>
static void launch_group_variant(const GemmGroupArgs& G, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((gemm_group_kernel<A_KC, B_KC, KS>), dim3(blocks), dim3(256), 0, st, G);
}

// `count` INDEPENDENT products (no product reads what another of the set writes; products that ADD into one C must
// ask for atomic accumulation).  Those that run on the 64x64 tile kernel with the same operand layouts share ONE
// launch, up to four at a time; the others are issued on their own, in the order given.
extern "C" int dpc_gemm_f32_group(const dpc_gemm_desc* d, int count, void* stream) {
  DPC_CHECK_ARG(d && count >= 1 && count <= 64, "gemm_group: 1..64 descriptors");
  hipStream_t st = (hipStream_t)stream;
  GemmPlan plans[64];
  for (int i = 0; i < count; ++i) {
    int rc = gemm_one(d[i], st, &plans[i]);
    if (rc != DPC_OK) return rc;
  }
  bool done[64];
  for (int i = 0; i < count; ++i) done[i] = !plans[i].groupable;
  for (int i = 0; i < count; ++i) {
    if (done[i]) continue;
    GemmGroupArgs G;
    G.count = 0;
    int blocks = 0;
    for (int j = i; j < count && G.count < DPC_GROUP_MAX; ++j) {
      if (done[j] || plans[j].a_kc != plans[i].a_kc || plans[j].b_kc != plans[i].b_kc || plans[j].ks != plans[i].ks) continue;
      G.start[G.count] = blocks;
      G.g[G.count] = plans[j].g;
      blocks += plans[j].g.tiles_m * plans[j].g.tiles_n * plans[j].g.splitk;
      ++G.count;
      done[j] = true;
    }
    for (int k = G.count; k <= DPC_GROUP_MAX; ++k) G.start[k] = blocks;
    const bool a_kc = plans[i].a_kc, b_kc = plans[i].b_kc, ks = plans[i].ks;
    if (ks) {
      if (b_kc) launch_group_variant<false, true, true>(G, blocks, st);
      else launch_group_variant<false, false, true>(G, blocks, st);
    } else if (a_kc && b_kc) launch_group_variant<true, true, false>(G, blocks, st);
    else if (a_kc && !b_kc) launch_group_variant<true, false, false>(G, blocks, st);
    else if (!a_kc && b_kc) launch_group_variant<false, true, false>(G, blocks, st);
    else launch_group_variant<false, false, false>(G, blocks, st);
    DPC_LAUNCH_CHECK();
  }
  return DPC_OK;
}

static dpc_gemm_desc gemm_plain_desc(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                                     const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                                     const float* bias, int epi) {
  dpc_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.M = M; d.N = N; d.K = K; d.alpha = alpha; d.beta = beta; d.A = A; d.a_rs = a_rs; d.a_cs = a_cs; d.B = B; d.b_rs = b_rs;
  d.b_cs = b_cs; d.C = C; d.ldc = ldc; d.bias = bias; d.epi = epi; d.leak = 0.2f;
  return d;
}

extern "C" int dpc_gemm_f32(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                            const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                            const float* bias, int epi, float* aux_out, int64_t ld_aux_out,
                            const float* aux_in, int64_t ld_aux_in, const uint8_t* keep, int64_t ld_keep,
                            const uint64_t* rng, uint32_t site, float leak, int splitk, const float* kscale,
                            void* stream) {
  dpc_gemm_desc d = gemm_plain_desc(M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, beta, C, ldc, bias, epi);
  d.aux_out = aux_out; d.ld_aux_out = ld_aux_out; d.aux_in = aux_in; d.ld_aux_in = ld_aux_in; d.keep = keep;
  d.ld_keep = ld_keep; d.rng = rng; d.site = site; d.leak = leak; d.splitk = splitk; d.kscale = kscale;
  return gemm_one(d, (hipStream_t)stream);
}

// Linear over an input whose trailing columns are a one-/two-hot conditional vector (the generator's
// input cat([z, cond]) and the Residual blocks' cat([out, input]), dp_cgan.py:83,606): A and B cover the
// dense columns only, and the hot columns' share of x W^T - the sum of those columns of W - is added
// from gat = W[:, cond columns]^T [C, ldg] by row ids gidx int32 [M][2] (negative = none).
extern "C" int dpc_gemm_f32_gather(int M, int N, int K, float alpha, const float* A, int64_t a_rs, int64_t a_cs,
                                   const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc, const float* bias,
                                   int epi, const float* gat, int64_t ldg, const int32_t* gidx, void* stream) {
  DPC_CHECK_ARG(gat && gidx, "gemm_gather: null pointer");
  dpc_gemm_desc d = gemm_plain_desc(M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, 0.f, C, ldc, bias, epi);
  d.gat = gat; d.ldg = ldg; d.gidx = gidx;
  return gemm_one(d, (hipStream_t)stream);
}
