// Generic fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain) for gfx950.
//
//   C[M][N] = act( alpha * sum_k A(m,k) * B(k,n) + bias + (accumulate ? C : 0) )
//
// A(m,k) = transA ? A[k*lda+m] : A[gatherA(m)*lda+k]      B(k,n) = transB ? B[n*ldb+k] : B[gatherB(k)*ldb+n]
// Used for every GEMM-shaped piece of the UMPR path that is not the 3x3 convolution: GRU input projection
// (fused embedding-row gather), co-attention projections, SNet / CNet projections, VGG classifier, and all
// the weight-gradient contractions (split-K, deterministic slab reduce).
//
// Tiling: 256 threads = 4 waves (2x2); block tile BMxBNx16, LDS images are k-major ([k][m], [k][n]) so an MFMA
// operand read is 32 consecutive floats per half-wave (conflict-free ds_read_b32); global->register prefetch
// of tile t+1 is issued before the MFMAs of tile t, written to the other LDS buffer afterwards (one barrier
// per k-tile).  MFMA-A = rows of C (m), MFMA-B = columns of C (n): lanes run along n, so C stores are 128-B
// row segments.
#include "umpr_common.h"
#include "umpr_internal.h"
#include "umpr_tiles.h"
#include <stdlib.h>

namespace {

struct GemmParams {
  const float* A; long lda;
  const float* B; long ldb;
  float* C; long ldc;
  int M, N, K;
  const int64_t* gatherA;
  const int64_t* gatherB;
  const float* bias; int bias_mode;  // 1: bias[n], 2: bias[m]
  int act; int accumulate; float alpha;
  int split_k; int k_per_split; float* ws;
  int vecA, vecB;
  int waL, waD, waP, wbL, wbD, wbP;   // sliding-window operands (UmprGemm::winA / winB), L == 0: off
};

// B16 (mixed-precision mode, text path): operands stay fp32 in memory and are rounded to bf16 when a stage is written to
// LDS ([row][16 k] images, 48-B rows); one v_mfma_f32_32x32x16_bf16 per 32 x 32 tile and stage instead of eight fp32 MFMAs,
// fp32 accumulation and epilogue unchanged.  The kernel is then bound by its global loads, not by the matrix pipe.
typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));

// Registers: the fp32 form holds two accumulator sets (chain + running total) and sits at 244-308 VGPR+AGPR, i.e. one or two
// workgroups per CU; the bf16 form needs one set and two staging sets (~170): three workgroups per CU, which is what hides its
// global-load latency (at 308 registers - one workgroup per CU - it was no faster than the fp32 form: tools/bench_gemm.py).
template <int BM, int BN, bool TA, bool TB, bool B16 = false>
__global__ __launch_bounds__(256, B16 && BM * BN < 128 * 128 ? 3 : 2) void gemm_f32_kernel(GemmParams p) {
  constexpr int BKS = BK;   // k-depth of a stage (32 for the bf16 path measured slower: 10.23-10.29 vs 10.08-10.13 ms per step)
  using LA = TileRegs<BM, !TA, BKS>;  // A non-trans is k-contiguous
  using LB = TileRegs<BN, TB, BKS>;   // B trans is k-contiguous
  constexpr int LDA = LA::LD, LDB = LB::LD;
  constexpr int WTM = BM / 2, WTN = BN / 2;  // 2x2 waves
  constexpr int TM = WTM / 32, TN = WTN / 32;
  __shared__ __attribute__((aligned(16))) float As[B16 ? 1 : 2][B16 ? 4 : BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[B16 ? 1 : 2][B16 ? 4 : BK * LDB];
  __shared__ __attribute__((aligned(16))) __bf16 Ah[B16 ? 2 : 1][B16 ? LA::B16_ELEMS : 8];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[B16 ? 2 : 1][B16 ? LB::B16_ELEMS : 8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int split = blockIdx.z;
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);

  // two-level accumulation: the MFMA chain runs over KFLUSH k-tiles, then folds into `tot`.  A single fp32 chain
  // over K = 4608 loses ~6x more bits than the blocked sums of the reference's CPU kernels (gradients through 13
  // conv layers showed it); chains of 128 restore parity at < 1% cost.
  f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

  LA ra;
  LB rb;
  const int nt = (kend - kbeg + BKS - 1) / BKS;
  ra.cache_rows(p.gatherA, m0, p.M, tid);
  if (nt > 0) {
    ra.load(p.A, p.lda, p.gatherA, m0, p.M, kbeg, kend, p.vecA, tid, p.waL, p.waD, p.waP);
    rb.load(p.B, p.ldb, p.gatherB, n0, p.N, kbeg, kend, p.vecB, tid, p.wbL, p.wbD, p.wbP);
    if (B16) { ra.store_b16(Ah[0], tid); rb.store_b16(Bh[0], tid); }
    else { ra.store(As[0], tid); rb.store(Bs[0], tid); }
  }
  __syncthreads();
  const int l31 = lane & 31, kh = lane >> 5;
  if constexpr (B16) {
    // bf16 path: the MFMA work of a stage is a few hundred cycles, the stage is bound by the latency of its global loads -
    // so TWO stages are kept in flight (register sets ra / ra2 alternate; the loop is unrolled by two so that both are
    // named at compile time).  One accumulation chain: the two-level fold below exists for fp32 operand accuracy.
    LA ra2 = ra;
    LB rb2 = rb;
    auto mfma_stage = [&](int cur) {
#pragma unroll
      for (int ks = 0; ks < BKS / 16; ++ks) {
        gemm_bf16x8 ah[TM], bh[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = LA::frag_b16(Ah[cur], wm * WTM + i * 32, lane, ks);
#pragma unroll
        for (int j = 0; j < TN; ++j) bh[j] = LB::frag_b16(Bh[cur], wn * WTN + j * 32, lane, ks);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) tot[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], tot[i][j], 0, 0, 0);
      }
    };
    if (nt > 1) {
      ra.load(p.A, p.lda, p.gatherA, m0, p.M, kbeg + BKS, kend, p.vecA, tid, p.waL, p.waD, p.waP);
      rb.load(p.B, p.ldb, p.gatherB, n0, p.N, kbeg + BKS, kend, p.vecB, tid, p.wbL, p.wbD, p.wbP);
    }
    for (int t = 0; t < nt; t += 2) {
      // LDS[0] holds stage t, ra / rb stage t + 1 (in flight)
      if (t + 2 < nt) {
        ra2.load(p.A, p.lda, p.gatherA, m0, p.M, kbeg + (t + 2) * BKS, kend, p.vecA, tid, p.waL, p.waD, p.waP);
        rb2.load(p.B, p.ldb, p.gatherB, n0, p.N, kbeg + (t + 2) * BKS, kend, p.vecB, tid, p.wbL, p.wbD, p.wbP);
      }
      mfma_stage(0);
      if (t + 1 < nt) { ra.store_b16(Ah[1], tid); rb.store_b16(Bh[1], tid); }
      __syncthreads();
      if (t + 1 >= nt) break;
      // LDS[1] holds stage t + 1, ra2 / rb2 stage t + 2 (in flight)
      if (t + 3 < nt) {
        ra.load(p.A, p.lda, p.gatherA, m0, p.M, kbeg + (t + 3) * BKS, kend, p.vecA, tid, p.waL, p.waD, p.waP);
        rb.load(p.B, p.ldb, p.gatherB, n0, p.N, kbeg + (t + 3) * BKS, kend, p.vecB, tid, p.wbL, p.wbD, p.wbP);
      }
      mfma_stage(1);
      if (t + 2 < nt) { ra2.store_b16(Ah[0], tid); rb2.store_b16(Bh[0], tid); }
      __syncthreads();
    }
  } else
  // outer loop = one MFMA accumulation chain (KFLUSH stages): inside it the accumulators are written only by
  // MFMAs and stay in AGPRs (a conditional fold inside the stage loop made hipcc move all of them through VGPRs
  // every stage: 128 v_accvgpr moves + an MFMA pipeline drain per stage)
  for (int t0 = 0; t0 < nt; t0 += KFLUSH) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int t1 = min(nt, t0 + KFLUSH);
  for (int t = t0; t < t1; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      ra.load(p.A, p.lda, p.gatherA, m0, p.M, kbeg + (t + 1) * BKS, kend, p.vecA, tid, p.waL, p.waD, p.waP);
      rb.load(p.B, p.ldb, p.gatherB, n0, p.N, kbeg + (t + 1) * BKS, kend, p.vecB, tid, p.wbL, p.wbD, p.wbP);
    }
    const float* as = As[cur] + wm * WTM + l31;
    const float* bs = Bs[cur] + wn * WTN + l31;
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = as[kh * LDA + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = bs[kh * LDB + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int cb = kk & 1, nb = cb ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nb][i] = as[(2 * (kk + 1) + kh) * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nb][j] = bs[(2 * (kk + 1) + kh) * LDB + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[cb][i], b[cb][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + 1 < nt) {
      ra.store(As[cur ^ 1], tid);
      rb.store(Bs[cur ^ 1], tid);
    }
    __syncthreads();
  }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) tot[i][j] += acc[i][j];
  }

  // epilogue
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + mfma_row(r, lane);
        if (row < p.M && col < p.N) {
          float v = p.alpha * tot[i][j][r];
          if (p.split_k > 1) {
            p.ws[((long)split * p.M + row) * p.N + col] = v;
          } else {
            if (p.bias_mode == 1) v += p.bias[col];
            else if (p.bias_mode == 2) v += p.bias[row];
            float* c = p.C + (long)row * p.ldc + col;
            if (p.accumulate) v += *c;
            *c = apply_act(v, p.act);
          }
        }
      }
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, float* __restrict__ C,
                                     long ldc, const float* __restrict__ bias, int bias_mode, int act, int accumulate) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / N), col = (int)(i % N);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;  // four interleaved chains, fixed order: bitwise reproducible
    int s = 0;
    for (; s + 3 < splits; s += 4) {
      v0 += ws[(long)s * total + i]; v1 += ws[(long)(s + 1) * total + i];
      v2 += ws[(long)(s + 2) * total + i]; v3 += ws[(long)(s + 3) * total + i];
    }
    for (; s < splits; ++s) v0 += ws[(long)s * total + i];
    float v = (v0 + v1) + (v2 + v3);
    if (bias_mode == 1) v += bias[col];
    else if (bias_mode == 2) v += bias[row];
    float* c = C + (long)row * ldc + col;
    if (accumulate) v += *c;
    *c = apply_act(v, act);
  }
}

template <int BM, int BN>
void launch_tile_b16(const GemmParams& p, bool ta, bool tb, dim3 grid, hipStream_t s) {
  if (!ta && !tb) gemm_f32_kernel<BM, BN, false, false, true><<<grid, 256, 0, s>>>(p);
  else if (!ta && tb) gemm_f32_kernel<BM, BN, false, true, true><<<grid, 256, 0, s>>>(p);
  else if (ta && !tb) gemm_f32_kernel<BM, BN, true, false, true><<<grid, 256, 0, s>>>(p);
  else gemm_f32_kernel<BM, BN, true, true, true><<<grid, 256, 0, s>>>(p);
}

template <int BM, int BN>
void launch_tile(const GemmParams& p, bool ta, bool tb, dim3 grid, hipStream_t s) {
  if (!ta && !tb) gemm_f32_kernel<BM, BN, false, false><<<grid, 256, 0, s>>>(p);
  else if (!ta && tb) gemm_f32_kernel<BM, BN, false, true><<<grid, 256, 0, s>>>(p);
  else if (ta && !tb) gemm_f32_kernel<BM, BN, true, false><<<grid, 256, 0, s>>>(p);
  else gemm_f32_kernel<BM, BN, true, true><<<grid, 256, 0, s>>>(p);
}

}  // namespace

// per host thread: products issued while it is set round their operands to bf16 (umpr_set_gemm_bf16, mixed-precision mode)
static thread_local int t_gemm_b16 = 0;
void umpr_gemm_set_b16(int on) { t_gemm_b16 = on; }

int umpr_gemm(const UmprGemm& g, hipStream_t stream) {
  UMPR_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0, "gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
  UMPR_REQUIRE(g.A && g.B && g.C, "gemm: null operand");
  UMPR_REQUIRE(!(g.gatherA && g.transA) || true, "gemm");
  GemmParams p;
  p.A = g.A; p.lda = g.lda; p.B = g.B; p.ldb = g.ldb; p.C = g.C; p.ldc = g.ldc;
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.gatherA = g.gatherA; p.gatherB = g.gatherB;
  p.bias = g.bias; p.bias_mode = g.bias ? g.bias_mode : 0;
  p.act = g.act; p.accumulate = g.accumulate ? 1 : 0; p.alpha = g.alpha;
  UMPR_REQUIRE(!(g.winA_L && (g.transA || g.gatherA || (g.winA_D & 3))) && !(g.winB_L && (g.transB || g.gatherB || (g.winB_D & 3))),
               "gemm: a sliding-window operand must be row-major, ungathered, with a row length that is a multiple of 4");
  UMPR_REQUIRE((!g.winA_L || g.lda == g.winA_D) && (!g.winB_L || g.ldb == g.winB_D), "gemm: window operand with a padded row pitch");
  p.waL = g.winA_L; p.waD = g.winA_D; p.waP = g.winA_pad; p.wbL = g.winB_L; p.wbD = g.winB_D; p.wbP = g.winB_pad;
  // float4 staging needs 16-B aligned rows and a contiguous extent that is a multiple of 4 (no partial vectors)
  p.vecA = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && (((g.transA ? g.M : g.K) & 3) == 0);
  p.vecB = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) && (((g.transB ? g.K : g.N) & 3) == 0);
  int BM = g.M <= 64 ? 64 : 128;
  int BN = g.N <= 64 ? 64 : 128;
  // a grid that leaves CUs idle (the text path's [12800 x 128] x 128 products: 100 tiles of 128 x 128) takes smaller
  // tiles: each halving doubles the workgroups in flight
  static const int small_grid = umpr_env_int("UMPR_GEMM_SMALL_GRID", 384);
  // (not for the deep-K products that split K over workgroups below: there the split fills the chip and a smaller
  // tile only re-reads more - the [384 x 300] dW_ih products got 16 % slower with 64 x 64 tiles)
  const bool will_split = g.split_k != 1 && g.K >= 512;
  if (!will_split && (long)cdiv(g.M, BM) * cdiv(g.N, BN) < small_grid && BM == 128) BM = 64;
  if (!will_split && (long)cdiv(g.M, BM) * cdiv(g.N, BN) < small_grid && BN == 128) BN = 64;
  const int tm = cdiv(g.M, BM), tn = cdiv(g.N, BN);
  int split = g.split_k;
  if (split <= 0) {  // auto: aim for >= 512 workgroups when K is deep enough to share
    split = 1;
    const long tiles = (long)tm * tn;
    // batch-sized M (the VGG classifier at 64 images): the kernel streams a weight matrix once and is bound by memory
    // latency, not by the MFMA pipe or HBM - more workgroups in flight (3 fit a CU) hide it
    static const int small_m_target = umpr_env_int("UMPR_GEMM_SMALL_M_WGS", 512);
    static const int b16_target = umpr_env_int("UMPR_GEMM_B16_WGS", 512);
    const int target = t_gemm_b16 ? b16_target : (g.M <= 64 ? small_m_target : 512);
    if (tiles < target / 2 && g.K >= 512) {
      split = (int)((target + tiles - 1) / tiles);
      const int maxs = g.K / 128;
      if (split > maxs) split = maxs;
      if (split < 1) split = 1;
    }
  }
  if (split > 1) {
    const size_t need = (size_t)split * g.M * g.N * sizeof(float);
    if (!g.ws || g.ws_bytes < need) {
      const size_t per = (size_t)g.M * g.N * sizeof(float);
      split = g.ws ? (int)(g.ws_bytes / per) : 1;
      if (split < 1) split = 1;
    }
  }
  int kps = cdiv(cdiv(g.K, split), BK) * BK;
  if (kps < BK) kps = BK;
  split = g.K > 0 ? cdiv(g.K, kps) : 1;
  p.split_k = split; p.k_per_split = kps; p.ws = g.ws;
  dim3 grid(tn, tm, split);
  UmprProfScope prof(UMPR_K_GEMM, 2.0 * g.M * g.N * g.K, stream);
  if (t_gemm_b16) {
    if (BM == 128 && BN == 128) launch_tile_b16<128, 128>(p, g.transA, g.transB, grid, stream);
    else if (BM == 64 && BN == 128) launch_tile_b16<64, 128>(p, g.transA, g.transB, grid, stream);
    else if (BM == 128 && BN == 64) launch_tile_b16<128, 64>(p, g.transA, g.transB, grid, stream);
    else launch_tile_b16<64, 64>(p, g.transA, g.transB, grid, stream);
  } else if (BM == 128 && BN == 128) launch_tile<128, 128>(p, g.transA, g.transB, grid, stream);
  else if (BM == 64 && BN == 128) launch_tile<64, 128>(p, g.transA, g.transB, grid, stream);
  else if (BM == 128 && BN == 64) launch_tile<128, 64>(p, g.transA, g.transB, grid, stream);
  else launch_tile<64, 64>(p, g.transA, g.transB, grid, stream);
  UMPR_LAUNCH_CHECK("gemm_f32");
  if (split > 1) {
    const long total = (long)g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    splitk_reduce_kernel<<<blocks, 256, 0, stream>>>(g.ws, split, g.M, g.N, g.C, g.ldc, p.bias, p.bias_mode, g.act,
                                                     p.accumulate);
    UMPR_LAUNCH_CHECK("splitk_reduce");
  }
  return 0;
}
