// R-Net co-attention (src/model.py:50-55) for gfx950, hidden width 2u = 128.
//
//   A = tanh(G_i M G_u^T)   soft_u = softmax_k(max_j A[j,k])   soft_i = softmax_j(max_k A[j,k])
//   atte_u = G_u^T soft_u   atte_i = G_i^T soft_i                       (no padding mask, like the reference)
//
// The SLxSL affinity matrix is never written to HBM: T = G_i M comes from the GEMM kernel, `scores` forms 64x64
// tiles of tanh(T G_u^T) on v_mfma_f32_32x32x2_f32 and keeps only row/column maxima with their (first) argmax, and
// the backward is sparse (<= 2*SL non-zeros of dA per sample), routed through the saved argmax indices.
#include "umpr_common.h"
#include "umpr_internal.h"

namespace {

constexpr int D = 128;   // 2 * gru_size

struct ScoreParams {
  const float* T;   // [B][SL][128] = G_i M
  const float* Gu;  // [B][SL][128]
  float* rowmax_part; int* argrow_part;  // [B][ksplit][SL]: maxima over the column tiles of one split
  float* colmax_part; int* argcol_part;  // [B][nblk][SL]
  int SL, nblk, ksplit;                  // grid (nblk row blocks, B, ksplit): column tiles are dealt out over blockIdx.z
};

__device__ __forceinline__ void better_first(float& v, int& i, float v2, int i2) {
  // keep the maximum; on ties the smaller index (first occurrence)
  if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// 64 rows x 128 columns of src (rows past SL as zeros) -> dst[row][LDR], row-major: float4 in, float4 out.  All eight
// loads of a thread are issued before the first is used, with clamped row addresses: a per-lane `valid ? load : 0`
// makes hipcc wait for each load inside its branch (one global round trip per element, ~15 us per tile).
constexpr int LDR = 132;   // 16 lanes x ds_read_b128 at a row stride of 132 floats touch 64 distinct banks
__device__ __forceinline__ void stage_tile_f32(const float* __restrict__ src, int row0, int SL, float* __restrict__ dst,
                                               int tid) {
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 256 * i;
    const int j = e >> 5, c4 = (e & 31) * 4;
    v[i] = *reinterpret_cast<const float4*>(src + (long)min(row0 + j, SL - 1) * D + c4);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 256 * i;
    const int j = e >> 5, c4 = (e & 31) * 4;
    *reinterpret_cast<float4*>(dst + j * LDR + c4) = row0 + j < SL ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// One workgroup = 64 rows (j) x the column tiles of its split.  Wave (wi, wu) owns a 32 x 32 tile; lane half kh takes
// the reduction columns c = 64 kh + kk (any split of the 128 columns works as long as A and B agree), so a lane's 64
// operand values per matrix are contiguous: 16 ds_read_b128 each, all in registers before the 64 MFMAs, which run as
// two independent accumulator chains.
__global__ __launch_bounds__(256) void coattn_scores_kernel(ScoreParams p) {
  __shared__ __attribute__((aligned(16))) float Ts[64 * LDR];
  __shared__ __attribute__((aligned(16))) float Us[64 * LDR];
  __shared__ float xv[2][64]; __shared__ int xi[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wu = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int SL = p.SL;
  const int j0 = blk * 64;
  const float* Tb = p.T + (long)b * SL * D;
  const float* Ub = p.Gu + (long)b * SL * D;

  stage_tile_f32(Tb, j0, SL, Ts, tid);
  float rbest[16]; int ridx[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { rbest[r] = -INFINITY; ridx[r] = 0x7fffffff; }
  __syncthreads();
  float4 ta[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) ta[q] = *reinterpret_cast<const float4*>(Ts + (wi * 32 + l31) * LDR + 64 * kh + 4 * q);

  const int ntile = (SL + 63) / 64;
  const int tper = (ntile + p.ksplit - 1) / p.ksplit;
  const int ut_end = min(ntile, ((int)blockIdx.z + 1) * tper);
  for (int ut = blockIdx.z * tper; ut < ut_end; ++ut) {
    const int k0 = ut * 64;
    __syncthreads();
    stage_tile_f32(Ub, k0, SL, Us, tid);
    __syncthreads();
    float4 ub[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) ub[q] = *reinterpret_cast<const float4*>(Us + (wu * 32 + l31) * LDR + 64 * kh + 4 * q);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc0 = mfma32(ta[q].x, ub[q].x, acc0);
      acc1 = mfma32(ta[q].y, ub[q].y, acc1);
      acc0 = mfma32(ta[q].z, ub[q].z, acc0);
      acc1 = mfma32(ta[q].w, ub[q].w, acc1);
    }
    const int kcol = k0 + wu * 32 + l31;
    const bool kvalid = kcol < SL;
    float cbest = -INFINITY; int cidx = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + wi * 32 + mfma_row(r, lane);
      const float a = tanhf(acc0[r] + acc1[r]);
      if (kvalid && j < SL) {
        better_first(rbest[r], ridx[r], a, kcol);
        better_first(cbest, cidx, a, j);
      }
    }
    // column max: combine the two lane halves (rows +4), then the two wi waves
    {
      const float ov = __shfl_xor(cbest, 32, 64);
      const int oi = __shfl_xor(cidx, 32, 64);
      better_first(cbest, cidx, ov, oi);
    }
    if (wi == 1 && kh == 0) { xv[wu][l31] = cbest; xi[wu][l31] = cidx; }
    __syncthreads();
    if (wi == 0 && kh == 0) {
      better_first(cbest, cidx, xv[wu][l31], xi[wu][l31]);
      if (kvalid) {
        const long o = ((long)b * p.nblk + blk) * SL + kcol;
        p.colmax_part[o] = cbest;
        p.argcol_part[o] = cidx;
      }
    }
  }
  // row max: reduce over the 32 lanes of each half (columns), then over the two wu waves
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = rbest[r]; int i = ridx[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(i, o, 64);
      better_first(v, i, ov, oi);
    }
    rbest[r] = v; ridx[r] = i;
  }
  // rows of this wave: wi*32 + mfma_row(r, lane); lanes l31==0 of each half hold the result
  if (wu == 1 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = wi * 32 + mfma_row(r, lane);
      xv[0][jr] = rbest[r]; xi[0][jr] = ridx[r];
    }
  }
  __syncthreads();
  if (wu == 0 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = wi * 32 + mfma_row(r, lane);
      float v = rbest[r]; int i = ridx[r];
      better_first(v, i, xv[0][jr], xi[0][jr]);
      if (j0 + jr < SL) {
        const long o = ((long)b * p.ksplit + blockIdx.z) * SL + j0 + jr;
        p.rowmax_part[o] = v;
        p.argrow_part[o] = i;
      }
    }
  }
}

// The same tile loop with the score contraction on v_mfma_f32_32x32x16_bf16 (BASELINE.json configs[4]: "MFMA bf16 ...
// attention"): T and G_u rows are rounded to bf16 when they are staged, products accumulate in fp32, tanh / max /
// argmax stay fp32.  LDS rows of 128 bf16 are padded to 136 (272 B = 17 slots of 16 B): conflict-free ds_read_b128.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void coattn_scores_bf16_kernel(ScoreParams p) {
  constexpr int LDB = 136;
  __shared__ __attribute__((aligned(16))) __bf16 Ts[64 * LDB];
  __shared__ __attribute__((aligned(16))) __bf16 Us[64 * LDB];
  __shared__ float xv[2][64]; __shared__ int xi[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wu = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.y, blk = blockIdx.x;
  const int SL = p.SL;
  const int j0 = blk * 64;
  const float* Tb = p.T + (long)b * SL * D;
  const float* Ub = p.Gu + (long)b * SL * D;
  auto stage = [&](const float* src, int row0, __bf16* dst) {
    float4 v[8];                                        // every load in flight before the first use (clamped rows)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      const int j = e >> 5, c4 = (e & 31) * 4;
      v[i] = *reinterpret_cast<const float4*>(src + (long)min(row0 + j, SL - 1) * D + c4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      const int j = e >> 5, c4 = (e & 31) * 4;
      const bool ok = row0 + j < SL;
      bf16x4_t o;
      o[0] = (__bf16)(ok ? v[i].x : 0.f); o[1] = (__bf16)(ok ? v[i].y : 0.f);
      o[2] = (__bf16)(ok ? v[i].z : 0.f); o[3] = (__bf16)(ok ? v[i].w : 0.f);
      *reinterpret_cast<bf16x4_t*>(dst + j * LDB + c4) = o;
    }
  };
  stage(Tb, j0, Ts);
  float rbest[16]; int ridx[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { rbest[r] = -INFINITY; ridx[r] = 0x7fffffff; }
  const int ntile = (SL + 63) / 64;
  const int tper = (ntile + p.ksplit - 1) / p.ksplit;
  const int ut_end = min(ntile, ((int)blockIdx.z + 1) * tper);
  for (int ut = blockIdx.z * tper; ut < ut_end; ++ut) {
    const int k0 = ut * 64;
    __syncthreads();
    stage(Ub, k0, Us);
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(Ts + (wi * 32 + l31) * LDB + ks * 16 + 8 * kh);
      const bf16x8_t u = *reinterpret_cast<const bf16x8_t*>(Us + (wu * 32 + l31) * LDB + ks * 16 + 8 * kh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, u, acc, 0, 0, 0);
    }
    const int kcol = k0 + wu * 32 + l31;
    const bool kvalid = kcol < SL;
    float cbest = -INFINITY; int cidx = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + wi * 32 + mfma_row(r, lane);
      const float a = tanhf(acc[r]);
      if (kvalid && j < SL) {
        better_first(rbest[r], ridx[r], a, kcol);
        better_first(cbest, cidx, a, j);
      }
    }
    {
      const float ov = __shfl_xor(cbest, 32, 64);
      const int oi = __shfl_xor(cidx, 32, 64);
      better_first(cbest, cidx, ov, oi);
    }
    if (wi == 1 && kh == 0) { xv[wu][l31] = cbest; xi[wu][l31] = cidx; }
    __syncthreads();
    if (wi == 0 && kh == 0) {
      better_first(cbest, cidx, xv[wu][l31], xi[wu][l31]);
      if (kvalid) {
        const long o = ((long)b * p.nblk + blk) * SL + kcol;
        p.colmax_part[o] = cbest;
        p.argcol_part[o] = cidx;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = rbest[r]; int i = ridx[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(i, o, 64);
      better_first(v, i, ov, oi);
    }
    rbest[r] = v; ridx[r] = i;
  }
  if (wu == 1 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = wi * 32 + mfma_row(r, lane);
      xv[0][jr] = rbest[r]; xi[0][jr] = ridx[r];
    }
  }
  __syncthreads();
  if (wu == 0 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = wi * 32 + mfma_row(r, lane);
      float v = rbest[r]; int i = ridx[r];
      better_first(v, i, xv[0][jr], xi[0][jr]);
      if (j0 + jr < SL) {
        const long o = ((long)b * p.ksplit + blockIdx.z) * SL + j0 + jr;
        p.rowmax_part[o] = v;
        p.argrow_part[o] = i;
      }
    }
  }
}

__device__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}
__device__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, red[w]);
  return s;
}

struct FinishParams {
  const float* Gu; const float* Gi;
  const float* colmax_part; const int* argcol_part; int nblk;
  const float* rowmax_part; const int* argrow_part; int ksplit;
  float* rowmax; int* argrow;
  float* colmax; int* argcol;
  float* soft_u; float* soft_i;
  float* atte_u; long ld_u;  // atte_u[b*ld_u + c]
  float* atte_i; long ld_i;
  int SL;
};

// 1024 threads per (sample, side): the weighted sums over the SL positions run as 8 chains of SL/8 terms per column.  Side 0 =
// the user's columns (colmax -> soft_u -> atte_u), side 1 = the item's rows; one workgroup each (grid (B, 2)): at batch 32 a grid
// of B workgroups doing both sides one after the other left 7/8 of the chip idle for 33 us.
__global__ __launch_bounds__(1024) void coattn_finish_kernel(FinishParams p) {
  extern __shared__ float sm[];  // s[SL]
  __shared__ float red[16];
  __shared__ float part[8][D];
  const int tid = threadIdx.x, b = blockIdx.x, side = blockIdx.y, SL = p.SL;
  const float* max_part = side ? p.rowmax_part : p.colmax_part;
  const int* arg_part = side ? p.argrow_part : p.argcol_part;
  const int nparts = side ? p.ksplit : p.nblk;     // (row parts: ascending column ranges, ties keep the first column)
  float* vmax = side ? p.rowmax : p.colmax;
  int* varg = side ? p.argrow : p.argcol;
  float* soft = side ? p.soft_i : p.soft_u;
  float* sv = sm;
  float mx = -INFINITY;
  for (int k = tid; k < SL; k += 1024) {
    float v = -INFINITY; int idx = 0x7fffffff;
    for (int q = 0; q < nparts; ++q) {
      const long o = ((long)b * nparts + q) * SL + k;
      better_first(v, idx, max_part[o], arg_part[o]);
    }
    vmax[(long)b * SL + k] = v; varg[(long)b * SL + k] = idx;
    sv[k] = v; mx = fmaxf(mx, v);
  }
  mx = block_max(mx, red);
  float z = 0.f;
  for (int k = tid; k < SL; k += 1024) {
    const float e = expf(sv[k] - mx);
    sv[k] = e; z += e;
  }
  z = block_sum(z, red);
  for (int k = tid; k < SL; k += 1024) {
    sv[k] /= z;
    soft[(long)b * SL + k] = sv[k];
  }
  __syncthreads();
  const int c = tid & 127, grp = tid >> 7;
  const float* G = (side ? p.Gi : p.Gu) + (long)b * SL * D;
  float a = 0.f;
#pragma unroll 4
  for (int k = grp; k < SL; k += 8) a += sv[k] * G[(long)k * D + c];
  part[grp][c] = a;
  __syncthreads();
  if (grp == 0) {
    const float v = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) + ((part[4][c] + part[5][c]) + (part[6][c] + part[7][c]));
    if (side == 0) p.atte_u[(long)b * p.ld_u + c] = v; else p.atte_i[(long)b * p.ld_i + c] = v;
  }
}

struct BwdPrepParams {
  const float* Gu; const float* Gi;
  const float* d_atte_u; long ld_du; const float* d_atte_i; long ld_di;   // [B][128] (strided rows)
  const float* d_soft_u; const float* d_soft_i;                           // [B][SL] or null
  const float* soft_u; const float* soft_i; const float* colmax; const float* rowmax;
  float* dSc; float* dSr;  // [B][SL]
  int SL;
};

// grid (B, 2): one workgroup per (sample, side), as coattn_finish_kernel
__global__ __launch_bounds__(1024) void coattn_bwd_prep_kernel(BwdPrepParams p) {
  extern __shared__ float sm[];  // ds[SL]
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, side = blockIdx.y, SL = p.SL;
  float* ds = sm;
  const float* Gs = (side ? p.Gi : p.Gu) + (long)b * SL * D;
  const float* d_atte = side ? p.d_atte_i + (long)b * p.ld_di : p.d_atte_u + (long)b * p.ld_du;
  const float* d_soft = side ? p.d_soft_i : p.d_soft_u;
  const float* soft = (side ? p.soft_i : p.soft_u) + (long)b * SL;
  const float* vmax = (side ? p.rowmax : p.colmax) + (long)b * SL;
  float* dS = (side ? p.dSr : p.dSc) + (long)b * SL;
  const float da0 = d_atte[lane], da1 = d_atte[64 + lane];
  for (int k = wave; k < SL; k += 32) {        // two positions per trip: both rows' loads in flight before either reduction
    const int k2 = k + 16;
    const bool has2 = k2 < SL;
    const float* g = Gs + (long)k * D;
    const float* g2 = Gs + (long)(has2 ? k2 : k) * D;
    const float a0 = g[lane], a1 = g[64 + lane], b0 = g2[lane], b1 = g2[64 + lane];
    const float v = wave_sum(a0 * da0 + a1 * da1);
    const float v2 = wave_sum(b0 * da0 + b1 * da1);
    if (lane == 0) {
      ds[k] = v + (d_soft ? d_soft[(long)b * SL + k] : 0.f);
      if (has2) ds[k2] = v2 + (d_soft ? d_soft[(long)b * SL + k2] : 0.f);
    }
  }
  __syncthreads();
  float dsum = 0.f;
  for (int k = tid; k < SL; k += 1024) dsum += soft[k] * ds[k];
  dsum = block_sum(dsum, red);
  for (int k = tid; k < SL; k += 1024) {
    const float m = vmax[k];
    dS[k] = soft[k] * (ds[k] - dsum) * (1.f - m * m);
  }
}

struct BwdRowsParams {
  const float* Gu; const float* T;
  const float* soft_u; const float* soft_i;
  const float* d_atte_u; long ld_du; const float* d_atte_i; long ld_di;
  const float* dSc; const float* dSr; const int* argcol; const int* argrow;
  float* dGu;  // [B][SL][128]  = soft_u d_atte_u + routed dS T
  float* dT;   // [B][SL][128]
  float* dGi;  // [B][SL][128]  initialised with soft_i d_atte_i (the GEMM dT M^T accumulates onto it)
  int SL; int accumulate;  // accumulate: add onto the caller's dGu / dGi instead of overwriting
};

__global__ __launch_bounds__(256) void coattn_bwd_rows_kernel(BwdRowsParams p) {
  extern __shared__ float sm[];  // dSc[SL], dSr[SL], argcol[SL], argrow[SL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, SL = p.SL;
  float* sc = sm; float* sr = sm + SL;
  int* ac = reinterpret_cast<int*>(sm + 2 * SL); int* ar = ac + SL;
  for (int k = tid; k < SL; k += 256) {
    const long o = (long)b * SL + k;
    sc[k] = p.dSc[o]; sr[k] = p.dSr[o]; ac[k] = p.argcol[o]; ar[k] = p.argrow[o];
  }
  __syncthreads();
  const float* Gu = p.Gu + (long)b * SL * D;
  const float* T = p.T + (long)b * SL * D;
  const float dau0 = p.d_atte_u[(long)b * p.ld_du + lane], dau1 = p.d_atte_u[(long)b * p.ld_du + 64 + lane];
  const float dai0 = p.d_atte_i[(long)b * p.ld_di + lane], dai1 = p.d_atte_i[(long)b * p.ld_di + 64 + lane];
  const int rows_per_block = 16;
  const int r0 = blockIdx.x * rows_per_block;
  for (int rr = wave; rr < rows_per_block; rr += 4) {
    const int row = r0 + rr;
    if (row >= SL) break;
    const long o = ((long)b * SL + row) * D;
    // ---- dG_u[row] (row is a u position k)
    {
      const float su = p.soft_u[(long)b * SL + row];
      const int j = ac[row];
      const float w = sc[row];
      float a0 = su * dau0 + w * T[(long)j * D + lane];
      float a1 = su * dau1 + w * T[(long)j * D + 64 + lane];
      for (int base = 0; base < SL; base += 64) {
        const int jj = base + lane;
        unsigned long long m = __ballot(jj < SL && ar[jj] == row);
        while (m) {
          const int q = base + __builtin_ctzll(m);
          m &= m - 1;
          const float wq = sr[q];
          a0 += wq * T[(long)q * D + lane];
          a1 += wq * T[(long)q * D + 64 + lane];
        }
      }
      if (p.accumulate) { a0 += p.dGu[o + lane]; a1 += p.dGu[o + 64 + lane]; }
      p.dGu[o + lane] = a0; p.dGu[o + 64 + lane] = a1;
    }
    // ---- dT[row] (row is an i position j)
    {
      const int k = ar[row];
      const float w = sr[row];
      float a0 = w * Gu[(long)k * D + lane];
      float a1 = w * Gu[(long)k * D + 64 + lane];
      for (int base = 0; base < SL; base += 64) {
        const int kk = base + lane;
        unsigned long long m = __ballot(kk < SL && ac[kk] == row);
        while (m) {
          const int q = base + __builtin_ctzll(m);
          m &= m - 1;
          const float wq = sc[q];
          a0 += wq * Gu[(long)q * D + lane];
          a1 += wq * Gu[(long)q * D + 64 + lane];
        }
      }
      p.dT[o + lane] = a0; p.dT[o + 64 + lane] = a1;
      const float si = p.soft_i[(long)b * SL + row];
      float g0 = si * dai0, g1 = si * dai1;
      if (p.accumulate) { g0 += p.dGi[o + lane]; g1 += p.dGi[o + 64 + lane]; }
      p.dGi[o + lane] = g0; p.dGi[o + 64 + lane] = g1;
    }
  }
}

}  // namespace

// workspace floats needed by forward: T [B*SL*128] + colmax_part [B*nblk*SL] + argcol_part (int) [B*nblk*SL]
size_t umpr_coattn_fwd_ws_bytes(int B, int SL) {
  const int nblk = cdiv(SL, 64);
  return ((size_t)B * nblk * SL * 4) * sizeof(float);
}

int umpr_coattn_fwd_impl(const float* Gu, const float* Gi, const float* M, int B, int SL, float* T, float* soft_u,
                         float* soft_i, float* atte_u, long ld_u, float* atte_i, long ld_i, float* colmax, int* argcol,
                         float* rowmax, int* argrow, float* ws, size_t ws_bytes, hipStream_t s, int bf16_scores) {
  UMPR_REQUIRE(B > 0 && SL > 0, "coattn: bad shape");
  UMPR_REQUIRE(ws_bytes >= umpr_coattn_fwd_ws_bytes(B, SL), "coattn: workspace too small");
  const int nblk = cdiv(SL, 64);
  UmprGemm g;
  g.A = Gi; g.lda = D; g.B = M; g.ldb = D; g.C = T; g.ldc = D; g.M = B * SL; g.N = D; g.K = D;
  if (int rc = umpr_gemm(g, s)) return rc;
  // one 64 x 64 score tile per workgroup: nblk^2 * B workgroups instead of nblk * B serial tile loops
  const int ksplit = nblk;
  float* cpart = ws;
  int* apart = reinterpret_cast<int*>(ws + (size_t)B * nblk * SL);
  float* rpart = ws + (size_t)2 * B * nblk * SL;
  int* arpart = reinterpret_cast<int*>(ws + (size_t)3 * B * nblk * SL);
  ScoreParams sp{T, Gu, rpart, arpart, cpart, apart, SL, nblk, ksplit};
  if (bf16_scores) coattn_scores_bf16_kernel<<<dim3(nblk, B, ksplit), 256, 0, s>>>(sp);
  else coattn_scores_kernel<<<dim3(nblk, B, ksplit), 256, 0, s>>>(sp);
  UMPR_LAUNCH_CHECK("coattn_scores");
  FinishParams fp{Gu, Gi, cpart, apart, nblk, rpart, arpart, ksplit, rowmax, argrow, colmax, argcol, soft_u, soft_i, atte_u, ld_u, atte_i, ld_i, SL};
  coattn_finish_kernel<<<dim3(B, 2), 1024, SL * sizeof(float), s>>>(fp);
  UMPR_LAUNCH_CHECK("coattn_finish");
  return 0;
}

// backward: needs dSc,dSr [B*SL] each + dT [B*SL*128] in ws; split-K slab for dM after that
size_t umpr_coattn_bwd_ws_bytes(int B, int SL) {
  return ((size_t)B * SL * 2 + (size_t)B * SL * D + (size_t)512 * D * D) * sizeof(float);
}

int umpr_coattn_bwd_impl(const float* Gu, const float* Gi, const float* M, const float* T, const float* soft_u,
                         const float* soft_i, const float* colmax, const int* argcol, const float* rowmax,
                         const int* argrow, const float* d_atte_u, long ld_du, const float* d_atte_i, long ld_di,
                         const float* d_soft_u, const float* d_soft_i, int B, int SL, float* dGu, float* dGi, float* dM,
                         int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(ws_bytes >= umpr_coattn_bwd_ws_bytes(B, SL), "coattn_bwd: workspace too small");
  float* dSc = ws; float* dSr = ws + (size_t)B * SL; float* dT = dSr + (size_t)B * SL;
  float* slab = dT + (size_t)B * SL * D;
  BwdPrepParams pp{Gu, Gi, d_atte_u, ld_du, d_atte_i, ld_di, d_soft_u, d_soft_i, soft_u, soft_i, colmax, rowmax, dSc, dSr, SL};
  coattn_bwd_prep_kernel<<<dim3(B, 2), 1024, SL * sizeof(float), s>>>(pp);
  UMPR_LAUNCH_CHECK("coattn_bwd_prep");
  BwdRowsParams rp{Gu, T, soft_u, soft_i, d_atte_u, ld_du, d_atte_i, ld_di, dSc, dSr, argcol, argrow, dGu, dT, dGi, SL, accumulate};
  coattn_bwd_rows_kernel<<<dim3(cdiv(SL, 16), B), 256, 4 * SL * sizeof(float), s>>>(rp);
  UMPR_LAUNCH_CHECK("coattn_bwd_rows");
  // dG_i += dT M^T
  UmprGemm g;
  g.A = dT; g.lda = D; g.B = M; g.ldb = D; g.transB = true; g.C = dGi; g.ldc = D; g.M = B * SL; g.N = D; g.K = D;
  g.accumulate = true;
  if (int rc = umpr_gemm(g, s)) return rc;
  // dM = G_i^T dT   (K = B*SL, split-K)
  UmprGemm h;
  h.A = Gi; h.lda = D; h.transA = true; h.B = dT; h.ldb = D; h.C = dM; h.ldc = D; h.M = D; h.N = D; h.K = B * SL;
  h.split_k = 0; h.ws = slab; h.ws_bytes = (size_t)512 * D * D * sizeof(float);
  return umpr_gemm(h, s);
}
