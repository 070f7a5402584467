// Batch-sized fully connected layers (the VGG16 classifier at <= 128 images per GPU) in exact fp32 on
// v_mfma_f32_32x32x2_f32.
//
// At M = 64 rows the generic LDS-tiled GEMM (gemm_f32.hip) streams the 411 MB fc1 weight matrix at 1.3 TB/s and 28 % of
// the fp32 MFMA rate: half of its 128-row tile is empty and every 16-deep k-tile pays a barrier.  Here nothing goes
// through LDS: a wave owns a strip of the output, keeps ALL batch rows of it in accumulators and streams its operands
// straight from memory into MFMA fragments ("GEMV-style" register streaming, cdna_hip_programming.md section 5).
//
//   fc_fwd   out[m][n] = sum_k x[m][k] W[n][k]          wave: all m x 32 columns n, a K-slice; split-K over waves / workgroups
//   fc_dx    dx[m][k]  = sum_n g[m][n] W[n][k]          wave: all m x 32 columns k, an N-slice; split-K likewise
//   fc_dw    dW[n][k]  = sum_m g[m][n] x[m][k]          wave: 32 rows n x 128 columns k, the whole (short) reduction over m
//
// The MFMA takes one k per lane half (A[i][k = l >> 5], B[k = l >> 5][j]).  A dot product does not care in which order
// its k are visited as long as both operands agree, so where an operand is k-contiguous in memory (x and W in fc_fwd, g in
// fc_dx) lane half h loads the float4 k = 8q + 4h .. 8q + 4h + 3 and the four MFMA steps of block q consume its four
// elements: 16-B loads instead of 4-B ones, no shuffles.
#include "umpr_common.h"
#include "umpr_internal.h"

namespace {

struct FcParams {
  const float* x;    // fwd: x [M][K];  dx: g [M][N];        dw: g [M][N]
  const float* w;    // fwd: W [N][K];  dx: W [N][K];        dw: x [M][K]
  float* out;        // fwd: partial slabs [split][M][N] (or out when split == 1); dx: likewise [split][M][K]; dw: dW [N][K]
  const float* bias; // fwd only, applied when split == 1
  int M, N, K;       // batch rows, output features, input features
  int split, per;    // reduction slices and their length (multiple of 8)
  int act;
  int accumulate;    // fwd, split == 1: out = act(out + x W^T + bias)
};

// out[m][n] over a K-slice.  TM row tiles of 32 batch rows.  Workgroup = 4 waves = 4 column groups of 32.
template <int TM>
__global__ __launch_bounds__(256) void fc_fwd_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 4 + wave) * 32 + c;          // this lane's output column
  const int split = blockIdx.y;
  const int k0 = split * p.per, k1 = min(p.K, k0 + p.per);
  const int nc = n < p.N ? n : p.N - 1;                    // columns past N compute garbage that is never stored
  const float* wrow = p.w + (long)nc * p.K + 4 * h;
  const float* xrow[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = i * 32 + c;
    mok[i] = m < p.M;
    xrow[i] = p.x + (long)(mok[i] ? m : 0) * p.K + 4 * h;
  }
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // K is a multiple of 8 for every classifier layer (25088, 4096); a ragged tail is handled by the generic GEMM instead.
  // (No partial-unroll pragma on these streaming loops: hipcc refused every one of them - 13 dropped requests in round 2's
  // build log - so the measured code never had them; `make check-passes` now fails the build on any dropped request.)
  for (int k = k0; k < k1; k += 8) {
    const float4 wv = *reinterpret_cast<const float4*>(wrow + k);
    float4 xv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      xv[i] = *reinterpret_cast<const float4*>(xrow[i] + k);
      if (!mok[i]) xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      acc[i] = mfma32(xv[i].x, wv.x, acc[i]);
      acc[i] = mfma32(xv[i].y, wv.y, acc[i]);
      acc[i] = mfma32(xv[i].z, wv.z, acc[i]);
      acc[i] = mfma32(xv[i].w, wv.w, acc[i]);
    }
  }
  if (n < p.N) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + mfma_row(r, lane);
        if (m < p.M) {
          float v = acc[i][r];
          if (p.split == 1) {
            if (p.bias) v += p.bias[n];
            if (p.accumulate) v += p.out[(long)m * p.N + n];
            v = apply_act(v, p.act);
          }
          p.out[((long)split * p.M + m) * p.N + n] = v;
        }
      }
  }
}

// out[m][n] over a K-slice, 32 reduction indices per trip: lane half h takes k = 32q + 16h .. + 15 as FOUR float4 (64 contiguous
// bytes of its W row and of each of its x rows; the two halves of a row together read one whole 128-B line per trip, where the
// 8-deep version above took 32 B of a line per trip and came back for the rest three trips later), and the loads of trip q + 1 are
// in flight while the 16 x TM MFMA steps of trip q run (two named register sets: the 8-deep loop waited for every load it issued -
// fc1 forward 294 us = 1.4 TB/s for a 411 MB weight stream, tools/bench_fc.py).  K and the slice length are multiples of 32.
template <int TM>
struct FcTrip { float4 w[4]; float4 x[TM][4]; };

template <int TM>
__global__ __launch_bounds__(256) void fc_fwd_k32_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 4 + wave) * 32 + c;
  const int split = blockIdx.y;
  const int k0 = split * p.per, k1 = min(p.K, k0 + p.per);
  const int nc = n < p.N ? n : p.N - 1;
  const float* wrow = p.w + (long)nc * p.K + 16 * h;
  const float* xrow[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = i * 32 + c;
    mok[i] = m < p.M;
    xrow[i] = p.x + (long)(mok[i] ? m : 0) * p.K + 16 * h;
  }
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  auto load = [&](int k, FcTrip<TM>& t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) t.w[j] = *reinterpret_cast<const float4*>(wrow + k + 4 * j);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t.x[i][j] = *reinterpret_cast<const float4*>(xrow[i] + k + 4 * j);
        if (!mok[i]) t.x[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
  };
  auto compute = [&](const FcTrip<TM>& t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = mfma32(t.x[i][j].x, t.w[j].x, acc[i]);
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = mfma32(t.x[i][j].y, t.w[j].y, acc[i]);
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = mfma32(t.x[i][j].z, t.w[j].z, acc[i]);
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = mfma32(t.x[i][j].w, t.w[j].w, acc[i]);
    }
  };
  FcTrip<TM> ta, tb;
  if (k0 < k1) load(k0, ta);
  for (int k = k0; k < k1; k += 64) {
    if (k + 32 < k1) load(k + 32, tb);
    compute(ta);
    if (k + 32 >= k1) break;
    if (k + 64 < k1) load(k + 64, ta);
    compute(tb);
  }
  if (n < p.N) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + mfma_row(r, lane);
        if (m < p.M) {
          float v = acc[i][r];
          if (p.split == 1) {
            if (p.bias) v += p.bias[n];
            if (p.accumulate) v += p.out[(long)m * p.N + n];
            v = apply_act(v, p.act);
          }
          p.out[((long)split * p.M + m) * p.N + n] = v;
        }
      }
  }
}

// dx[m][k] over an N-slice: A = g (k-contiguous in n: float4 with the permuted order), B[n][col k] = W[n][k] (rows n,
// lanes run along k: 128-B segments)
template <int TM>
__global__ __launch_bounds__(256) void fc_dx_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int kcol = (blockIdx.x * 4 + wave) * 32 + c;       // this lane's output column (input feature)
  const int split = blockIdx.y;
  const int n0 = split * p.per, n1 = min(p.N, n0 + p.per);
  const int kc = kcol < p.K ? kcol : p.K - 1;
  const float* wcol = p.w + kc;
  const float* grow[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = i * 32 + c;
    mok[i] = m < p.M;
    grow[i] = p.x + (long)(mok[i] ? m : 0) * p.N + 4 * h;
  }
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int n = n0; n < n1; n += 8) {
    float wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = wcol[(long)(n + 4 * h + j) * p.K];
    float4 gv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      gv[i] = *reinterpret_cast<const float4*>(grow[i] + n);
      if (!mok[i]) gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      acc[i] = mfma32(gv[i].x, wv[0], acc[i]);
      acc[i] = mfma32(gv[i].y, wv[1], acc[i]);
      acc[i] = mfma32(gv[i].z, wv[2], acc[i]);
      acc[i] = mfma32(gv[i].w, wv[3], acc[i]);
    }
  }
  if (kcol < p.K) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + mfma_row(r, lane);
        if (m < p.M) p.out[((long)split * p.M + m) * p.K + kcol] = acc[i][r];
      }
  }
}

// dW[n][k] = sum_m g[m][n] x[m][k]: wave tile 32 rows n x 128 columns k (4 accumulator tiles), reduction over the batch in
// natural order (k_mfma = 2s + h = batch row).  Both operands are dword loads whose lanes run along the contiguous index.
__global__ __launch_bounds__(256) void fc_dw_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int n = blockIdx.y * 32 + c;                        // A row
  const int kb = (blockIdx.x * 4 + wave) * 128;             // first output column of this wave
  if (kb >= p.K) return;
  const int nc = n < p.N ? n : p.N - 1;
  const int Mp = (p.M + 1) & ~1;                            // the MFMA consumes batch rows in pairs
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int kc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = kb + j * 32 + c; kc[j] = k < p.K ? k : p.K - 1; }
  for (int m2 = 0; m2 < Mp; m2 += 2) {
    const int m = m2 + h;
    const bool ok = m < p.M;
    const long mr = ok ? m : 0;
    float a = p.x[mr * p.N + nc];
    a = ok ? a : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = p.w[mr * p.K + kc[j]];
      acc[j] = mfma32(a, b, acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kb + j * 32 + c;
    if (k < p.K) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = blockIdx.y * 32 + mfma_row(r, lane);
        if (nn < p.N) p.out[(long)nn * p.K + k] = acc[j][r];
      }
    }
  }
}

// ---- float4 form of dW (K % 128 == 0).  A wave owns 128 output columns as FOUR MFMA column tiles interleaved column by column
// (tile t holds the columns kb + 4c + t): a lane's four B values of a batch row are ONE float4 of that row of x - 512 contiguous
// bytes per row and half wave instead of 128 - and its four results of an accumulator row leave as one float4 store.  fc1
// (64 x 4096 x 25088) in the training step, rocprofv3 medians: 215 -> 178 us.  (The same interleave for dx measured SLOWER -
// 252 -> 285 us, the small layers 33 -> 58-68 us: 128 accumulators per row tile leave too few waves to hide its loads - and was
// removed again; profiles/r03_e_c21_ab.txt.)
__global__ __launch_bounds__(256) void fc_dw_v4_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int n = blockIdx.y * 32 + c;                        // A row
  const int kb = (blockIdx.x * 4 + wave) * 128;
  if (kb >= p.K) return;
  const int nc = n < p.N ? n : p.N - 1;
  const int Mp = (p.M + 1) & ~1;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* xcol = p.w + kb + 4 * c;
  for (int m2 = 0; m2 < Mp; m2 += 2) {
    const int m = m2 + h;
    const bool ok = m < p.M;
    const long mr = ok ? m : 0;
    float a = p.x[mr * p.N + nc];
    float4 b = *reinterpret_cast<const float4*>(xcol + mr * p.K);
    if (!ok) { a = 0.f; b = make_float4(0.f, 0.f, 0.f, 0.f); }
    acc[0] = mfma32(a, b.x, acc[0]);
    acc[1] = mfma32(a, b.y, acc[1]);
    acc[2] = mfma32(a, b.z, acc[2]);
    acc[3] = mfma32(a, b.w, acc[3]);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int nn = blockIdx.y * 32 + mfma_row(r, lane);
    if (nn < p.N)
      *reinterpret_cast<float4*>(p.out + (long)nn * p.K + kb + 4 * c) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same three products with bf16 operands on v_mfma_f32_32x32x16_bf16 (BASELINE.json configs[4]: mixed precision).
// Operands stay fp32 in memory (master weights, fp32 activations) and are rounded to bf16 in registers on their way into
// the MFMA; accumulation is fp32.  At 1/16 of the fp32 MFMA cycles the kernels are bound by the 411 MB weight stream
// (fc1) instead of the matrix pipe.  Lane half h takes the eight reduction indices 16q + 8h .. 16q + 8h + 7.
typedef __bf16 fc_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ fc_bf16x8 to_bf16x8(const float4& a, const float4& b) {
  fc_bf16x8 r;
  r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
  r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
  return r;
}
__device__ __forceinline__ f32x16 mfma_b16(fc_bf16x8 a, fc_bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// out[m][n] over a K-slice (K % 16 == 0).  A wave owns 64 output columns (two B tiles): x is re-read half as often.
template <int TM>
__global__ __launch_bounds__(256) void fc_fwd_b16_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int nb = (blockIdx.x * 4 + wave) * 64;
  if (nb >= p.N) return;
  const int split = blockIdx.y;
  const int k0 = split * p.per, k1 = min(p.K, k0 + p.per);
  const float* wrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nb + j * 32 + c;
    wrow[j] = p.w + (long)(n < p.N ? n : p.N - 1) * p.K + 8 * h;   // columns past N compute garbage that is never stored
  }
  const float* xrow[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = i * 32 + c;
    mok[i] = m < p.M;
    xrow[i] = p.x + (long)(mok[i] ? m : 0) * p.K + 8 * h;
  }
  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int k = k0; k < k1; k += 16) {
    fc_bf16x8 wv[2], xv[TM];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      wv[j] = to_bf16x8(*reinterpret_cast<const float4*>(wrow[j] + k), *reinterpret_cast<const float4*>(wrow[j] + k + 4));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float4 a = *reinterpret_cast<const float4*>(xrow[i] + k), b = *reinterpret_cast<const float4*>(xrow[i] + k + 4);
      if (!mok[i]) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
      xv[i] = to_bf16x8(a, b);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma_b16(xv[i], wv[j], acc[i][j]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nb + j * 32 + c;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + mfma_row(r, lane);
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.split == 1) {
            if (p.bias) v += p.bias[n];
            if (p.accumulate) v += p.out[(long)m * p.N + n];
            v = apply_act(v, p.act);
          }
          p.out[((long)split * p.M + m) * p.N + n] = v;
        }
      }
  }
}

// dx[m][k] over an N-slice (slices are multiples of 16; N itself only of 8: a half-empty last step is zero-filled)
template <int TM>
__global__ __launch_bounds__(256) void fc_dx_b16_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int kcol = (blockIdx.x * 4 + wave) * 32 + c;       // this lane's output column (input feature)
  const int split = blockIdx.y;
  const int n0 = split * p.per, n1 = min(p.N, n0 + p.per);
  const int kc = kcol < p.K ? kcol : p.K - 1;
  const float* wcol = p.w + kc;
  const float* grow[TM];
  bool mok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = i * 32 + c;
    mok[i] = m < p.M;
    grow[i] = p.x + (long)(mok[i] ? m : 0) * p.N;
  }
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int n = n0; n < n1; n += 16) {
    const bool nok = n + 8 * h < n1;                       // this lane half's eight n exist (n1 is a multiple of 8)
    const int nn = nok ? n + 8 * h : n0;
    float wf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = wcol[(long)(nn + j) * p.K];
    fc_bf16x8 wv;
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = (__bf16)(nok ? wf[j] : 0.f);
    fc_bf16x8 gv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float4 a = *reinterpret_cast<const float4*>(grow[i] + nn), b = *reinterpret_cast<const float4*>(grow[i] + nn + 4);
      if (!mok[i] || !nok) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
      gv[i] = to_bf16x8(a, b);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = mfma_b16(gv[i], wv, acc[i]);
  }
  if (kcol < p.K) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = i * 32 + mfma_row(r, lane);
        if (m < p.M) p.out[((long)split * p.M + m) * p.K + kcol] = acc[i][r];
      }
  }
}

// dW[n][k] = sum_m g[m][n] x[m][k]: wave tile 32 rows n x 128 columns k, batch rows in steps of 16
__global__ __launch_bounds__(256) void fc_dw_b16_kernel(FcParams p) {
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int n = blockIdx.y * 32 + c;                        // A row
  const int kb = (blockIdx.x * 4 + wave) * 128;             // first output column of this wave
  if (kb >= p.K) return;
  const int nc = n < p.N ? n : p.N - 1;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int kc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int k = kb + j * 32 + c; kc[j] = k < p.K ? k : p.K - 1; }
  for (int m0 = 0; m0 < p.M; m0 += 16) {
    float af[8], bf[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = m0 + 8 * h + e;
      const long mr = m < p.M ? m : 0;
      af[e] = p.x[mr * p.N + nc];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j][e] = p.w[mr * p.K + kc[j]];
    }
    fc_bf16x8 a, b[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = m0 + 8 * h + e < p.M;
      a[e] = (__bf16)(ok ? af[e] : 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j][e] = (__bf16)(ok ? bf[j][e] : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mfma_b16(a, b[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = kb + j * 32 + c;
    if (k < p.K) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = blockIdx.y * 32 + mfma_row(r, lane);
        if (nn < p.N) p.out[(long)nn * p.K + k] = acc[j][r];
      }
    }
  }
}

// out[m][c] = act(sum over slabs + bias): four interleaved chains in a fixed order
__global__ void fc_reduce_kernel(const float* __restrict__ ws, int splits, long total, int cols, float* __restrict__ out,
                                 const float* __restrict__ bias, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int s = 0;
    for (; s + 3 < splits; s += 4) {
      v0 += ws[(long)s * total + i]; v1 += ws[(long)(s + 1) * total + i];
      v2 += ws[(long)(s + 2) * total + i]; v3 += ws[(long)(s + 3) * total + i];
    }
    for (; s < splits; ++s) v0 += ws[(long)s * total + i];
    float v = (v0 + v1) + (v2 + v3);
    if (bias) v += bias[i % cols];
    out[i] = apply_act(v, act);
  }
}

int pick_split(long strips, int red, int target_waves) {
  // reduction slices so that about target_waves waves exist; each slice a multiple of 8 and at least 256 deep
  int split = (int)((target_waves + strips - 1) / strips);
  const int maxs = red / 256;
  if (split > maxs) split = maxs;
  if (split < 1) split = 1;
  return split;
}

}  // namespace

bool umpr_fc_small_ok(int M, int N, int K) { return M >= 1 && M <= 128 && (K % 8) == 0 && (N % 8) == 0 && N >= 32 && K >= 32; }

size_t umpr_fc_small_ws_bytes(int M, int N, int K) {
  const size_t a = (size_t)64 * M * N, b = (size_t)64 * M * K;
  return (a > b ? a : b) * sizeof(float);
}

// out [M][N] = act(x [M][K] W[N][K]^T + bias)
int umpr_fc_small_fwd(const float* x, const float* W, const float* bias, float* out, int M, int N, int K, int act,
                      float* ws, size_t ws_bytes, hipStream_t s, int accumulate, int bf16) {
  UMPR_REQUIRE(umpr_fc_small_ok(M, N, K), "fc_small_fwd: unsupported shape %d x %d x %d", M, N, K);
  if (bf16 && (K % 16) == 0) {
    const int strips = cdiv(N, 64);
    int split = accumulate ? 1 : pick_split(strips, K, 2048);
    while (split > 1 && (size_t)split * M * N * sizeof(float) > ws_bytes) --split;
    FcParams p{x, W, split > 1 ? ws : out, bias, M, N, K, split, cdiv(cdiv(K, split), 16) * 16, act, accumulate};
    p.split = cdiv(K, p.per);
    if (p.split == 1) p.out = out;
    dim3 grid(cdiv(strips, 4), p.split);
    UmprProfScope prof(UMPR_K_GEMM, 2.0 * M * N * K, s);
    if (M <= 32) fc_fwd_b16_kernel<1><<<grid, 256, 0, s>>>(p);
    else if (M <= 64) fc_fwd_b16_kernel<2><<<grid, 256, 0, s>>>(p);
    else fc_fwd_b16_kernel<4><<<grid, 256, 0, s>>>(p);
    UMPR_LAUNCH_CHECK("fc_fwd_b16");
    if (p.split > 1) {
      const long total = (long)M * N;
      fc_reduce_kernel<<<cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256), 256, 0, s>>>(ws, p.split, total, N, out, bias, act);
      UMPR_LAUNCH_CHECK("fc_reduce");
    }
    return 0;
  }
  const int strips = cdiv(N, 32);
  static const bool k32_on = umpr_env_on("UMPR_FC_K32");       // 0: the 8-deep loops (A/B runs)
  const bool k32 = k32_on && (K % 32) == 0;
  const int gran = k32 ? 32 : 8;
  // the 32-deep kernel is fastest with ONE wave per SIMD (fc1 forward in the step: 241 us 8-deep / 213 at 4096 waves / 207 at 2048 / 186 at
  // 1024; profiles/r03_e_c21_ab.txt) - what is left is the texture path: a float4 load whose 64 lanes sit on 32 different rows
  static const int k32_waves = umpr_env_int("UMPR_FC_K32_WAVES", 1024);
  int split = accumulate ? 1 : pick_split(strips, K, k32 ? k32_waves : 4096);
  while (split > 1 && (size_t)split * M * N * sizeof(float) > ws_bytes) --split;
  FcParams p{x, W, split > 1 ? ws : out, bias, M, N, K, split, cdiv(cdiv(K, split), gran) * gran, act, accumulate};
  p.split = cdiv(K, p.per);
  if (p.split == 1) p.out = out;
  dim3 grid(cdiv(strips, 4), p.split);
  UmprProfScope prof(UMPR_K_GEMM, 2.0 * M * N * K, s);
  if (k32) {
    if (M <= 32) fc_fwd_k32_kernel<1><<<grid, 256, 0, s>>>(p);
    else if (M <= 64) fc_fwd_k32_kernel<2><<<grid, 256, 0, s>>>(p);
    else fc_fwd_k32_kernel<4><<<grid, 256, 0, s>>>(p);
  } else if (M <= 32) fc_fwd_kernel<1><<<grid, 256, 0, s>>>(p);
  else if (M <= 64) fc_fwd_kernel<2><<<grid, 256, 0, s>>>(p);
  else fc_fwd_kernel<4><<<grid, 256, 0, s>>>(p);
  UMPR_LAUNCH_CHECK("fc_fwd");
  if (p.split > 1) {
    const long total = (long)M * N;
    fc_reduce_kernel<<<cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256), 256, 0, s>>>(ws, p.split, total, N, out, bias, act);
    UMPR_LAUNCH_CHECK("fc_reduce");
  }
  return 0;
}

// dx [M][K] = g [M][N] W [N][K]
int umpr_fc_small_dx(const float* g, const float* W, float* dx, int M, int N, int K, float* ws, size_t ws_bytes,
                     hipStream_t s, int bf16) {
  UMPR_REQUIRE(umpr_fc_small_ok(M, N, K), "fc_small_dx: unsupported shape %d x %d x %d", M, N, K);
  const int strips = cdiv(K, 32);
  int split = pick_split(strips, N, 4096);
  while (split > 1 && (size_t)split * M * K * sizeof(float) > ws_bytes) --split;
  const int gran = bf16 ? 16 : 8;
  FcParams p{g, W, split > 1 ? ws : dx, nullptr, M, N, K, split, cdiv(cdiv(N, split), gran) * gran, 0};
  p.split = cdiv(N, p.per);
  if (p.split == 1) p.out = dx;
  dim3 grid(cdiv(strips, 4), p.split);
  UmprProfScope prof(UMPR_K_GEMM, 2.0 * M * N * K, s);
  if (bf16) {
    if (M <= 32) fc_dx_b16_kernel<1><<<grid, 256, 0, s>>>(p);
    else if (M <= 64) fc_dx_b16_kernel<2><<<grid, 256, 0, s>>>(p);
    else fc_dx_b16_kernel<4><<<grid, 256, 0, s>>>(p);
  } else if (M <= 32) fc_dx_kernel<1><<<grid, 256, 0, s>>>(p);
  else if (M <= 64) fc_dx_kernel<2><<<grid, 256, 0, s>>>(p);
  else fc_dx_kernel<4><<<grid, 256, 0, s>>>(p);
  UMPR_LAUNCH_CHECK("fc_dx");
  if (p.split > 1) {
    const long total = (long)M * K;
    fc_reduce_kernel<<<cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256), 256, 0, s>>>(ws, p.split, total, K, dx, nullptr, 0);
    UMPR_LAUNCH_CHECK("fc_reduce");
  }
  return 0;
}

// dW [N][K] = g [M][N]^T x [M][K]   (overwrites)
int umpr_fc_small_dw(const float* g, const float* x, float* dW, int M, int N, int K, hipStream_t s, int bf16) {
  UMPR_REQUIRE(umpr_fc_small_ok(M, N, K), "fc_small_dw: unsupported shape %d x %d x %d", M, N, K);
  FcParams p{g, x, dW, nullptr, M, N, K, 1, 0, 0};
  dim3 grid(cdiv(cdiv(K, 128), 4), cdiv(N, 32));
  UmprProfScope prof(UMPR_K_GEMM, 2.0 * M * N * K, s);
  static const bool v4_on = umpr_env_on("UMPR_FC_V4");        // 0: the dword form of dW (A/B runs)
  if (bf16) fc_dw_b16_kernel<<<grid, 256, 0, s>>>(p);
  else if (v4_on && (K % 128) == 0) fc_dw_v4_kernel<<<grid, 256, 0, s>>>(p);
  else fc_dw_kernel<<<grid, 256, 0, s>>>(p);
  UMPR_LAUNCH_CHECK("fc_dw");
  return 0;
}
