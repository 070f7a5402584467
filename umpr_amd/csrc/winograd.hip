// Winograd path for the deep VGG16 layers (56x56, 28x28, 14x14 maps; in backward also conv2_2 at 112x112), fp32.
//
// Forward, F(2x2,3x3):   Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 2x2 output tile / 4x4 input tile
//   16 independent GEMMs  M[xi][m][t] = sum_c U[xi][m][c] * V[xi][c][t]  (t = tile index over the batch) replace the
//   9-tap implicit GEMM: 2.25x fewer MFMA FLOPs.  Kernels per layer call:
//     wino_weights_kernel      : U = G g G^T (for dgrad on the flipped, channel-transposed kernel), LDS image order
//     wino_input[_pair]_kernel : x -> V   (HBM bound: reads |x|, writes 4|x|; zero-pads channels / tiles to the GEMM tile)
//     wino_gemm_dma_kernel     : batched GEMM on v_mfma_f32_32x32x2_f32, operands copied global -> LDS by LDS-DMA
//                                (wino_gemm_kernel: the same loop staged through registers, UMPR_WINO_DMA=0)
//     wino_output[_pair]_kernel: M -> y   with the fused epilogue (+bias, ReLU) or (dgrad) * [mask > 0]
// Data gradient, F(4x4,3x3) where the map is a multiple of 4 (wino4_* kernels, the same GEMM over 36 planes and a quarter of
//   the tiles: 4x fewer FLOPs, V / M 2.25x instead of 4x the activation), else F(2x2,3x3) as in forward.  The forward pass
//   keeps the 2x2 tile: the larger tile's rounding (5e-6 of max|y| instead of 3e-7) flips ReLU / pool decisions and moves
//   the early layers' gradients outside the parity bound; in backward it is a smooth perturbation (UMPR_WINO_F4, below).
// Weight gradient, F(3x3,2x2) / F(3x3,4x4): see the second half of this file (wino[4]_dy / wino_wgrad_gemm / finish).
// The 224x224 layers, conv2_1 and the forward pass of conv2_2 stay on the direct kernels of conv3x3.hip: with 64 channels
// on one side the batched GEMM is HBM-bound on V / M, and in forward the 4x transform traffic of the 2x2 tile costs more
// than the MFMA time it saves.
#include <type_traits>

#include "umpr_common.h"
#include "umpr_internal.h"

namespace {

constexpr int WK = 32;    // channels per GEMM stage
constexpr int WBM = 128;  // output-channel tile
constexpr int WBN = 128;  // tile-index tile
constexpr int WLDA = WBM + 4;

// U[xi][mt][s][k][m_local]  (mt = m / 128, s = c / 32, k = c % 32), zero padded in both m and c
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int M, int C, int CinW,
                                    int transposed) {
  const int MT = (M + WBM - 1) / WBM, S = (C + WK - 1) / WK;
  const long total = (long)MT * WBM * S * WK;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ml = (int)(i % WBM);
    long r = i / WBM;
    const int k = (int)(r % WK); r /= WK;
    const int s = (int)(r % S);
    const int mt = (int)(r / S);
    const int m = mt * WBM + ml, c = s * WK + k;
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.f;
      if (m < M && c < C) v = transposed ? w[((long)c * CinW + m) * 9 + 8 - t] : w[((long)m * CinW + c) * 9 + t];
      g[t] = v;
    }
    // Gg = G g  (4x3), then (G g) G^T (4x4);  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
    float gg[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      gg[0][j] = g[j];
      gg[1][j] = 0.5f * (g[j] + g[3 + j] + g[6 + j]);
      gg[2][j] = 0.5f * (g[j] - g[3 + j] + g[6 + j]);
      gg[3][j] = g[6 + j];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = gg[a][0], u1 = 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]),
                  u2 = 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]), u3 = gg[a][2];
      const long per = total;
      U[(long)(a * 4 + 0) * per + i] = u0;
      U[(long)(a * 4 + 1) * per + i] = u1;
      U[(long)(a * 4 + 2) * per + i] = u2;
      U[(long)(a * 4 + 3) * per + i] = u3;
    }
  }
}

// V[xi][c][t] (c < Cpad = S * 32),  t = (n*TH + ty)*TW + tx,  d = x[n][c][2ty-1 .. 2ty+2][2tx-1 .. 2tx+2] (zero outside)
__global__ void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C, int Cpad, int H,
                                  int W, long Tpad, long Tw) {
  const int TH = H / 2, TW = W / 2;
  const long T = (long)N * TH * TW;
  const long total = (long)Cpad * Tw;   // channels / tiles rounded up to the GEMM tile: the padding is written as zeros
  const long per = (long)Cpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i % Tw;
    const int c = (int)(i / Tw);
    float* dst = V + (long)c * Tpad + t;
    if (t >= T || c >= C) {
#pragma unroll
      for (int a = 0; a < 16; ++a) dst[(long)a * per] = 0.f;
      continue;
    }
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = x + ((long)n * C + c) * H * W;
    float d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = 2 * ty - 1 + a;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = 2 * tx - 1 + b;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float v = src[ok ? yy * W + xx : 0];
        d[a][b] = ok ? v : 0.f;
      }
    }
    // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
    float bd[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      bd[0][b] = d[0][b] - d[2][b];
      bd[1][b] = d[1][b] + d[2][b];
      bd[2][b] = d[2][b] - d[1][b];
      bd[3][b] = d[1][b] - d[3][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dst[(long)(a * 4 + 0) * per] = bd[a][0] - bd[a][2];
      dst[(long)(a * 4 + 1) * per] = bd[a][1] + bd[a][2];
      dst[(long)(a * 4 + 2) * per] = bd[a][2] - bd[a][1];
      dst[(long)(a * 4 + 3) * per] = bd[a][1] - bd[a][3];
    }
  }
}

// Same transform, two horizontally adjacent tiles per thread (even tile rows only: 56x56 and 28x28 maps): the six
// input columns the pair touches are loaded as scalar | float2 | float2 | scalar per row (half the load and store
// instructions of the one-tile kernel), and the results leave as float2.
__global__ void wino_input_pair_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C, int Cpad,
                                       int H, int W, long Tpad, long Tw) {
  const int TH = H / 2, TW = W / 2;
  const long T = (long)N * TH * TW;
  const long Tw2 = Tw / 2;
  const long total = (long)Cpad * Tw2;
  const long per = (long)Cpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = 2 * (i % Tw2);
    const int c = (int)(i / Tw2);
    float* dst = V + (long)c * Tpad + t;
    if (t >= T || c >= C) {
#pragma unroll
      for (int a = 0; a < 16; ++a) *reinterpret_cast<float2*>(dst + (long)a * per) = make_float2(0.f, 0.f);
      continue;
    }
    const int tx = (int)(t % TW);   // even; tx + 1 is in the same tile row
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = x + ((long)n * C + c) * H * W;
    const int x0 = 2 * tx;
    const bool okl = x0 > 0, okr = x0 + 4 < W;
    float d[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = 2 * ty - 1 + a;
      const bool oky = yy >= 0 && yy < H;
      const float* row = src + (oky ? yy * W + x0 : x0);
      const float l = row[okl ? -1 : 0];
      const float2 m0 = *reinterpret_cast<const float2*>(row);
      const float2 m1 = *reinterpret_cast<const float2*>(row + 2);
      const float rr = row[okr ? 4 : 0];
      d[a][0] = oky && okl ? l : 0.f;
      d[a][1] = oky ? m0.x : 0.f; d[a][2] = oky ? m0.y : 0.f;
      d[a][3] = oky ? m1.x : 0.f; d[a][4] = oky ? m1.y : 0.f;
      d[a][5] = oky && okr ? rr : 0.f;
    }
    float bd[4][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      bd[0][b] = d[0][b] - d[2][b];
      bd[1][b] = d[1][b] + d[2][b];
      bd[2][b] = d[2][b] - d[1][b];
      bd[3][b] = d[1][b] - d[3][b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // tile 0 uses columns 0..3, tile 1 columns 2..5
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 0) * per) = make_float2(bd[a][0] - bd[a][2], bd[a][2] - bd[a][4]);
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 1) * per) = make_float2(bd[a][1] + bd[a][2], bd[a][3] + bd[a][4]);
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 2) * per) = make_float2(bd[a][2] - bd[a][1], bd[a][4] - bd[a][3]);
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 3) * per) = make_float2(bd[a][1] - bd[a][3], bd[a][3] - bd[a][5]);
    }
  }
}

// y[n][m][2ty+i][2tx+j] = epilogue( (A^T M A)[i][j] ),  A^T = [1 1 1 0; 0 1 -1 -1]
__global__ void wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias,
                                   const float* __restrict__ mask, float* __restrict__ y, int N, int Mch, int H,
                                   int W, long Tpad, int Mpad, int relu) {
  const int TH = H / 2, TW = W / 2;
  const long T = (long)N * TH * TW;
  const long total = (long)Mch * T;
  const long per = (long)Mpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i % T;
    const int m = (int)(i / T);
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = Mx + (long)m * Tpad + t;
    float mm[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) mm[a][b] = src[(long)(a * 4 + b) * per];
    float am[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      am[0][b] = mm[0][b] + mm[1][b] + mm[2][b];
      am[1][b] = mm[1][b] - mm[2][b] - mm[3][b];
    }
    float o[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      o[a][0] = am[a][0] + am[a][1] + am[a][2];
      o[a][1] = am[a][1] - am[a][2] - am[a][3];
    }
    const float bv = bias ? bias[m] : 0.f;
    const long ob = (((long)n * Mch + m) * H + 2 * ty) * W + 2 * tx;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float v0 = o[a][0] + bv, v1 = o[a][1] + bv;
      if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      const long oo = ob + (long)a * W;
      if (mask) {
        const float2 mk = *reinterpret_cast<const float2*>(mask + oo);
        v0 = mk.x > 0.f ? v0 : 0.f; v1 = mk.y > 0.f ? v1 : 0.f;
      }
      *reinterpret_cast<float2*>(y + oo) = make_float2(v0, v1);
    }
  }
}

// Two horizontally adjacent tiles per thread (even tile rows): float2 loads of M, float4 stores of y.
__global__ void wino_output_pair_kernel(const float* __restrict__ Mx, const float* __restrict__ bias,
                                        const float* __restrict__ mask, float* __restrict__ y, int N, int Mch, int H,
                                        int W, long Tpad, int Mpad, int relu) {
  const int TH = H / 2, TW = W / 2;
  const long T2 = (long)N * TH * TW / 2;
  const long total = (long)Mch * T2;
  const long per = (long)Mpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = 2 * (i % T2);
    const int m = (int)(i / T2);
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = Mx + (long)m * Tpad + t;
    float2 mm[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) mm[a][b] = *reinterpret_cast<const float2*>(src + (long)(a * 4 + b) * per);
    float o[2][4];   // [output row][4 consecutive output columns: tile 0 cols 0,1 | tile 1 cols 0,1]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float am[2][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float m0 = q ? mm[0][b].y : mm[0][b].x, m1 = q ? mm[1][b].y : mm[1][b].x;
        const float m2 = q ? mm[2][b].y : mm[2][b].x, m3 = q ? mm[3][b].y : mm[3][b].x;
        am[0][b] = m0 + m1 + m2;
        am[1][b] = m1 - m2 - m3;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        o[a][2 * q + 0] = am[a][0] + am[a][1] + am[a][2];
        o[a][2 * q + 1] = am[a][1] - am[a][2] - am[a][3];
      }
    }
    const float bv = bias ? bias[m] : 0.f;
    const long ob = (((long)n * Mch + m) * H + 2 * ty) * W + 2 * tx;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = o[a][j] + bv;
        if (relu) v[j] = fmaxf(v[j], 0.f);
      }
      const long oo = ob + (long)a * W;
      if (mask) {
        const float4 mk = *reinterpret_cast<const float4*>(mask + oo);
        v[0] = mk.x > 0.f ? v[0] : 0.f; v[1] = mk.y > 0.f ? v[1] : 0.f;
        v[2] = mk.z > 0.f ? v[2] : 0.f; v[3] = mk.w > 0.f ? v[3] : 0.f;
      }
      *reinterpret_cast<float4*>(y + oo) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// F(4x4,3x3): 6x6 input tile -> 4x4 outputs, 36 planes.  4x fewer multiplies than the direct convolution (F(2x2,3x3):
// 2.25x) and V / M are 2.25x the activation they transform instead of 4x - both the GEMM and the HBM-bound transforms
// shrink.  Used where 36 planes over ceil(H/4) x ceil(W/4) tiles are less work than 16 planes over (H/2) x (W/2).
// Interpolation points (0, +-a, +-b, inf), Cook-Toom construction (tools/wino_points.py):
//   A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1]
//   B^T = [a^2b^2 0 -(a^2+b^2) 0 1 0; 0 -ab^2 -b^2 a 1 0; 0 ab^2 -b^2 -a 1 0; 0 -a^2b -a^2 b 1 0; 0 a^2b -a^2 -b 1 0;
//          0 a^2b^2 0 -(a^2+b^2) 0 1]
//   G   = [1/(a^2b^2) 0 0; ca(1 a a^2); ca(1 -a a^2); cb(1 b b^2); cb(1 -b b^2); 0 0 1],  ca = 1/(2a^2(a^2-b^2)), cb = 1/(2b^2(b^2-a^2))
// The price of the larger tile is rounding, and the points set it.  The textbook points (a, b) = (1, 2) (constants 4, 5, 8,
// 1/24) put the fp32 result 3-4e-6 of max|y| (rms 1.1e-6) from the float64 convolution where the direct kernel and
// F(2x2,3x3) are at 2e-7: enough to flip ~15x more ReLU / max-pool decisions than the direct kernel in a TRAINING forward
// pass, which moved the first layers' gradients outside the golden-fixture bound (round 2: forward stayed on the 2x2 tile).
// Round 3: (a, b) = (3/4, 3/2) - every constant of B^T and A^T is dyadic (exact in fp32), the Vandermonde rows are better
// balanced - measures max 0.8-1.2e-6, rms 5.6e-7 on the same experiment (tools/wino_points.py), and the CPU simulation of the
// golden-fixture criterion (tools/wino_flip_sim.py) puts it level with F(2x2,3x3) (first conv's gradient 1.2e-3 from the
// float64 run vs 1.1e-3; textbook points 2.8e-3).  UMPR_WINO_POINTS=0 selects the textbook points (A/B runs).
// ---------------------------------------------------------------------------------------------------------------------
struct Wino4C {
  float a, b, a2, b2, a3, b3;   // A^T
  float p, sm, ab2, a2b;        // B^T: p = a^2 b^2, sm = a^2 + b^2
  float g0, ca, cb;             // G: 1 / (a^2 b^2), ca, cb
};
static Wino4C wino4_consts() {
  static const int textbook = umpr_env_int("UMPR_WINO_POINTS", 1) == 0;
  const double a = textbook ? 1.0 : 0.75, b = textbook ? 2.0 : 1.5;
  Wino4C c;
  c.a = (float)a; c.b = (float)b; c.a2 = (float)(a * a); c.b2 = (float)(b * b); c.a3 = (float)(a * a * a); c.b3 = (float)(b * b * b);
  c.p = (float)(a * a * b * b); c.sm = (float)(a * a + b * b); c.ab2 = (float)(a * b * b); c.a2b = (float)(a * a * b);
  c.g0 = (float)(1.0 / (a * a * b * b));
  c.ca = (float)(1.0 / (2.0 * a * a * (a * a - b * b)));
  c.cb = (float)(1.0 / (2.0 * b * b * (b * b - a * a)));
  return c;
}
__device__ __forceinline__ void wino4_bt(const Wino4C& c, const float d0, const float d1, const float d2, const float d3,
                                         const float d4, const float d5, float* __restrict__ o) {
  const float ea = d4 - c.b2 * d2, oa = c.a * d3 - c.ab2 * d1;    // rows +-a: even and odd part
  const float eb = d4 - c.a2 * d2, ob = c.b * d3 - c.a2b * d1;    // rows +-b
  o[0] = c.p * d0 - c.sm * d2 + d4;
  o[1] = ea + oa;
  o[2] = ea - oa;
  o[3] = eb + ob;
  o[4] = eb - ob;
  o[5] = c.p * d1 - c.sm * d3 + d5;
}
// G g for one column of three filter taps
__device__ __forceinline__ void wino4_g(const Wino4C& c, const float g0, const float g1, const float g2, float* __restrict__ o) {
  const float ea = g0 + c.a2 * g2, eb = g0 + c.b2 * g2;
  o[0] = c.g0 * g0;
  o[1] = c.ca * (ea + c.a * g1);
  o[2] = c.ca * (ea - c.a * g1);
  o[3] = c.cb * (eb + c.b * g1);
  o[4] = c.cb * (eb - c.b * g1);
  o[5] = g2;
}
// A^T q for six plane values
__device__ __forceinline__ void wino4_at(const Wino4C& c, const float* __restrict__ q, float* __restrict__ y) {
  const float s12 = q[1] + q[2], d12 = q[1] - q[2], s34 = q[3] + q[4], d34 = q[3] - q[4];
  y[0] = q[0] + s12 + s34;
  y[1] = c.a * d12 + c.b * d34;
  y[2] = c.a2 * s12 + c.b2 * s34;
  y[3] = c.a3 * d12 + c.b3 * d34 + q[5];
}
// |A^T| q for six NON-NEGATIVE values: the magnitude that feeds each output (decision fix-up below)
__device__ __forceinline__ void wino4_at_abs(const Wino4C& c, const float* __restrict__ q, float* __restrict__ y) {
  const float s12 = q[1] + q[2], s34 = q[3] + q[4];
  y[0] = q[0] + s12 + s34;
  y[1] = c.a * s12 + c.b * s34;
  y[2] = c.a2 * s12 + c.b2 * s34;
  y[3] = c.a3 * s12 + c.b3 * s34 + q[5];
}

// U[xi][mt][s][k][m_local], xi = 6a + b: the image order of wino_weights_kernel with 36 planes
__global__ void wino4_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int M, int C, int CinW,
                                     int transposed, Wino4C wc) {
  const int MT = (M + WBM - 1) / WBM, S = (C + WK - 1) / WK;
  const long total = (long)MT * WBM * S * WK;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ml = (int)(i % WBM);
    long r = i / WBM;
    const int k = (int)(r % WK); r /= WK;
    const int s = (int)(r % S);
    const int mt = (int)(r / S);
    const int m = mt * WBM + ml, c = s * WK + k;
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.f;
      if (m < M && c < C) v = transposed ? w[((long)c * CinW + m) * 9 + 8 - t] : w[((long)m * CinW + c) * 9 + t];
      g[t] = v;
    }
    float gg[6][3];   // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float o[6];
      wino4_g(wc, g[j], g[3 + j], g[6 + j], o);
#pragma unroll
      for (int a = 0; a < 6; ++a) gg[a][j] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float o[6];
      wino4_g(wc, gg[a][0], gg[a][1], gg[a][2], o);
      float* dst = U + (long)(a * 6) * total + i;
#pragma unroll
      for (int b = 0; b < 6; ++b) dst[(long)b * total] = o[b];
    }
  }
}

// V[xi][c][t] (c < Cpad),  t = (n*TH + ty)*TW + tx over 4x4 output tiles,  d = x[n][c][4ty-1 .. 4ty+4][4tx-1 .. 4tx+4]
// One thread per (c, t): per input row one aligned float4 and the two halo scalars; 36 stores, each coalesced over t.
// EDGE: the map is not a multiple of 4 (14x14): tiles hang over the right / bottom border (zero there), element-wise loads.
// "The same input" for the tie predicates below: equal to 2^-16 of the input tensor's largest magnitude.  Exact equality would
// do for a photo's flat regions themselves, but not one layer later: the previous Winograd layers' outputs over a flat region are
// equal only up to their own position-dependent rounding (~2e-6 of THEIR tensor's maximum, whatever the channel's own level), and
// half the outputs of a constant image were listed (tools/fix_counts.py).  A pool decision between two outputs whose 3x3xC inputs
// agree to 1.5e-5 of the maximum moves a gradient by that much at most.  The maximum comes from wino_absmax_kernel (one pass over
// the input of the three pooled layers, ~0.1 ms per step; a maximum does not depend on the order of its atomics).
__device__ __forceinline__ bool wino_same(float u, float v, float tol) { return fabsf(u - v) <= tol; }

__global__ __launch_bounds__(256) void wino_absmax_kernel(const float* __restrict__ x, long n4, unsigned int* __restrict__ slot) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic per workgroup: 16 384 of them on one word took longer than reading the tensor
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.f) atomicMax(slot, __float_as_uint(m));   // non-negative floats order like their bit patterns
  }
}

// PF (layers a max-pool follows, training forward): also publishes, per tile, which of its four 2x2 pool windows have contenders
// with THE SAME inputs - bits[t], bit 2w = "some channel's 4x4 input patch of window w has a row that is not constant along x",
// bit 2w+1 = "... a column that is not constant along y" (w = 2wy + wx; the zero padding counts as input).  If every row of the
// patch is constant along x, horizontally adjacent outputs of the window see the same 3x3xC inputs and tie (exactly, for exactly
// equal inputs, in exact arithmetic); likewise columns / vertical neighbours; both / diagonal.  See the pool rule of wino4_output_kernel.
template <bool EDGE, bool PF>
__global__ __launch_bounds__(256) void wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int C,
                                                          int Cpad, int H, int W, long Tpad, long Tw, Wino4C wc,
                                                          unsigned int* __restrict__ zero, unsigned int* __restrict__ bits) {
  if (zero && blockIdx.x == 0 && threadIdx.x == 0) *zero = 0u;   // the fix-up list counter of this pass (see WinoFix)
  const float tol = PF ? 1.52587890625e-05f * __uint_as_float(bits[-1]) : 0.f;   // bits[-1]: max |x| (wino_absmax_kernel)
  const int TH = (H + 3) / 4, TW = (W + 3) / 4;
  const long T = (long)N * TH * TW;
  const long total = (long)Cpad * Tw;
  const long per = (long)Cpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i % Tw;
    const int c = (int)(i / Tw);
    float* dst = V + (long)c * Tpad + t;
    if (t >= T || c >= C) {
#pragma unroll
      for (int a = 0; a < 36; ++a) dst[(long)a * per] = 0.f;
      continue;
    }
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = x + ((long)n * C + c) * H * W;
    const int x0 = 4 * tx;
    const bool okl = x0 > 0, okr = x0 + 4 < W;
    float e[6][6];   // e[a] = (row a of d) B  - the horizontal transform, applied as each row arrives
    float prev[6];
    unsigned rowL = 0, rowR = 0, colEq[5];   // PF: row a constant over columns 0..3 / 2..5; colEq[a] bit b: d[a][b] == d[a+1][b]
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const int yy = 4 * ty - 1 + a;
      const bool oky = yy >= 0 && yy < H;
      const float* row = src + (oky ? yy * W : 0) + x0;
      float d[6];
      if (EDGE) {
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          const int xx = x0 - 1 + b;
          const bool ok = oky && xx >= 0 && xx < W;
          const float v = row[ok ? b - 1 : 0 - x0];   // invalid: element (yy or 0, 0) of the plane, always in range
          d[b] = ok ? v : 0.f;
        }
      } else {
        const float l = row[okl ? -1 : 0];
        const float4 m = *reinterpret_cast<const float4*>(row);
        const float rr = row[okr ? 4 : 0];
        d[0] = oky && okl ? l : 0.f; d[1] = oky ? m.x : 0.f; d[2] = oky ? m.y : 0.f; d[3] = oky ? m.z : 0.f;
        d[4] = oky ? m.w : 0.f; d[5] = oky && okr ? rr : 0.f;
      }
      wino4_bt(wc, d[0], d[1], d[2], d[3], d[4], d[5], e[a]);
      if (PF) {
        const bool mid = wino_same(d[2], d[3], tol);
        if (mid && wino_same(d[0], d[1], tol) && wino_same(d[1], d[2], tol)) rowL |= 1u << a;
        if (mid && wino_same(d[3], d[4], tol) && wino_same(d[4], d[5], tol)) rowR |= 1u << a;
        if (a > 0) {
          unsigned m6 = 0;
#pragma unroll
          for (int b = 0; b < 6; ++b) m6 |= (unsigned)wino_same(prev[b], d[b], tol) << b;
          colEq[a - 1] = m6;
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) prev[b] = d[b];
      }
    }
    if (PF) {
      const unsigned colT = colEq[0] & colEq[1] & colEq[2];   // bit b: column b constant over rows 0..3
      const unsigned colB = colEq[2] & colEq[3] & colEq[4];   // rows 2..5
      unsigned viol = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int wy = w >> 1, wx = w & 1;
        const unsigned rows = wx ? rowR : rowL, cols = wy ? colB : colT;
        if (((rows >> (2 * wy)) & 0xFu) != 0xFu) viol |= 1u << (2 * w);        // rows 2wy .. 2wy+3 of the window's patch
        if (((cols >> (2 * wx)) & 0xFu) != 0xFu) viol |= 2u << (2 * w);        // columns 2wx .. 2wx+3
      }
      if (viol & ~bits[t]) atomicOr(&bits[t], viol);   // textured tiles: set by their first channels, then only read
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float o[6];
      wino4_bt(wc, e[0][b], e[1][b], e[2][b], e[3][b], e[4][b], e[5][b], o);
#pragma unroll
      for (int a = 0; a < 6; ++a) dst[(long)(a * 6 + b) * per] = o[a];
    }
  }
}

// Decision fix-up of the TRAINING forward (round 3).  The 4x4 tile's fp32 result carries ~9e-7 (rms, of rms y) of rounding
// where the direct kernel carries 1.7e-7 (tools/conv_error.py).  As a perturbation of the values that is harmless; but every
// ReLU / max-pool DECISION of the forward pass is replayed by the backward pass, and one decision that lands on the other side
// of the float64 one moves the gradients of all earlier layers by ~1e-3 relative L2 (tools/relu_flip_experiment.py) - five
// times the noise means five times the flipped decisions, which is what kept the training forward on the 2x2 tile in round 2.
// So the decisions are taken at the direct kernel's accuracy instead: the output transform flags every output whose decision
// the tile's rounding could change - |y| < tau (ReLU) and, in a layer that a 2x2 max-pool follows, the outputs of a pool window
// whose leader is less than tau ahead of the runner-up - with tau = kappa * 2^-24 * S,  S = sum_ab |A^T_ia| |A^T_jb| |M_ab| the magnitude that fed that output (local and
// scale-free: no pass over the tensor, no global maximum), and appends its index to a list; wino_fixup_kernel then recomputes
// the listed outputs (a few per 100 000) as plain 9 C-term dot products, one wave each.
// Pool rule and exact ties: where the image is constant - a missing photo (all zeros, src/dataset.py:142-143), the white
// background of a product photo along the image border - neighbouring outputs tie EXACTLY in exact arithmetic, and the tile's
// position-dependent rounding separates them by noise: every such window would be listed (a quarter of a million outputs per
// constant image) although no choice among equals can move a gradient.  The input transform therefore publishes, per tile and
// window, whether the contenders' inputs are identical (wino4_input_kernel<.., PF>), and a window whose leader and runner-up are
// such equals is not listed.  The list order depends on the
// atomics, the values do not: every listed element is recomputed independently in a fixed summation order.
struct WinoFix {
  unsigned int* count;   // zeroed by the input-transform kernel of the same pass
  unsigned int* list;    // output element indices
  unsigned int cap;
  float kappa_eps;       // kappa * 2^-24
  const unsigned int* tie_bits;   // per tile, from wino4_input_kernel<.., PF>; NULL: no max-pool follows, ReLU rule only
};

// y[n][m][4ty+i][4tx+j] = epilogue( (A^T M A)[i][j] ): 36 loads coalesced over t, one float4 store per output row
template <bool EDGE, bool FIX>
__global__ __launch_bounds__(256) void wino4_output_kernel(const float* __restrict__ Mx, const float* __restrict__ bias,
                                                           const float* __restrict__ mask, float* __restrict__ y, int N,
                                                           int Mch, int H, int W, long Tpad, int Mpad, int relu, Wino4C wc,
                                                           WinoFix fx) {
  const int TH = (H + 3) / 4, TW = (W + 3) / 4;
  const long T = (long)N * TH * TW;
  const long total = (long)Mch * T;
  const long per = (long)Mpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i % T;
    const int m = (int)(i / T);
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = Mx + (long)m * Tpad + t;
    float am[4][6];   // A^T M, accumulated plane row by plane row
    float as[4][6];   // |A^T| |M| (FIX only)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float q[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) q[a] = src[(long)(a * 6 + b) * per];
      float o[4];
      wino4_at(wc, q, o);
      am[0][b] = o[0]; am[1][b] = o[1]; am[2][b] = o[2]; am[3][b] = o[3];
      if (FIX) {
#pragma unroll
        for (int a = 0; a < 6; ++a) q[a] = fabsf(q[a]);
        wino4_at_abs(wc, q, o);
        as[0][b] = o[0]; as[1][b] = o[1]; as[2][b] = o[2]; as[3][b] = o[3];
      }
    }
    const float bv = bias ? bias[m] : 0.f;
    const long ob = (((long)n * Mch + m) * H + 4 * ty) * W + 4 * tx;
    float vv[4][4], tau[4][4];   // FIX only
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float v[4];
      wino4_at(wc, am[a], v);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += bv;
      if (FIX) {
        float sv[4];
        wino4_at_abs(wc, as[a], sv);
#pragma unroll
        for (int j = 0; j < 4; ++j) { vv[a][j] = v[j]; tau[a][j] = fx.kappa_eps * sv[j]; }
      }
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      const long oo = ob + (long)a * W;
      if (EDGE) {   // outputs past the border are dropped
        if (4 * ty + a >= H) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (4 * tx + j < W) {
            float o = v[j];
            if (mask) o = mask[oo + j] > 0.f ? o : 0.f;
            y[oo + j] = o;
          }
        }
        continue;
      }
      if (mask) {
        const float4 mk = *reinterpret_cast<const float4*>(mask + oo);
        v[0] = mk.x > 0.f ? v[0] : 0.f; v[1] = mk.y > 0.f ? v[1] : 0.f;
        v[2] = mk.z > 0.f ? v[2] : 0.f; v[3] = mk.w > 0.f ? v[3] : 0.f;
      }
      *reinterpret_cast<float4*>(y + oo) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (FIX) {
      unsigned flags = 0;   // bit 4a + j
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (fabsf(vv[a][j]) < tau[a][j]) flags |= 1u << (4 * a + j);
      if (fx.tie_bits) {
        const unsigned viol = fx.tie_bits[t];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
          for (int wx = 0; wx < 2; ++wx) {   // the four 2x2 pool windows of the tile (tiles are 4-aligned, windows 2-aligned)
            float top = 0.f, second = 0.f, tw = 0.f;   // post-ReLU values: a window of non-positive outputs pools to 0 anyway
            int e1 = 0, e2 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int a = 2 * wy + (e >> 1), j = 2 * wx + (e & 1);
              const float rv = fmaxf(vv[a][j], 0.f);
              if (rv > top) { second = top; e2 = e1; top = rv; e1 = e; } else if (rv > second) { second = rv; e2 = e; }
              tw = fmaxf(tw, tau[a][j]);
            }
            if (top > 0.f && top - second < tw) {
              const unsigned v2 = (viol >> (2 * (2 * wy + wx))) & 3u;      // bit 0: rows not constant, bit 1: columns not constant
              const int diff = e1 ^ e2;                                    // 1: horizontal neighbours, 2: vertical, 3: diagonal
              const bool equals = diff == 1 ? !(v2 & 1u) : (diff == 2 ? !(v2 & 2u) : v2 == 0u);
              if (!equals) flags |= 0x33u << (8 * wy + 2 * wx);
            }
          }
      }
      while (flags) {
        const int bit = __ffs((int)flags) - 1;
        flags &= flags - 1;
        const int a = bit >> 2, j = bit & 3;
        if (EDGE && (4 * ty + a >= H || 4 * tx + j >= W)) continue;
        const unsigned pos = atomicAdd(fx.count, 1u);
        if (pos < fx.cap) fx.list[pos] = (unsigned)(ob + (long)a * W + j);
      }
    }
  }
}

// One wave per listed output: y = relu?(sum_{c,dy,dx} x[n][c][yy+dy-1][xx+dx-1] w[m][c][dy][dx] + bias[m]); lane l takes the
// channels l, l + 64, ... in ascending order (fma chain over its taps), then a fixed butterfly over the lanes.
__global__ __launch_bounds__(256) void wino_fixup_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         const unsigned int* __restrict__ list, const unsigned int* __restrict__ count,
                                                         unsigned cap, int C, int Mch, int H, int W, int relu) {
  const unsigned n_fix = min(*count, cap);
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), waves = gridDim.x * 4;
  const long HW = (long)H * W;
  for (unsigned i = wave; i < n_fix; i += waves) {
    const unsigned idx = list[i];
    const int xx = (int)(idx % W);
    unsigned r = idx / W;
    const int yy = (int)(r % H); r /= H;
    const int m = (int)(r % Mch);
    const int n = (int)(r / Mch);
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float* xp = x + ((long)n * C + c) * HW;
      const float* wp = w + ((long)m * C + c) * 9;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int py = yy + dy - 1;
        if (py < 0 || py >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int px = xx + dx - 1;
          if (px < 0 || px >= W) continue;
          acc = fmaf(xp[(long)py * W + px], wp[dy * 3 + dx], acc);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      float v = acc + (bias ? bias[m] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      y[idx] = v;
    }
  }
}

static long wino_tpad(long T) { return (T + WBN - 1) / WBN * WBN; }

struct WinoGemmParams {
  const float* U;   // [16][MT][S][32][128]
  const float* V;   // [16][S*32][Tpad]
  float* Mx;        // [16][Mpad][Tpad]
  int MT, S, C;
  long Tpad;        // >= TT * 128
  long TT;          // 128-wide tile columns
  int Mpad;         // MT * 128
  int planes;       // 16: F(2x2,3x3), 36: F(4x4,3x3) - the leading dimension of U / V / Mx
};

// grid.x = 16 * TT * MT, XCD-aware: the MT workgroups that share one V tile get consecutive slots on one XCD
__global__ __launch_bounds__(256, 2) void wino_gemm_kernel(WinoGemmParams p) {
  constexpr int LDA = WLDA, LDB = WBN;
  constexpr int TM = 2, TN = 2;
  constexpr int KS = WK / 2;                 // 16 MFMA k-steps per stage
  constexpr int NV = 4;                      // float4 units per thread per operand per stage
  constexpr int SFLUSH = 4;                  // stages per MFMA accumulation chain (128 k)
  __shared__ __attribute__((aligned(16))) float As[2][WK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][WK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const long TT = p.TT;
  const int xcd = blockIdx.x & 7;
  const long qq = blockIdx.x >> 3;
  const int mt = (int)(qq % p.MT);
  const long bt = (qq / p.MT) * 8 + xcd;     // (xi, t-tile) index
  if (bt >= p.planes * TT) return;
  const int xi = (int)(bt / TT);
  const long t0 = (bt % TT) * WBN;
  const float* Ub = p.U + (((long)xi * p.MT + mt) * p.S) * (WK * WBM);
  const float* Vb = p.V + (long)xi * (p.S * WK) * p.Tpad + t0;

  // staging: unit u = tid + 256 v covers row k = u / 32, quad q = u % 32 of a [32][128] operand image, i.e. float
  // offset 4u in the U image and (k, 4q) in V.  Named registers (no arrays): hipcc spills indexed float4 arrays here.
  const float* pa = Ub + 4 * tid;
  const float* pb = Vb + (long)(tid >> 5) * p.Tpad + 4 * (tid & 31);
  const long bstep = 8 * p.Tpad;
  float* sa = &As[0][(tid >> 5) * LDA + 4 * (tid & 31)];
  float* sb = &Bs[0][(tid >> 5) * LDB + 4 * (tid & 31)];
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  auto lda = [&](int v, int s) { return *reinterpret_cast<const float4*>(pa + (long)s * (WK * WBM) + 1024 * v); };
  auto ldb = [&](int v, int s) { return *reinterpret_cast<const float4*>(pb + (long)s * 4 * bstep + v * bstep); };
  auto sta = [&](int v, int buf, const float4& r) { *reinterpret_cast<float4*>(sa + buf * (WK * LDA) + 8 * v * LDA) = r; };
  auto stb = [&](int v, int buf, const float4& r) { *reinterpret_cast<float4*>(sb + buf * (WK * LDB) + 8 * v * LDB) = r; };
  auto piece = [&](int q, int sn, int nbuf) {   // q = 0..15: 8 loads for stage sn, then 8 stores into buffer nbuf
    switch (q) {
      case 0: ra0 = lda(0, sn); break;
      case 1: ra1 = lda(1, sn); break;
      case 2: ra2 = lda(2, sn); break;
      case 3: ra3 = lda(3, sn); break;
      case 4: rb0 = ldb(0, sn); break;
      case 5: rb1 = ldb(1, sn); break;
      case 6: rb2 = ldb(2, sn); break;
      case 7: rb3 = ldb(3, sn); break;
      case 8: sta(0, nbuf, ra0); break;
      case 9: sta(1, nbuf, ra1); break;
      case 10: sta(2, nbuf, ra2); break;
      case 11: sta(3, nbuf, ra3); break;
      case 12: stb(0, nbuf, rb0); break;
      case 13: stb(1, nbuf, rb1); break;
      case 14: stb(2, nbuf, rb2); break;
      default: stb(3, nbuf, rb3); break;
    }
  };

  f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

  const int ns = p.S;
#pragma unroll
  for (int q = 0; q < 16; ++q) piece(q, 0, 0);
  __syncthreads();
  for (int s0 = 0; s0 < ns; s0 += SFLUSH) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int s1 = min(ns, s0 + SFLUSH);
    for (int s = s0; s < s1; ++s) {
      const int cur = s & 1;
      const int sn = min(s + 1, ns - 1);
      const float* as = As[cur] + half * LDA + wm * 64 + l31;
      const float* bs = Bs[cur] + half * LDB + wn * 64 + l31;
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const int cb = kk & 1, nb = cb ^ 1;
        const int k1 = kk + 1;
#pragma unroll
        for (int m = 0; m < TM * TN; ++m) {
          const int i = m / TN, j = m % TN;
          acc[i][j] = mfma32(a[cb][i], b[cb][j], acc[i][j]);
          if (m == 0 && k1 < KS) {
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) a[nb][ii] = as[2 * k1 * LDA + ii * 32];
          }
          if (m == 1 && k1 < KS) {
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) b[nb][jj] = bs[2 * k1 * LDB + jj * 32];
          }
          if (m == 3) piece(kk, sn, cur ^ 1);   // one staging piece per k-step
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) tot[i][j] += acc[i][j];
  }
  float* Mb = p.Mx + ((long)xi * p.Mpad + mt * WBM) * p.Tpad + t0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Mb[(long)(wm * 64 + i * 32 + mfma_row(r, lane)) * p.Tpad + wn * 64 + j * 32 + l31] = tot[i][j][r];
}

// Same GEMM with the operand tiles copied global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: no staging
// registers, no ds_write, no address VALU).  Both stage images are lane-linear: U is stored in image order, and a V
// stage is 32 rows of 128 contiguous floats, so one wave-instruction (64 lanes x 16 B) fills two unpadded 512-B rows.
// Unpadded rows are conflict-free for the fragment reads (32 consecutive floats per half-wave).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// FOLD: how the MFMA accumulation chains are folded into the running total.  In a Winograd GEMM the accumulation rounding is
// the dominant error of the whole convolution (CPU ablation, round 3: with exact accumulation the 4x4 tile's result is 2-3e-7
// rms from float64, like the direct kernel; with one fp32 chain over all channels 1e-6): the outputs are differences of plane
// values ~20x their own size, so every eps of a partial sum counts 20-fold.  0: chains of 128 channels, fp32 total (rounds 1-2:
// 9.2e-7 measured on the 4x4 tile); 1: chains of one 32-channel stage, fp32 total; 2: chains of one stage, FLOAT64 total (the
// fold's cvt + v_add_f64 run under the other wave's MFMAs).
template <int SK, int OCC, int FOLD>
__global__ __launch_bounds__(256, OCC) void wino_gemm_dma_kernel(WinoGemmParams p) {
  constexpr int LD = 128;
  constexpr int TM = 2, TN = 2;
  constexpr int KS = SK / 2;            // MFMA k-steps per stage
  constexpr int NP = SK / 8;            // 1-KiB DMA pieces per wave, operand and stage
  constexpr int SFLUSH = FOLD == 0 ? 128 / SK : 1;      // stages per MFMA accumulation chain
  typedef typename std::conditional<FOLD == 2, double, float>::type tot_t;
  __shared__ __attribute__((aligned(1024))) float As[2][SK * LD];
  __shared__ __attribute__((aligned(1024))) float Bs[2][SK * LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const long TT = p.TT;
  const int xcd = blockIdx.x & 7;
  const long qq = blockIdx.x >> 3;
  const int mt = (int)(qq % p.MT);
  const long bt = (qq / p.MT) * 8 + xcd;
  if (bt >= p.planes * TT) return;
  const int xi = (int)(bt / TT);
  const long t0 = (bt % TT) * WBN;
  const float* Ub = p.U + (((long)xi * p.MT + mt) * p.S) * (WK * WBM);
  const float* Vb = p.V + (long)xi * (p.S * WK) * p.Tpad + t0;
  // piece j (0..3) of this wave: 256 floats of the stage image at float offset (wave * 4 + j) * 256
  const float* ga = Ub + wave * (NP * 256) + lane * 4;
  const float* gb = Vb + (long)(wave * (2 * NP) + half) * p.Tpad + l31 * 4;
  const long bstep = 2 * p.Tpad;
  // Issued by inline asm: through the builtin hipcc waits vmcnt(0) right after every DMA (it cannot prove that the
  // DMA target and the fragment reads do not alias), which serialises the copy with the MFMAs.  M0 = LDS byte address
  // of the piece (wave-uniform); the hardware adds lane * 16.  M0 is saved and restored around the instruction.
  const unsigned lds_a = (unsigned)(size_t)(lds_ptr_t)(&As[0][0]) + (unsigned)wave * (NP * 1024u);
  const unsigned lds_b = (unsigned)(size_t)(lds_ptr_t)(&Bs[0][0]) + (unsigned)wave * (NP * 1024u);
  auto glds = [&](const float* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto dma = [&](int q, int sn, int nbuf) {   // q = 0..NP-1: A pieces, NP..2NP-1: B pieces; sn counts SK-deep stages
    if (q < NP) glds(ga + (long)sn * (SK * WBM) + q * 256, lds_a + (unsigned)nbuf * (SK * LD * 4u) + (unsigned)q * 1024u);
    else glds(gb + (long)sn * SK * p.Tpad + (q - NP) * bstep, lds_b + (unsigned)nbuf * (SK * LD * 4u) + (unsigned)(q - NP) * 1024u);
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  f32x16 acc[TM][TN];
  tot_t tot[TM][TN][16];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = (tot_t)0; }

  const int ns = p.S * (WK / SK);
#pragma unroll
  for (int q = 0; q < 2 * NP; ++q) dma(q, 0, 0);
  dma_wait();
  __syncthreads();
  for (int s0 = 0; s0 < ns; s0 += SFLUSH) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int s1 = min(ns, s0 + SFLUSH);
    for (int s = s0; s < s1; ++s) {
      const int cur = s & 1;
      const int sn = min(s + 1, ns - 1);
      const float* as = As[cur] + half * LD + wm * 64 + l31;
      const float* bs = Bs[cur] + half * LD + wn * 64 + l31;
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const int cb = kk & 1, nb = cb ^ 1;
        const int k1 = kk + 1;
#pragma unroll
        for (int m = 0; m < TM * TN; ++m) {
          const int i = m / TN, j = m % TN;
          acc[i][j] = mfma32(a[cb][i], b[cb][j], acc[i][j]);
          if (m == 0 && k1 < KS) {
#pragma unroll
            for (int ii = 0; ii < TM; ++ii) a[nb][ii] = as[2 * k1 * LD + ii * 32];
          }
          if (m == 1 && k1 < KS) {
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) b[nb][jj] = bs[2 * k1 * LD + jj * 32];
          }
          if (m == 3 && kk < 2 * NP) dma(kk, sn, cur ^ 1);   // the other buffer: every wave left it at the last barrier
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      dma_wait();        // this wave's pieces of stage s + 1 have landed ...
      __syncthreads();   // ... and after the barrier every wave's have
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] += (tot_t)acc[i][j][r];
  }
  float* Mb = p.Mx + ((long)xi * p.Mpad + mt * WBM) * p.Tpad + t0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Mb[(long)(wm * 64 + i * 32 + mfma_row(r, lane)) * p.Tpad + wn * 64 + j * 32 + l31] = (float)tot[i][j][r];
}

const int g_wino_dma = umpr_env_int("UMPR_WINO_DMA", 1);   // 0 off, 1: 32-deep stages, 2: 16-deep stages at 3 waves/SIMD

inline int nblk(long n, int cap) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

// UMPR_WINO_F4: 0 = F(2x2,3x3) everywhere; 1 = F(4x4,3x3) for the data gradient only (the round-2 default);
// 2 (default since round 3) = the training forward as well.  The backward pass is linear in its inputs (the ReLU / pool
// decisions were taken in forward), so the larger tile's rounding stays a ~1e-6 perturbation of the gradient.  In FORWARD the
// same rounding would flip several times more ReLU / max-pool decisions than the direct kernel's 1.7e-7 does (round 2, textbook
// points: the first layers' gradients of golden umpr_full_V1_B2_randnM moved to 5e-3 relative L2 from the float64 run where the
// reference's own fp32 is at 1.5e-3).  Round 3 removes the cause twice over: better-conditioned interpolation points (half the
// rounding) and the decision fix-up of wino4_output_kernel / wino_fixup_kernel, which retakes every decision the tile's rounding
// could change at the direct kernel's accuracy (the same fixture: 8e-4; tools/count_flips.py counts the decisions).
int umpr_wino_f4_mode() {
  static const int mode = umpr_env_int("UMPR_WINO_F4", 2);   // function-local: also read by conv3x3.hip's initialisers
  return mode;
}
// Inference (umpr_set_conv_inference, per host thread): no backward pass will read this forward's ReLU / pool decisions,
// so the forward pass may take the larger tile as well - predictions move by 3e-6 (bound 1e-4).
static thread_local int t_wino_infer = 0;
void umpr_wino_set_inference(int on) { t_wino_infer = on; }
int umpr_wino_inference() { return t_wino_infer && umpr_wino_f4_mode() >= 1; }
// The 4x4 tile pays where 36 planes over ceil(H/4) x ceil(W/4) tiles are less GEMM work than 16 planes over (H/2) x (W/2):
// always on maps that are a multiple of 4, and on 14x14 (16 tiles of which 3.75 hang over the border: 576 vs 784 plane-tiles).
static inline bool wino_f4_shape(int H, int W) {
  return 36L * ((H + 3) / 4) * ((W + 3) / 4) < 16L * (H / 2) * (W / 2);
}
// A 2x2 max-pool follows this forward convolution (set by the VGG16 forward around conv3_3 / conv4_3 / conv5_3; thread-local
// like the inference hint): the decision fix-up then also covers the pool windows' argmax (see WinoFix).
static thread_local int t_wino_pool_follows = 0;
void umpr_wino_set_pool_follows(int on) { t_wino_pool_follows = on; }
// V slot (round 3): the transformed input V = B^T d B of a training forward on the 4x4 tile is exactly what the weight gradient
// of the same layer transforms again in backward (same input, same tile, same padded layout).  When the caller names a buffer
// that outlives the forward (umpr_vgg16_features_fwd: a region of the activation arena), the forward GEMM reads its B operand
// from there and the weight gradient skips its own input transform (seven of the ten Winograd layers qualify; ~0.9 ms of pure
// HBM traffic per step for 2.7 GB of arena at batch 64).  Per host thread, like the other hints.
static thread_local float* t_v_slot = nullptr;
static thread_local size_t t_v_slot_floats = 0;
void umpr_wino_set_v_slot(float* p, size_t floats) { t_v_slot = p; t_v_slot_floats = floats; }

// test / tooling aid: the list counter of this host thread's most recent fix-up pass (device memory inside that call's workspace)
static thread_local const unsigned int* t_last_fix_count = nullptr;
long umpr_wino_last_fix_count() {
  if (!t_last_fix_count) return -1;
  unsigned int v = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, t_last_fix_count, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return (long)v;
}
static inline bool wino_f4_map(int H, int W, int transposed) {
  const int need = transposed || t_wino_infer ? 1 : 2;
  return umpr_wino_f4_mode() >= need && wino_f4_shape(H, W);
}

// workspace: U [P][MT*128][S*32] + V [P][S*32][Tpad] + M [P][MT*128][Tpad]  (floats);  P = 16 planes over 2x2 tiles, or
// 36 planes over 4x4 tiles where the map allows F(4x4,3x3)
size_t umpr_wino_ws_floats(int N, int C, int M, int H, int W, int transposed) {
  const long MT = (M + WBM - 1) / WBM, S = (C + WK - 1) / WK;
  auto layout = [&](bool f4) {
    const long T = f4 ? (long)N * ((H + 3) / 4) * ((W + 3) / 4) : (long)N * (H / 2) * (W / 2);
    const long Tpad = wino_tpad(T);
    return (size_t)(f4 ? 36 : 16) * (MT * WBM * S * WK + S * WK * Tpad + MT * WBM * Tpad) + 64 + (f4 ? Tpad : 0);   // + fix-up counter and per-tile tie bits
  };
  const bool map4 = wino_f4_shape(H, W);
  const int mode = umpr_wino_f4_mode();
  if (!map4 || mode == 0) return layout(false);
  if (transposed) return layout(true);
  // forward: the 2x2 tile (mode 1 training; in mode 2 the layers a max-pool follows) or the 4x4 one (mode 2; inference)
  const size_t a = layout(false), b = layout(true);
  return a > b ? a : b;
}

// Image chunking: V and M (each 4x the activation they transform) are written by one kernel and read back by the next.
// With the whole batch in one pass (411 MB + 822 MB at 56x56x128->256, batch 64) every one of those bytes goes to HBM and
// comes back; a chunk whose V + M stay under the 256 MiB Infinity Cache is re-read on die
// (MI355X_MICROARCH.md, Infinity Cache: a line stays resident while the bytes touched between two uses fit).
// UMPR_WINO_CHUNK_MB = budget for V + M of one chunk (0 = whole batch in one pass, the default).
// Measured (batch 64, fp32 step): 0 -> 42.5 ms, 224 -> 45.3, 160 -> 46.5, 96 -> 50.7: the smaller GEMMs and the extra
// launches cost more than on-die re-reads give back, so it stays an experiment switch (profiles/README.md, r02_n).
static const long g_wino_chunk_mb = (long)umpr_env_int("UMPR_WINO_CHUNK_MB", 0);
static int wino_chunk_images(int N, long floats_per_image) {
  if (g_wino_chunk_mb <= 0) return N;
  long nc = (g_wino_chunk_mb << 20) / (floats_per_image * 4);
  if (nc < 1) nc = 1;
  if (nc >= N) return N;
  const int parts = (int)((N + nc - 1) / nc);          // equal parts: no small tail chunk
  return (N + parts - 1) / parts;
}

// floats of the forward V of a layer whose weight gradient can read it back (0: the layer does not qualify): training forward
// and weight gradient both on the 4x4 tile over the same tile grid, and the two GEMMs' channel paddings agree
size_t umpr_wino_v_floats(int N, int Cin, int Cout, int H, int W) {
  if (umpr_wino_f4_mode() < 2 || !wino_f4_shape(H, W) || (H % 4) != 0 || (W % 4) != 0) return 0;
  static const int on = umpr_env_int("UMPR_WINO_V_REUSE", 1);
  if (!on || g_wino_chunk_mb > 0) return 0;
  const long S = (Cin + WK - 1) / WK;
  const long bnc = Cin <= 64 ? 64 : WBN;
  if (S * WK != (Cin + bnc - 1) / bnc * bnc) return 0;
  const long T = (long)N * (H / 4) * (W / 4);
  return (size_t)36 * S * WK * wino_tpad(T);
}

static int wino_conv3x3_pass(const float* x, const float* w, int transposed, const float* bias, const float* mask, float* y,
                             int N, int Cin, int Cout, int H, int W, int relu, float* ws, size_t ws_floats, hipStream_t s,
                             bool weights_ready);

// forward (transposed = 0) or data gradient (transposed = 1), same contract as umpr_conv3x3_run
int umpr_wino_conv3x3(const float* x, const float* w, int transposed, const float* bias, const float* mask, float* y,
                      int N, int Cin, int Cout, int H, int W, int relu, float* ws, size_t ws_floats, hipStream_t s) {
  const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
  UMPR_REQUIRE(ws_floats >= umpr_wino_ws_floats(N, C, M, H, W, transposed), "winograd: workspace too small");
  const int nc = wino_chunk_images(N, (long)16 * (C + M) * (H / 2) * (W / 2));
  for (int n0 = 0; n0 < N; n0 += nc) {
    const int n = N - n0 < nc ? N - n0 : nc;
    // U sits at the start of the workspace and does not depend on the chunk size: transformed once
    if (int rc = wino_conv3x3_pass(x + (size_t)n0 * C * H * W, w, transposed, bias, mask ? mask + (size_t)n0 * M * H * W : nullptr,
                                   y + (size_t)n0 * M * H * W, n, Cin, Cout, H, W, relu, ws, ws_floats, s, n0 > 0)) return rc;
  }
  return 0;
}

static int wino_conv3x3_pass(const float* x, const float* w, int transposed, const float* bias, const float* mask, float* y,
                             int N, int Cin, int Cout, int H, int W, int relu, float* ws, size_t ws_floats, hipStream_t s,
                             bool weights_ready) {
  UMPR_REQUIRE((H % 2) == 0 && (W % 2) == 0, "winograd: odd map %dx%d", H, W);
  const int M = transposed ? Cin : Cout;
  const int C = transposed ? Cout : Cin;
  UMPR_REQUIRE(ws_floats >= umpr_wino_ws_floats(N, C, M, H, W, transposed), "winograd: workspace too small");
  const bool f4 = wino_f4_map(H, W, transposed);
  const int planes = f4 ? 36 : 16;
  const long T = f4 ? (long)N * ((H + 3) / 4) * ((W + 3) / 4) : (long)N * (H / 2) * (W / 2);
  const bool edge = f4 && ((H % 4) != 0 || (W % 4) != 0);
  const long Tpad = wino_tpad(T);
  const int MT = (M + WBM - 1) / WBM, S = (C + WK - 1) / WK;
  float* U = ws;
  float* V = U + (size_t)planes * MT * WBM * S * WK;
  float* Mx = V + (size_t)planes * S * WK * Tpad;
  float* list_region = V;        // the fix-up list lives in the workspace's V region (free once the GEMM has read it, or at once
                                 // when V itself lives in the caller's slot)
  if (t_v_slot && !transposed) {
    const size_t vfl = umpr_wino_v_floats(N, Cin, Cout, H, W);
    UMPR_REQUIRE(f4 && !t_wino_infer && vfl > 0 && vfl == (size_t)planes * S * WK * Tpad && t_v_slot_floats >= vfl,
                 "winograd: the V slot does not fit this forward (%d -> %d at %dx%d)", Cin, Cout, H, W);
    V = t_v_slot;
  }
  if (!weights_ready) {
    if (f4) wino4_weights_kernel<<<nblk((long)MT * WBM * S * WK, 2048), 256, 0, s>>>(w, U, M, C, Cin, transposed, wino4_consts());
    else wino_weights_kernel<<<nblk((long)MT * WBM * S * WK, 2048), 256, 0, s>>>(w, U, M, C, Cin, transposed);
    UMPR_LAUNCH_CHECK("wino_weights");
  }
  const long TT = (T + WBN - 1) / WBN;
  // decision fix-up (training forward on the 4x4 tile only): counter in the 64 spare floats behind Mx, list in V once the GEMM
  // has consumed it
  // tau = kappa * 2^-24 * S.  The tile's error is ~9e-7 rms of rms(y) and S ~ 20 |y|, so kappa = 8 is ~8 standard deviations for a
  // typical output (CPU emulation, tools/wino_flip_sim.py: kappa 8, 32 and 128 fix the same decisions); the list grows in
  // proportion - 6e-6 of the outputs on the golden fixtures' images, 3e-3 on the benchmark's i.i.d.-noise images, whose deep
  // feature maps are nearly flat (neighbouring outputs differ by ~3e-3 of their magnitude), at kappa = 8.
  static const int kappa = umpr_env_int("UMPR_WINO_FIX_KAPPA", 8);   // 0 disables
  const bool fix = f4 && !transposed && !t_wino_infer && kappa > 0 && mask == nullptr;
  WinoFix fx{nullptr, nullptr, 0u, 0.f, nullptr};
  unsigned int* tie_bits = nullptr;
  if (fix) {
    fx.count = reinterpret_cast<unsigned int*>(Mx + (size_t)planes * MT * WBM * Tpad);
    fx.list = reinterpret_cast<unsigned int*>(list_region);
    const size_t vfl = (size_t)planes * S * WK * Tpad;
    fx.cap = (unsigned)(vfl < (size_t)0x7fffffff ? vfl : (size_t)0x7fffffff);
    fx.kappa_eps = (float)kappa * 5.9604644775390625e-08f;
    UMPR_REQUIRE((long)N * M * H * W < 0xffffffffL, "winograd fix-up: output tensor too large for 32-bit element indices");
    if (t_wino_pool_follows) {   // one word per tile behind the counter (umpr_wino_ws_floats reserves Tpad + 64 floats there)
      tie_bits = fx.count + 16;   // tie_bits[-1]: the input's largest magnitude
      if (hipMemsetAsync(tie_bits - 1, 0, (size_t)(T + 1) * sizeof(unsigned int), s) != hipSuccess) { umpr_set_error("winograd fix-up: memset"); return -2; }
      UMPR_REQUIRE(((long)N * C * H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "winograd fix-up: unaligned input");
      wino_absmax_kernel<<<nblk((long)N * C * H * W / 4, 1024), 256, 0, s>>>(x, (long)N * C * H * W / 4, tie_bits - 1);
      UMPR_LAUNCH_CHECK("wino_absmax");
      fx.tie_bits = tie_bits;
    }
  }
  if (f4 && edge) {
    if (tie_bits) wino4_input_kernel<true, true><<<nblk((long)S * WK * TT * WBN, 16384), 256, 0, s>>>(x, V, N, C, S * WK, H, W, Tpad, TT * WBN, wino4_consts(), fx.count, tie_bits);
    else wino4_input_kernel<true, false><<<nblk((long)S * WK * TT * WBN, 16384), 256, 0, s>>>(x, V, N, C, S * WK, H, W, Tpad, TT * WBN, wino4_consts(), fx.count, nullptr);
  } else if (f4) {
    if (tie_bits) wino4_input_kernel<false, true><<<nblk((long)S * WK * TT * WBN, 16384), 256, 0, s>>>(x, V, N, C, S * WK, H, W, Tpad, TT * WBN, wino4_consts(), fx.count, tie_bits);
    else wino4_input_kernel<false, false><<<nblk((long)S * WK * TT * WBN, 16384), 256, 0, s>>>(x, V, N, C, S * WK, H, W, Tpad, TT * WBN, wino4_consts(), fx.count, nullptr);
  }
  else if ((W / 2) % 2 == 0)
    wino_input_pair_kernel<<<nblk((long)S * WK * TT * WBN / 2, 16384), 256, 0, s>>>(x, V, N, C, S * WK, H, W, Tpad, TT * WBN);
  else
    wino_input_kernel<<<nblk((long)S * WK * TT * WBN, 16384), 256, 0, s>>>(x, V, N, C, S * WK, H, W, Tpad, TT * WBN);
  UMPR_LAUNCH_CHECK("wino_input");
  WinoGemmParams p{U, V, Mx, MT, S, C, Tpad, TT, MT * WBM, planes};
  const long groups = (planes * TT + 7) / 8 * 8;
  {
    UmprProfScope prof(UMPR_K_WINO_GEMM, 2.0 * planes * (double)M * C * T, s);
    static const int fold = umpr_env_int("UMPR_WINO_FOLD", 1);   // measured (4x4 tile, rms of rms y): 0: 9.2e-7, 1: 6.0e-7 at the same speed, 2: 5.4e-7 for -9 % GEMM rate
    if (g_wino_dma == 2) wino_gemm_dma_kernel<16, 3, 0><<<(unsigned)(groups * MT), 256, 0, s>>>(p);
    else if (g_wino_dma && fold == 2) wino_gemm_dma_kernel<32, 2, 2><<<(unsigned)(groups * MT), 256, 0, s>>>(p);
    else if (g_wino_dma && fold == 1) wino_gemm_dma_kernel<32, 2, 1><<<(unsigned)(groups * MT), 256, 0, s>>>(p);
    else if (g_wino_dma) wino_gemm_dma_kernel<32, 2, 0><<<(unsigned)(groups * MT), 256, 0, s>>>(p);
    else wino_gemm_kernel<<<(unsigned)(groups * MT), 256, 0, s>>>(p);
  }
  UMPR_LAUNCH_CHECK("wino_gemm");
  if (f4 && fix) {
    if (edge) wino4_output_kernel<true, true><<<nblk((long)M * T, 16384), 256, 0, s>>>(Mx, bias, mask, y, N, M, H, W, Tpad, MT * WBM, relu, wino4_consts(), fx);
    else wino4_output_kernel<false, true><<<nblk((long)M * T, 16384), 256, 0, s>>>(Mx, bias, mask, y, N, M, H, W, Tpad, MT * WBM, relu, wino4_consts(), fx);
    UMPR_LAUNCH_CHECK("wino_output");
    // 512 waves: the list is usually a few hundred outputs (most workgroups leave at once; 8192 waves cost 27 us per launch
    // just to start and leave, 2048 waves 25 us), each a latency-bound gather of 9 C inputs
    wino_fixup_kernel<<<128, 256, 0, s>>>(x, w, bias, y, fx.list, fx.count, fx.cap, C, M, H, W, relu);
    UMPR_LAUNCH_CHECK("wino_fixup");
    t_last_fix_count = fx.count;
    return 0;
  }
  if (f4 && edge)
    wino4_output_kernel<true, false><<<nblk((long)M * T, 16384), 256, 0, s>>>(Mx, bias, mask, y, N, M, H, W, Tpad, MT * WBM, relu, wino4_consts(), fx);
  else if (f4)
    wino4_output_kernel<false, false><<<nblk((long)M * T, 16384), 256, 0, s>>>(Mx, bias, mask, y, N, M, H, W, Tpad, MT * WBM, relu, wino4_consts(), fx);
  else if ((W / 2) % 2 == 0)
    wino_output_pair_kernel<<<nblk((long)M * T / 2, 16384), 256, 0, s>>>(Mx, bias, mask, y, N, M, H, W, Tpad, MT * WBM, relu);
  else
    wino_output_kernel<<<nblk((long)M * T, 16384), 256, 0, s>>>(Mx, bias, mask, y, N, M, H, W, Tpad, MT * WBM, relu);
  UMPR_LAUNCH_CHECK("wino_output");
  return 0;
}

// =====================================================================================================================
// Winograd weight gradient, F(3x3, 2x2):   dW = G^T [ sum_t (A g_t A^T) .* (B^T d_t B) ] G
//   g_t = 2x2 tile of the output gradient, d_t = the 4x4 input tile of the forward transform (same B), t over all
//   tiles of the batch.  16 GEMMs  P[xi][m][c] = sum_t Gy[xi][m][t] * V[xi][c][t]  with the reduction over tiles:
//   both operands are t-contiguous ("NT"), the tile range is split over workgroups (split-K) and the partial
//   products are summed in a fixed order by the finish kernel, which also applies G^T . G (4x4 -> 3x3).
//   A = [1 0; 1 1; 1 -1; 0 -1],  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1].
// =====================================================================================================================
namespace {

// Gy[xi][m][t] = (A g A^T)[xi],  g = dy[n][m][2ty..2ty+1][2tx..2tx+1];  columns t in [T, Tw) are written as zeros
__global__ void wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ Gy, int N, int Mch, int H, int W,
                               long Tpad, long Tw, int Mpad) {
  const int TH = H / 2, TW = W / 2;
  const long T = (long)N * TH * TW;
  const long total = (long)Mch * Tw;
  const long per = (long)Mpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i % Tw;
    const int m = (int)(i / Tw);
    float* dst = Gy + (long)m * Tpad + t;
    if (t >= T) {
#pragma unroll
      for (int a = 0; a < 16; ++a) dst[(long)a * per] = 0.f;
      continue;
    }
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = dy + (((long)n * Mch + m) * H + 2 * ty) * W + 2 * tx;
    const float2 g0 = *reinterpret_cast<const float2*>(src);
    const float2 g1 = *reinterpret_cast<const float2*>(src + W);
    // A g (4x2): rows g0, g0+g1, g0-g1, -g1;  then (A g) A^T per row: p, p+q, p-q, -q
    const float rp[4] = {g0.x, g0.x + g1.x, g0.x - g1.x, -g1.x};
    const float rq[4] = {g0.y, g0.y + g1.y, g0.y - g1.y, -g1.y};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      dst[(long)(a * 4 + 0) * per] = rp[a];
      dst[(long)(a * 4 + 1) * per] = rp[a] + rq[a];
      dst[(long)(a * 4 + 2) * per] = rp[a] - rq[a];
      dst[(long)(a * 4 + 3) * per] = -rq[a];
    }
  }
}

// two horizontally adjacent tiles per thread (even tile rows): float4 loads of dy, float2 stores
__global__ void wino_dy_pair_kernel(const float* __restrict__ dy, float* __restrict__ Gy, int N, int Mch, int H, int W,
                                    long Tpad, long Tw, int Mpad) {
  const int TH = H / 2, TW = W / 2;
  const long T = (long)N * TH * TW;
  const long Tw2 = Tw / 2;
  const long total = (long)Mch * Tw2;
  const long per = (long)Mpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = 2 * (i % Tw2);
    const int m = (int)(i / Tw2);
    float* dst = Gy + (long)m * Tpad + t;
    if (t >= T) {
#pragma unroll
      for (int a = 0; a < 16; ++a) *reinterpret_cast<float2*>(dst + (long)a * per) = make_float2(0.f, 0.f);
      continue;
    }
    const int tx = (int)(t % TW);
    const long r = t / TW;
    const int ty = (int)(r % TH), n = (int)(r / TH);
    const float* src = dy + (((long)n * Mch + m) * H + 2 * ty) * W + 2 * tx;
    const float4 g0 = *reinterpret_cast<const float4*>(src);        // row 0: tile 0 (x, y), tile 1 (z, w)
    const float4 g1 = *reinterpret_cast<const float4*>(src + W);    // row 1
    const float2 rp[4] = {make_float2(g0.x, g0.z), make_float2(g0.x + g1.x, g0.z + g1.z),
                          make_float2(g0.x - g1.x, g0.z - g1.z), make_float2(-g1.x, -g1.z)};
    const float2 rq[4] = {make_float2(g0.y, g0.w), make_float2(g0.y + g1.y, g0.w + g1.w),
                          make_float2(g0.y - g1.y, g0.w - g1.w), make_float2(-g1.y, -g1.w)};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 0) * per) = rp[a];
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 1) * per) = make_float2(rp[a].x + rq[a].x, rp[a].y + rq[a].y);
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 2) * per) = make_float2(rp[a].x - rq[a].x, rp[a].y - rq[a].y);
      *reinterpret_cast<float2*>(dst + (long)(a * 4 + 3) * per) = make_float2(-rq[a].x, -rq[a].y);
    }
  }
}

// db[m] = sum over n, y, x of dy[n][m][y][x], two fixed-order stages: one workgroup per (n, m) plane, then over n
// The same with the splits shared out over four thread groups of a workgroup (64 input channels x 4 groups): the layers with
// few output tiles have 16-32 splits, i.e. 256-512 slab loads per output for the one-thread-per-output form above
// (350 us for the 32 768 outputs of 128->256).  Group g sums the splits q = g, g+4, ... in order; the groups are combined
// in order through LDS.
__global__ __launch_bounds__(256) void wino_wgrad_finish_wide_kernel(const float* __restrict__ P, int splits, int Mch, int C,
                                                                     int Mpad, int Cpad, float* __restrict__ dw, int accumulate) {
  __shared__ float red[3][16][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int chunks = (C + 63) / 64;
  const int m = blockIdx.x / chunks, c = (blockIdx.x % chunks) * 64 + cl;
  const long per = (long)Mpad * Cpad;
  const bool ok = c < C;
  float v[4][4];
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    float acc = 0.f;
    if (ok)
      for (int sp = g; sp < splits; sp += 4) acc += P[((long)sp * 16 + a) * per + (long)m * Cpad + c];
    v[a >> 2][a & 3] = acc;
  }
  if (g > 0) {
#pragma unroll
    for (int a = 0; a < 16; ++a) red[g - 1][a][cl] = v[a >> 2][a & 3];
  }
  __syncthreads();
  if (g != 0 || !ok) return;
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a >> 2][a & 3] = ((v[a >> 2][a & 3] + red[0][a][cl]) + red[1][a][cl]) + red[2][a][cl];
  float gv[3][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    gv[0][b] = v[0][b] + 0.5f * (v[1][b] + v[2][b]);
    gv[1][b] = 0.5f * (v[1][b] - v[2][b]);
    gv[2][b] = 0.5f * (v[1][b] + v[2][b]) + v[3][b];
  }
  float* d = dw + ((long)m * C + c) * 9;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float o0 = gv[a][0] + 0.5f * (gv[a][1] + gv[a][2]);
    const float o1 = 0.5f * (gv[a][1] - gv[a][2]);
    const float o2 = 0.5f * (gv[a][1] + gv[a][2]) + gv[a][3];
    if (accumulate) { d[a * 3 + 0] += o0; d[a * 3 + 1] += o1; d[a * 3 + 2] += o2; }
    else { d[a * 3 + 0] = o0; d[a * 3 + 1] = o1; d[a * 3 + 2] = o2; }
  }
}

// db partial sums: one wave per (image, channel) plane, float4 loads, shuffle reduction (the workgroup-per-plane tree above
// spends most of its time in eight barriers for 3 KB of data)
__global__ __launch_bounds__(256) void wino_bias_part_wave_kernel(const float* __restrict__ dy, float* __restrict__ part, long HW,
                                                                  long planes) {
  const int lane = threadIdx.x & 63;
  const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const float4* src = reinterpret_cast<const float4*>(dy + pl * HW);
  float a0 = 0.f, a1 = 0.f;
  long i = lane;
  for (; i + 64 < HW / 4; i += 128) {
    const float4 u = src[i], w = src[i + 64];
    a0 += (u.x + u.y) + (u.z + u.w); a1 += (w.x + w.y) + (w.z + w.w);
  }
  for (; i < HW / 4; i += 64) { const float4 u = src[i]; a0 += (u.x + u.y) + (u.z + u.w); }
  const float a = wave_sum(a0 + a1);
  if (lane == 0) part[pl] = a;
}

__global__ void wino_bias_part_kernel(const float* __restrict__ dy, float* __restrict__ part, long HW) {
  __shared__ float red[256];
  const float4* src = reinterpret_cast<const float4*>(dy + (long)blockIdx.x * HW);   // plane index = n * Mch + m
  float a = 0.f;
  for (long i = threadIdx.x; i < HW / 4; i += 256) { const float4 v = src[i]; a += (v.x + v.y) + (v.z + v.w); }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// second stage of the fused form (wino4_dy_kernel's bias_part): one wave per channel over its `chunks` per-wave sums
__global__ __launch_bounds__(256) void wino_bias_sum_waves_kernel(const float* __restrict__ part, float* __restrict__ db, int Mch,
                                                                  long chunks, int accumulate) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= Mch) return;
  const float* src = part + (long)m * chunks;
  float a = 0.f;
  for (long i = lane; i < chunks; i += 64) a += src[i];
  a = wave_sum(a);
  if (lane == 0) db[m] = accumulate ? db[m] + a : a;
}
__global__ void wino_bias_sum_kernel(const float* __restrict__ part, float* __restrict__ db, int N, int Mch,
                                     int accumulate) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mch) return;
  float a = 0.f;
  for (int n = 0; n < N; ++n) a += part[(long)n * Mch + m];
  db[m] = accumulate ? db[m] + a : a;
}

// F(3x3,4x4), the weight-gradient counterpart of F(4x4,3x3):  dW = G^T [ sum_t (A g_t A^T) .* (B^T d_t B) ] G  with g_t the 4x4
// tile of the output gradient, d_t the 6x6 input tile, A = (A^T)^T (6x4), G (6x3) as above: 36 GEMMs over a quarter of the
// tiles - 4x fewer multiplies than the direct weight gradient (F(3x3,2x2): 2.25x).
// Gy[xi][m][t] = (A g A^T)[xi];  columns t in [T, Tw) are written as zeros
__device__ __forceinline__ void wino4_a(const Wino4C& c, const float g0, const float g1, const float g2, const float g3,
                                        float* __restrict__ o) {
  const float ea = g0 + c.a2 * g2, oa = c.a * g1 + c.a3 * g3, eb = g0 + c.b2 * g2, ob = c.b * g1 + c.b3 * g3;
  o[0] = g0; o[1] = ea + oa; o[2] = ea - oa; o[3] = eb + ob; o[4] = eb - ob; o[5] = g3;
}
// bias_part (optional, !EDGE only): [Mch][Tw / 64] - the sum of the gradient pixels each WAVE transformed (64 consecutive tiles of
// one channel: Tw is a multiple of 64), the first stage of the bias gradient; wino_bias_sum_waves_kernel adds them per channel in a
// fixed order.  Replaces a separate pass over dy (wino_bias_part_wave_kernel: 0.35 ms per step of pure re-reading).
template <bool EDGE>
__global__ __launch_bounds__(256) void wino4_dy_kernel(const float* __restrict__ dy, float* __restrict__ Gy, int N, int Mch, int H,
                                                       int W, long Tpad, long Tw, int Mpad, Wino4C wc, float* __restrict__ bias_part) {
  const int TH = (H + 3) / 4, TW = (W + 3) / 4;
  const long T = (long)N * TH * TW;
  const long total = (long)Mch * Tw;
  const long per = (long)Mpad * Tpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i % Tw;
    const int m = (int)(i / Tw);
    float* dst = Gy + (long)m * Tpad + t;
    float bsum = 0.f;
    if (t >= T) {
#pragma unroll
      for (int a = 0; a < 36; ++a) dst[(long)a * per] = 0.f;
    } else {
      const int tx = (int)(t % TW);
      const long r = t / TW;
      const int ty = (int)(r % TH), n = (int)(r / TH);
      const float* src = dy + (((long)n * Mch + m) * H + 4 * ty) * W + 4 * tx;
      float e[4][6];   // (row of g) A^T
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (EDGE) {   // gradient rows / columns past the border do not exist: zero
          const bool oky = 4 * ty + a < H;
          float d[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = oky && 4 * tx + j < W;
            const float v = src[ok ? (long)a * W + j : 0];
            d[j] = ok ? v : 0.f;
          }
          wino4_a(wc, d[0], d[1], d[2], d[3], e[a]);
          continue;
        }
        const float4 g = *reinterpret_cast<const float4*>(src + (long)a * W);
        bsum += (g.x + g.y) + (g.z + g.w);
        wino4_a(wc, g.x, g.y, g.z, g.w, e[a]);
      }
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        float o[6];
        wino4_a(wc, e[0][b], e[1][b], e[2][b], e[3][b], o);
#pragma unroll
        for (int a = 0; a < 6; ++a) dst[(long)(a * 6 + b) * per] = o[a];
      }
    }
    if (!EDGE && bias_part) {    // the whole wave is here: total and the wave's first i are multiples of 64
      const float wsum = wave_sum(bsum);
      if ((threadIdx.x & 63) == 0) bias_part[i >> 6] = wsum;
    }
  }
}

// dw[m][c][3][3] (+)= G^T (sum_split P[split][.][m][c]) G for 36 planes; the splits are shared out over four thread groups
// (q = g, g+4, ... in order) and the groups combined in order through LDS, as in wino_wgrad_finish_wide_kernel
__device__ __forceinline__ void wino4_gt(const Wino4C& c, const float* __restrict__ v, int stride, float* __restrict__ o) {
  const float s12 = v[stride] + v[2 * stride], d12 = v[stride] - v[2 * stride];
  const float s34 = v[3 * stride] + v[4 * stride], d34 = v[3 * stride] - v[4 * stride];
  o[0] = c.g0 * v[0] + c.ca * s12 + c.cb * s34;
  o[1] = (c.ca * c.a) * d12 + (c.cb * c.b) * d34;
  o[2] = (c.ca * c.a2) * s12 + (c.cb * c.b2) * s34 + v[5 * stride];
}
__global__ __launch_bounds__(256) void wino4_wgrad_finish_kernel(const float* __restrict__ P, int splits, int Mch, int C, int Mpad,
                                                                 int Cpad, float* __restrict__ dw, int accumulate, Wino4C wc) {
  __shared__ float red[3][36][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int chunks = (C + 63) / 64;
  const int m = blockIdx.x / chunks, c = (blockIdx.x % chunks) * 64 + cl;
  const long per = (long)Mpad * Cpad;
  const bool ok = c < C;
  float v[36];
#pragma unroll
  for (int a = 0; a < 36; ++a) {
    float acc = 0.f;
    if (ok)
      for (int sp = g; sp < splits; sp += 4) acc += P[((long)sp * 36 + a) * per + (long)m * Cpad + c];
    v[a] = acc;
  }
  if (g > 0) {
#pragma unroll
    for (int a = 0; a < 36; ++a) red[g - 1][a][cl] = v[a];
  }
  __syncthreads();
  // group 0 finishes; its nine outputs per (m, c) leave through LDS so that the 64 channels of the chunk are written as one
  // contiguous run of 576 floats (dw is [m][c][9]: a direct store is nine 36-B-strided store instructions per wave)
  __shared__ float outs[64 * 9];
  if (g == 0 && ok) {
#pragma unroll
    for (int a = 0; a < 36; ++a) v[a] = ((v[a] + red[0][a][cl]) + red[1][a][cl]) + red[2][a][cl];
    float gv[3][6];   // G^T v (over the plane row index), per plane column b
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float o[3];
      wino4_gt(wc, v + b, 6, o);
      gv[0][b] = o[0]; gv[1][b] = o[1]; gv[2][b] = o[2];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float o[3];
      wino4_gt(wc, gv[a], 1, o);
      outs[cl * 9 + a * 3 + 0] = o[0]; outs[cl * 9 + a * 3 + 1] = o[1]; outs[cl * 9 + a * 3 + 2] = o[2];
    }
  }
  __syncthreads();
  const int c0 = (blockIdx.x % chunks) * 64;
  const int nout = (C - c0 < 64 ? C - c0 : 64) * 9;
  float* d = dw + ((long)m * C + c0) * 9;
  for (int i = threadIdx.x; i < nout; i += 256) d[i] = accumulate ? d[i] + outs[i] : outs[i];
}

struct WinoWgradParams {
  const float* Gy;   // [planes][Mpad][Tpad]
  const float* V;    // [planes][Cpad][Tpad]
  float* P;          // [splits][planes][Mpad][Cpad]
  int MT, CT, Mpad, Cpad;
  long Tpad;
  int stages;            // Tpad / 32
  int stages_per_split;
  int planes;            // 16: F(3x3,2x2), 36: F(3x3,4x4)
};

constexpr int WLDK = WK + 4;   // LDS row pitch (floats) of the row-major ([row][k]) operand images: 144-B rows

// grid.x = planes * MT * CT * splits.  Both operands arrive k-contiguous ([row][t]) and stay that way in LDS: a thread's float4
// (four consecutive t of one row) is ONE ds_write_b128, and a lane's MFMA operands for four consecutive k-steps are ONE
// ds_read_b128.  That works because the MFMA sums over k in any order as long as A and B agree: in every group of 8 tiles the
// lower half-wave takes t = 0..3 and the upper t = 4..7, one per step (the textbook assignment - t = 2 * step + half - would need
// stride-2 reads).  144-B rows put the 16 lanes of a b128 phase on 16 distinct 16-B bank groups.  Against the [k][row] image
// with scalar scatter stores this is a quarter of the LDS instructions; same-box A/B over the four Winograd weight-gradient layer
// shapes: 1.5-2.8 % faster per layer call (e.g. 512->512@28 0.807 -> 0.787 ms), the step unchanged within noise.
// TN = 2: 128 input channels per workgroup; TN = 1: 64 (layers with 64 input channels - conv2_1 - would leave half of a
// 128-wide tile multiplying zeros).
template <int TN>
__global__ __launch_bounds__(256, 2) void wino_wgrad_gemm_kernel(WinoWgradParams p) {
  constexpr int LD = WLDK;
  constexpr int TM = 2;
  constexpr int BNC = 64 * TN;    // input channels (rows of the V image) per workgroup
  constexpr int NG = WK / 8;      // groups of 8 tiles (4 MFMA k-steps) per stage
  constexpr int SFLUSH = 4;
  __shared__ __attribute__((aligned(16))) float As[2][WBM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BNC * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  long b = blockIdx.x;
  const int ct = (int)(b % p.CT); b /= p.CT;
  const int mt = (int)(b % p.MT); b /= p.MT;
  const int xi = (int)(b % p.planes);
  const int split = (int)(b / p.planes);
  const int s_begin = split * p.stages_per_split;
  const int s_end = min(p.stages, s_begin + p.stages_per_split);
  const int ns = s_end - s_begin;   // >= 1 by construction

  // unit u = tid + 256 v: row = u / 8 (0..127), k-quad = u % 8: eight lanes cover one row's 128 contiguous bytes
  const int urow = tid >> 3, uq = tid & 7;
  const float* pa = p.Gy + ((long)xi * p.Mpad + mt * WBM + urow) * p.Tpad + (long)s_begin * WK + 4 * uq;
  const float* pb = p.V + ((long)xi * p.Cpad + ct * BNC + urow) * p.Tpad + (long)s_begin * WK + 4 * uq;
  const long rstep = 32 * p.Tpad;   // 32 rows per v
  float* sa = &As[0][urow * LD + 4 * uq];
  float* sb = &Bs[0][urow * LD + 4 * uq];
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  auto lda = [&](int v, int s) { return *reinterpret_cast<const float4*>(pa + (long)s * WK + v * rstep); };
  auto ldb = [&](int v, int s) { return *reinterpret_cast<const float4*>(pb + (long)s * WK + v * rstep); };
  auto sta = [&](int v, int buf, const float4& r) { *reinterpret_cast<float4*>(sa + buf * (WBM * LD) + 32 * v * LD) = r; };
  auto stb = [&](int v, int buf, const float4& r) { *reinterpret_cast<float4*>(sb + buf * (BNC * LD) + 32 * v * LD) = r; };
  auto piece = [&](int q, int sn, int nbuf) {   // the V image has 2 * TN row groups of 32: pieces 6, 7, 14, 15 only for TN = 2
    switch (q) {
      case 0: ra0 = lda(0, sn); break;
      case 1: ra1 = lda(1, sn); break;
      case 2: ra2 = lda(2, sn); break;
      case 3: ra3 = lda(3, sn); break;
      case 4: rb0 = ldb(0, sn); break;
      case 5: rb1 = ldb(1, sn); break;
      case 6: if (TN == 2) rb2 = ldb(2, sn); break;
      case 7: if (TN == 2) rb3 = ldb(3, sn); break;
      case 8: sta(0, nbuf, ra0); break;
      case 9: sta(1, nbuf, ra1); break;
      case 10: sta(2, nbuf, ra2); break;
      case 11: sta(3, nbuf, ra3); break;
      case 12: stb(0, nbuf, rb0); break;
      case 13: stb(1, nbuf, rb1); break;
      case 14: if (TN == 2) stb(2, nbuf, rb2); break;
      default: if (TN == 2) stb(3, nbuf, rb3); break;
    }
  };
  auto comp = [](const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };

  f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

#pragma unroll
  for (int q = 0; q < 16; ++q) piece(q, 0, 0);
  __syncthreads();
  for (int s0 = 0; s0 < ns; s0 += SFLUSH) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int s1 = min(ns, s0 + SFLUSH);
    for (int s = s0; s < s1; ++s) {
      const int cur = s & 1;
      const int sn = min(s + 1, ns - 1);
      const float* as = As[cur] + (wm * 64 + l31) * LD + 4 * half;
      const float* bs = Bs[cur] + (wn * 32 * TN + l31) * LD + 4 * half;
      float4 a[2][TM], bq[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = *reinterpret_cast<const float4*>(as + i * 32 * LD);
#pragma unroll
      for (int j = 0; j < TN; ++j) bq[0][j] = *reinterpret_cast<const float4*>(bs + j * 32 * LD);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cb = g & 1, nb = cb ^ 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int m = 0; m < TM * TN; ++m) {
            const int i = m / TN, j = m % TN;
            acc[i][j] = mfma32(comp(a[cb][i], e), comp(bq[cb][j], e), acc[i][j]);
            if (e == 0 && m == 0 && g + 1 < NG) {
#pragma unroll
              for (int ii = 0; ii < TM; ++ii) a[nb][ii] = *reinterpret_cast<const float4*>(as + ii * 32 * LD + 8 * (g + 1));
            }
            if (e == 0 && m == 1 && g + 1 < NG) {
#pragma unroll
              for (int jj = 0; jj < TN; ++jj) bq[nb][jj] = *reinterpret_cast<const float4*>(bs + jj * 32 * LD + 8 * (g + 1));
            }
            if (m == TM * TN - 1) piece(4 * g + e, sn, cur ^ 1);   // one staging piece per k-step
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) tot[i][j] += acc[i][j];
  }
  float* Pb = p.P + ((((long)split * p.planes + xi) * p.Mpad + mt * WBM) * p.Cpad) + ct * BNC;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Pb[(long)(wm * 64 + i * 32 + mfma_row(r, lane)) * p.Cpad + wn * 32 * TN + j * 32 + l31] = tot[i][j][r];
}

// dw[m][c][3][3] (+)= G^T (sum_split P[split][.][m][c]) G
__global__ void wino_wgrad_finish_kernel(const float* __restrict__ P, int splits, int Mch, int C, int Mpad, int Cpad,
                                         float* __restrict__ dw, int accumulate) {
  const long total = (long)Mch * C;
  const long per = (long)Mpad * Cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int m = (int)(i / C);
    float v[4][4];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      float acc = 0.f;
      for (int sp = 0; sp < splits; ++sp) acc += P[((long)sp * 16 + a) * per + (long)m * Cpad + c];
      v[a >> 2][a & 3] = acc;
    }
    float gv[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      gv[0][b] = v[0][b] + 0.5f * (v[1][b] + v[2][b]);
      gv[1][b] = 0.5f * (v[1][b] - v[2][b]);
      gv[2][b] = 0.5f * (v[1][b] + v[2][b]) + v[3][b];
    }
    float* d = dw + i * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float o0 = gv[a][0] + 0.5f * (gv[a][1] + gv[a][2]);
      const float o1 = 0.5f * (gv[a][1] - gv[a][2]);
      const float o2 = 0.5f * (gv[a][1] + gv[a][2]) + gv[a][3];
      if (accumulate) { d[a * 3 + 0] += o0; d[a * 3 + 1] += o1; d[a * 3 + 2] += o2; }
      else { d[a * 3 + 0] = o0; d[a * 3 + 1] = o1; d[a * 3 + 2] = o2; }
    }
  }
}

constexpr int kWinoWgradTargetWgs = 1024;
static const int g_wino_wgrad_rounds = umpr_env_int("UMPR_WINO_WGRAD_ROUNDS", 1);   // 0: the plain 1024-workgroup target

struct WinoWgradGeom { long T, Tpad; int MT, CT, Mpad, Cpad, stages, splits, stages_per_split, planes, bnc; };
WinoWgradGeom wino_wgrad_geom(int N, int Cin, int Cout, int H, int W) {
  WinoWgradGeom g;
  // padded 14x14 maps stay on F(3x3,2x2): a quarter of the tiles is too short a reduction for the split-K GEMM and the 36-plane
  // finish (measured 0.367 vs 0.353 ms per layer), while the data gradient gains (0.312 vs 0.351 ms)
  const bool f4 = wino_f4_map(H, W, 1) && (H % 4) == 0 && (W % 4) == 0;
  g.planes = f4 ? 36 : 16;
  g.T = f4 ? (long)N * ((H + 3) / 4) * ((W + 3) / 4) : (long)N * (H / 2) * (W / 2);
  g.Tpad = wino_tpad(g.T);
  g.bnc = Cin <= 64 ? 64 : WBN;   // input-channel tile of the GEMM
  g.MT = (Cout + WBM - 1) / WBM; g.CT = (Cin + g.bnc - 1) / g.bnc;
  g.Mpad = g.MT * WBM; g.Cpad = g.CT * g.bnc;
  g.stages = (int)(g.Tpad / WK);
  const int tiles = g.planes * g.MT * g.CT;
  const int max_splits = g.stages / 8 > 0 ? g.stages / 8 : 1;   // at least 8 stages (256 tiles) per workgroup
  // The workgroups of a launch are equally long and 512 fit the chip at a time (2 per CU), so the launch takes
  // ceil(tiles * splits / 512) rounds: pick the split count that fills its last round best (36 tiles x 29 splits = 1044
  // workgroups ran 3 rounds for 2.04 rounds of work), the smallest one among those within 3 % of the best, never above the
  // plain 1024-workgroup target (28x28 layers with 7 splits of 14 stages measured 5-10 % slower than 2-4 splits); below one
  // round the grid cannot fill the chip anyway and the plain target applies.
  int splits = (kWinoWgradTargetWgs + tiles - 1) / tiles;
  if (g_wino_wgrad_rounds && (long)tiles * max_splits >= 512) {
    double best = 0.0;
    const int hi = max_splits < splits ? max_splits : splits;   // never more splits than the plain target: short workgroups lose more than a full round wins
    for (int sp = 1; sp <= hi; ++sp) {
      const long wgs = (long)tiles * sp;
      if (wgs < 512) continue;
      const double eff = (double)wgs / (double)((wgs + 511) / 512 * 512);
      if (eff > best) best = eff;
    }
    for (int sp = 1; sp <= hi; ++sp) {
      const long wgs = (long)tiles * sp;
      if (wgs < 512) continue;
      if ((double)wgs / (double)((wgs + 511) / 512 * 512) >= best - 0.03) { splits = sp; break; }
    }
  }
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  g.stages_per_split = (g.stages + splits - 1) / splits;
  g.splits = (g.stages + g.stages_per_split - 1) / g.stages_per_split;
  return g;
}

}  // namespace

// workspace: Gy [planes][Mpad][Tpad] + V [planes][Cpad][Tpad] + P [splits][planes][Mpad][Cpad] + bias partials [Mpad][Tpad / 64]
// (floats)
size_t umpr_wino_wgrad_ws_floats(int N, int Cin, int Cout, int H, int W) {
  const WinoWgradGeom g = wino_wgrad_geom(N, Cin, Cout, H, W);
  return (size_t)g.planes * ((size_t)g.Mpad * g.Tpad + (size_t)g.Cpad * g.Tpad + (size_t)g.splits * g.Mpad * g.Cpad) +
         (size_t)g.Mpad * (g.Tpad / 64) + 64;
}

static int wino_wgrad_pass(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Cout, int H, int W,
                           int accumulate, float* ws, size_t ws_floats, hipStream_t s);

int umpr_wino_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Cout, int H, int W,
                    int accumulate, float* ws, size_t ws_floats, hipStream_t s) {
  UMPR_REQUIRE(ws_floats >= umpr_wino_wgrad_ws_floats(N, Cin, Cout, H, W), "winograd wgrad: workspace too small");
  const int nc = wino_chunk_images(N, (long)16 * (Cin + Cout) * (H / 2) * (W / 2));
  for (int n0 = 0; n0 < N; n0 += nc) {   // chunks accumulate in image order: fixed summation order
    const int n = N - n0 < nc ? N - n0 : nc;
    if (int rc = wino_wgrad_pass(dy + (size_t)n0 * Cout * H * W, x + (size_t)n0 * Cin * H * W, dw, db, n, Cin, Cout, H, W,
                                 accumulate || n0 > 0, ws, ws_floats, s)) return rc;
  }
  return 0;
}

static int wino_wgrad_pass(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Cout, int H, int W,
                           int accumulate, float* ws, size_t ws_floats, hipStream_t s) {
  UMPR_REQUIRE((H % 2) == 0 && (W % 2) == 0, "winograd wgrad: odd map %dx%d", H, W);
  UMPR_REQUIRE(ws_floats >= umpr_wino_wgrad_ws_floats(N, Cin, Cout, H, W), "winograd wgrad: workspace too small");
  const WinoWgradGeom g = wino_wgrad_geom(N, Cin, Cout, H, W);
  float* Gy = ws;
  float* V = Gy + (size_t)g.planes * g.Mpad * g.Tpad;
  float* P = V + (size_t)g.planes * g.Cpad * g.Tpad;
  float* BP = P + (size_t)g.planes * g.splits * g.Mpad * g.Cpad;     // per-wave bias partial sums of the dy transform
  const bool f4 = g.planes == 36;
  // rows m >= Cout of Gy are never written: every P element is a dot product of ONE Gy row with ONE V row, so garbage
  // stays in rows of P that the finish kernel does not read.
  const bool edge = f4 && ((H % 4) != 0 || (W % 4) != 0);
  static const bool bias_fuse = umpr_env_on("UMPR_WINO_BIAS_FUSE");     // 0: the separate pass over dy
  const bool fused_bias = db && f4 && !edge && bias_fuse && (g.Tpad % 64) == 0;
  if (f4 && edge)
    wino4_dy_kernel<true><<<nblk((long)Cout * g.Tpad, 16384), 256, 0, s>>>(dy, Gy, N, Cout, H, W, g.Tpad, g.Tpad, g.Mpad, wino4_consts(), nullptr);
  else if (f4)
    wino4_dy_kernel<false><<<nblk((long)Cout * g.Tpad, 16384), 256, 0, s>>>(dy, Gy, N, Cout, H, W, g.Tpad, g.Tpad, g.Mpad, wino4_consts(),
                                                                            fused_bias ? BP : nullptr);
  else if ((W / 2) % 2 == 0)
    wino_dy_pair_kernel<<<nblk((long)Cout * g.Tpad / 2, 16384), 256, 0, s>>>(dy, Gy, N, Cout, H, W, g.Tpad, g.Tpad, g.Mpad);
  else
    wino_dy_kernel<<<nblk((long)Cout * g.Tpad, 16384), 256, 0, s>>>(dy, Gy, N, Cout, H, W, g.Tpad, g.Tpad, g.Mpad);
  UMPR_LAUNCH_CHECK("wino_dy");
  if (t_v_slot) {   // the forward pass left this layer's transformed input in the caller's slot
    const size_t vfl = umpr_wino_v_floats(N, Cin, Cout, H, W);
    UMPR_REQUIRE(f4 && !edge && vfl == (size_t)g.planes * g.Cpad * g.Tpad && t_v_slot_floats >= vfl,
                 "winograd wgrad: the V slot does not match this layer (%d -> %d at %dx%d)", Cin, Cout, H, W);
    V = t_v_slot;
  } else if (f4 && edge)
    wino4_input_kernel<true, false><<<nblk((long)g.Cpad * g.Tpad, 16384), 256, 0, s>>>(x, V, N, Cin, g.Cpad, H, W, g.Tpad, g.Tpad, wino4_consts(), nullptr, nullptr);
  else if (f4)
    wino4_input_kernel<false, false><<<nblk((long)g.Cpad * g.Tpad, 16384), 256, 0, s>>>(x, V, N, Cin, g.Cpad, H, W, g.Tpad, g.Tpad, wino4_consts(), nullptr, nullptr);
  else if ((W / 2) % 2 == 0)
    wino_input_pair_kernel<<<nblk((long)g.Cpad * g.Tpad / 2, 16384), 256, 0, s>>>(x, V, N, Cin, g.Cpad, H, W, g.Tpad, g.Tpad);
  else
    wino_input_kernel<<<nblk((long)g.Cpad * g.Tpad, 16384), 256, 0, s>>>(x, V, N, Cin, g.Cpad, H, W, g.Tpad, g.Tpad);
  UMPR_LAUNCH_CHECK("wino_input(wgrad)");
  WinoWgradParams p{Gy, V, P, g.MT, g.CT, g.Mpad, g.Cpad, g.Tpad, g.stages, g.stages_per_split, g.planes};
  {
    UmprProfScope prof(UMPR_K_WINO_WGRAD_GEMM, 2.0 * g.planes * (double)Cout * Cin * g.T, s);
    if (g.bnc == 64) wino_wgrad_gemm_kernel<1><<<(unsigned)(g.planes * g.MT * g.CT * g.splits), 256, 0, s>>>(p);
    else wino_wgrad_gemm_kernel<2><<<(unsigned)(g.planes * g.MT * g.CT * g.splits), 256, 0, s>>>(p);
  }
  UMPR_LAUNCH_CHECK("wino_wgrad_gemm");
  if (f4)
    wino4_wgrad_finish_kernel<<<(unsigned)((long)Cout * ((Cin + 63) / 64)), 256, 0, s>>>(P, g.splits, Cout, Cin, g.Mpad, g.Cpad, dw,
                                                                                     accumulate, wino4_consts());
  else if (g.splits >= 8)
    wino_wgrad_finish_wide_kernel<<<(unsigned)((long)Cout * ((Cin + 63) / 64)), 256, 0, s>>>(P, g.splits, Cout, Cin, g.Mpad, g.Cpad,
                                                                                         dw, accumulate);
  else
    wino_wgrad_finish_kernel<<<nblk((long)Cout * Cin, 4096), 256, 0, s>>>(P, g.splits, Cout, Cin, g.Mpad, g.Cpad, dw, accumulate);
  UMPR_LAUNCH_CHECK("wino_wgrad_finish");
  if (fused_bias) {
    wino_bias_sum_waves_kernel<<<(unsigned)((Cout + 3) / 4), 256, 0, s>>>(BP, db, Cout, g.Tpad / 64, accumulate);
    UMPR_LAUNCH_CHECK("wino_bias_sum_waves");
  } else if (db) {  // P is free again after the finish kernel (same stream): N * Cout partial sums fit in it
    UMPR_REQUIRE((size_t)N * Cout <= (size_t)g.splits * g.planes * g.Mpad * g.Cpad && ((long)H * W) % 4 == 0,
                 "winograd wgrad: bias-gradient scratch");
    wino_bias_part_wave_kernel<<<(unsigned)(((long)N * Cout + 3) / 4), 256, 0, s>>>(dy, P, (long)H * W, (long)N * Cout);
    UMPR_LAUNCH_CHECK("wino_bias_part");
    wino_bias_sum_kernel<<<(Cout + 255) / 256, 256, 0, s>>>(P, db, N, Cout, accumulate);
    UMPR_LAUNCH_CHECK("wino_bias_sum");
  }
  return 0;
}
