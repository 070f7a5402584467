// VGG16 3x3 / stride 1 / pad 1 convolution for gfx950 in exact fp32 (v_mfma_f32_32x32x2_f32), NCHW in and out.
//
//  forward / dgrad : implicit GEMM  D[cout][pixel] = sum_k Wm[cout][k] * im2col(x)[k][pixel],  k = c*9 + kh*3 + kw
//                    (torch's OIHW flatten, so the forward weight matrix is the parameter itself).  dgrad is the
//                    same kernel on the flip-transposed weights Wt[cin][cout*9+tap'] = W[cout][cin][8-tap'].
//                    MFMA-A = weights (rows = output channel), MFMA-B = pixels (lanes run along x), so loads of x and
//                    stores of y are 128-B row segments.  Epilogue: +bias, ReLU; or (dgrad) * [mask_src > 0].
//  wgrad           : per workgroup a 64(cout) x 64(cin) x 9(tap) tile, K = pixels; the gz tile and the x halo patch
//                    of a 32-pixel segment are staged once in LDS and serve all nine taps (9 accumulator tiles per
//                    wave).  Split-K over pixel segments into slabs [split][tap][cout][cin] + deterministic reduce.
#include "umpr_common.h"
#include "umpr_internal.h"
#include "umpr_tiles.h"

namespace {

struct ConvParams {
  const float* x;     // [N][C][H][W]
  const float* wm;    // [Cout][C*9]
  const float* bias;  // [Cout] or null
  const float* mask;  // [N][Cout][H][W] or null: out *= (mask > 0)
  float* y;           // [N][Cout][H][W]
  int N, C, H, W, Cout, relu;
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv3x3_igemm_kernel(ConvParams p) {
  using LA = TileRegs<BM, true>;
  constexpr int LDA = LA::LD;
  constexpr int LDB = BN;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KROWS = 256 / BN;       // k rows staged per pass
  constexpr int NPASS = BK / KROWS;     // scalar im2col loads per thread per stage
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int HW = p.H * p.W;
  const long NP = (long)p.N * HW;
  const long p0 = (long)blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const int K = p.C * 9;
  const int vecA = (K & 3) == 0;

  // this thread's im2col pixel (fixed for the whole K loop)
  const int pl = tid % BN, krow = tid / BN;
  const long pp = p0 + pl;
  const bool pvalid = pp < NP;
  int tapmask = 0;
  long xbase = 0;
  if (pvalid) {
    const int n = (int)(pp / HW), yx = (int)(pp % HW);
    const int y = yx / p.W, x = yx % p.W;
    xbase = (long)n * p.C * HW + yx;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) tapmask |= 1 << t;
    }
  }

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
  float rb[NPASS];
  auto load_b = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
      const int k = k0 + j * KROWS + krow;
      const int c = k / 9, t = k - c * 9;
      float v = 0.f;
      if (k < K && ((tapmask >> t) & 1)) v = p.x[xbase + (long)c * HW + (t / 3 - 1) * p.W + (t % 3 - 1)];
      rb[j] = v;
    }
  };
  auto store_b = [&](float* S) {
#pragma unroll
    for (int j = 0; j < NPASS; ++j) S[(j * KROWS + krow) * LDB + pl] = rb[j];
  };

  const int nt = (K + BK - 1) / BK;
  ra.load(p.wm, K, nullptr, m0, p.Cout, 0, K, vecA, tid);
  load_b(0);
  ra.store(As[0], tid);
  store_b(Bs[0]);
  __syncthreads();
  const int l31 = lane & 31, kh = lane >> 5;
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      ra.load(p.wm, K, nullptr, m0, p.Cout, (t + 1) * BK, K, vecA, tid);
      load_b((t + 1) * BK);
    }
    const float* as = As[cur] + wm * WTM + l31;
    const float* bs = Bs[cur] + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[(2 * kk + kh) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[(2 * kk + kh) * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
    if (((t + 1) & (KFLUSH - 1)) == 0 || t + 1 == nt) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
    if (t + 1 < nt) {
      ra.store(As[cur ^ 1], tid);
      store_b(Bs[cur ^ 1]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long pq = p0 + wn * WTN + j * 32 + l31;
    if (pq >= NP) continue;
    const int n = (int)(pq / HW), yx = (int)(pq % HW);
    const long obase = (long)n * p.Cout * HW + yx;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * WTM + i * 32 + mfma_row(r, lane);
        if (co < p.Cout) {
          float v = tot[i][j][r];
          if (p.bias) v += p.bias[co];
          if (p.relu) v = fmaxf(v, 0.f);
          const long o = obase + (long)co * HW;
          if (p.mask) v = p.mask[o] > 0.f ? v : 0.f;
          p.y[o] = v;
        }
      }
    }
  }
}

// Wt[cin][cout*9 + t] = W[cout][cin][8 - t]
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % 9);
    const long r = i / 9;
    const int co = (int)(r % Cout), ci = (int)(r / Cout);
    wt[i] = w[((long)co * Cin + ci) * 9 + (8 - t)];
  }
}

// ------------------------------------------------------------------------------------------------------------
// wgrad
struct WgradParams {
  const float* gz;  // [N][Cout][H][W]
  const float* x;   // [N][Cin][H][W]
  float* slab;      // [splits][9][Cout][Cin]
  float* bslab;     // [splits][Cout] or null
  int N, Cin, Cout, H, W;
  int R, CW;        // segment = R rows x CW cols (R*CW <= 32)
  int segs_y, segs_x, nsegs, segs_per_split;
};

constexpr int WG_LDG = 33;    // Gs[64][33]
constexpr int WG_PLMAX = 103;  // Xs[64][PL], PL = (R+2)*(CW+2) (+1 if even)

__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgradParams p) {
  __shared__ float Gs[2][64 * WG_LDG];
  __shared__ float Xs[2][64 * WG_PLMAX];
  __shared__ int pxoff[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  const int split = blockIdx.z;
  const int RW = p.CW + 2;
  const int patch = (p.R + 2) * RW;
  const int PL = patch | 1;
  const int npx = p.R * p.CW;
  const int HW = p.H * p.W;
  if (tid < 32) pxoff[tid] = tid < npx ? (tid / p.CW) * RW + (tid % p.CW) : 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) bsum[v] = 0.f;

  const int sbeg = split * p.segs_per_split;
  const int send = min(p.nsegs, sbeg + p.segs_per_split);
  const int xs_total = 64 * patch;
  constexpr int XV = (64 * 102 + 255) / 256;  // 26 loads cover the largest patch
  float rg[8];
  float rx[XV];

  auto load = [&](int seg) {
    const int sx = seg % p.segs_x;
    const int sy = (seg / p.segs_x) % p.segs_y;
    const int n = seg / (p.segs_x * p.segs_y);
    const int y0 = sy * p.R, x0 = sx * p.CW;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int e = tid + v * 256;
      const int co = e >> 5, px = e & 31;
      const int r = px / p.CW, c = px - r * p.CW;
      float val = 0.f;
      if (px < npx && co0 + co < p.Cout && y0 + r < p.H && x0 + c < p.W)
        val = p.gz[((long)n * p.Cout + co0 + co) * HW + (y0 + r) * p.W + x0 + c];
      rg[v] = val;
    }
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int e = tid + v * 256;
      float val = 0.f;
      if (e < xs_total) {
        const int ci = e / patch, q = e - ci * patch;
        const int rr = q / RW, cc = q - rr * RW;
        const int yy = y0 - 1 + rr, xx = x0 - 1 + cc;
        if (ci0 + ci < p.Cin && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
          val = p.x[((long)n * p.Cin + ci0 + ci) * HW + yy * p.W + xx];
      }
      rx[v] = val;
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int e = tid + v * 256;
      Gs[buf][(e >> 5) * WG_LDG + (e & 31)] = rg[v];
      if (p.bslab && blockIdx.x == 0) bsum[v] += rg[v];
    }
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int e = tid + v * 256;
      if (e < xs_total) {
        const int ci = e / patch, q = e - ci * patch;
        Xs[buf][ci * PL + q] = rx[v];
      }
    }
  };

  if (sbeg < send) {
    load(sbeg);
    store(0);
  }
  __syncthreads();
  for (int s = sbeg; s < send; ++s) {
    const int cur = (s - sbeg) & 1;
    if (s + 1 < send) load(s + 1);
    const float* gs = Gs[cur] + (wm * 32 + l31) * WG_LDG;
    const float* xs = Xs[cur] + (wn * 32 + l31) * PL;
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int px = 2 * kk + kh;
      const float a = gs[px];
      const float* xp = xs + pxoff[px];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float b = xp[(t / 3) * RW + (t % 3)];
        acc[t] = mfma32(a, b, acc[t]);
      }
    }
    if (s + 1 < send) store(cur ^ 1);
    __syncthreads();
  }

  // slab[split][t][co][ci]
#pragma unroll
  for (int t = 0; t < 9; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + mfma_row(r, lane);
      const int ci = ci0 + wn * 32 + l31;
      if (co < p.Cout && ci < p.Cin) p.slab[(((long)split * 9 + t) * p.Cout + co) * p.Cin + ci] = acc[t][r];
    }
  }
  if (p.bslab && blockIdx.x == 0) {
    // thread's 8 partial sums belong to couts (tid>>5)+8v; reduce over the 32 lanes sharing a cout
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      float sv = bsum[v];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sv += __shfl_xor(sv, o, 64);
      const int co = co0 + (tid >> 5) + 8 * v;
      if ((tid & 31) == 0 && co < p.Cout) p.bslab[(long)split * p.Cout + co] = sv;
    }
  }
}

// dW[co][ci][t] (+)= sum_split slab[split][t][co][ci];  db[co] (+)= sum_split bslab[split][co]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab, int splits,
                                    int Cout, int Cin, float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const long per = (long)9 * Cout * Cin;
  const long total = per + (bslab ? Cout : 0);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < per) {
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
      int s = 0;
      for (; s + 3 < splits; s += 4) {
        v0 += slab[(long)s * per + i]; v1 += slab[(long)(s + 1) * per + i];
        v2 += slab[(long)(s + 2) * per + i]; v3 += slab[(long)(s + 3) * per + i];
      }
      for (; s < splits; ++s) v0 += slab[(long)s * per + i];
      const float v = (v0 + v1) + (v2 + v3);
      const int ci = (int)(i % Cin);
      const long r = i / Cin;
      const int co = (int)(r % Cout), t = (int)(r / Cout);
      float* d = dw + ((long)co * Cin + ci) * 9 + t;
      *d = accumulate ? *d + v : v;
    } else {
      const int co = (int)(i - per);
      float v = 0.f;
      for (int s = 0; s < splits; ++s) v += bslab[(long)s * Cout + co];
      db[co] = accumulate ? db[co] + v : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// 2x2/2 max pooling, and its backward fused with the ReLU mask of the (post-ReLU) pooled input
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = planes * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const long r = i / Wo;
    const int yo = (int)(r % Ho);
    const long pl = r / Ho;
    const float* s = x + (pl * H + 2 * yo) * W + 2 * xo;
    const float2 a = *reinterpret_cast<const float2*>(s);
    const float2 b = *reinterpret_cast<const float2*>(s + W);
    y[i] = fmaxf(fmaxf(a.x, a.y), fmaxf(b.x, b.y));
  }
}

// gx = route(gy) to the first maximum of each window, zeroed where the maximum is not > 0 (ReLU backward of
// the activation that fed the pool)
__global__ void maxpool2_bwd_relu_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                         float* __restrict__ gx, long planes, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const long total = planes * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const long r = i / Wo;
    const int yo = (int)(r % Ho);
    const long pl = r / Ho;
    const long o = (pl * H + 2 * yo) * W + 2 * xo;
    const float2 a = *reinterpret_cast<const float2*>(x + o);
    const float2 b = *reinterpret_cast<const float2*>(x + o + W);
    int arg = 0;
    float m = a.x;
    if (a.y > m) { m = a.y; arg = 1; }
    if (b.x > m) { m = b.x; arg = 2; }
    if (b.y > m) { m = b.y; arg = 3; }
    const float g = m > 0.f ? gy[i] : 0.f;
    float2 ga = make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f);
    float2 gb = make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f);
    *reinterpret_cast<float2*>(gx + o) = ga;
    *reinterpret_cast<float2*>(gx + o + W) = gb;
  }
}

}  // namespace

// ---- internal host entry points ------------------------------------------------------------------------------
int umpr_conv3x3_igemm(const float* x, const float* wm, const float* bias, const float* mask, float* y, int N, int C,
                       int H, int W, int Cout, int relu, hipStream_t s) {
  UMPR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3: bad shape");
  ConvParams p{x, wm, bias, mask, y, N, C, H, W, Cout, relu};
  const long NP = (long)N * H * W;
  UmprProfScope prof(UMPR_K_CONV_IGEMM, 2.0 * NP * Cout * C * 9, s);
  if (Cout <= 64) {
    dim3 grid(cdiv(NP, 128), cdiv(Cout, 64));
    conv3x3_igemm_kernel<64, 128><<<grid, 256, 0, s>>>(p);
  } else {
    dim3 grid(cdiv(NP, 128), cdiv(Cout, 128));
    conv3x3_igemm_kernel<128, 128><<<grid, 256, 0, s>>>(p);
  }
  UMPR_LAUNCH_CHECK("conv3x3_igemm");
  return 0;
}

int umpr_conv3x3_flip_transpose(const float* w, float* wt, int Cout, int Cin, hipStream_t s) {
  const long total = (long)Cout * Cin * 9;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  flip_transpose_kernel<<<blocks, 256, 0, s>>>(w, wt, Cout, Cin);
  UMPR_LAUNCH_CHECK("flip_transpose");
  return 0;
}

size_t umpr_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W) {
  const int CW = W >= 32 ? 32 : W;
  const int R = W >= 32 ? 1 : (32 / W);
  const int nsegs = N * cdiv(H, R) * cdiv(W, CW);
  const int tiles = cdiv(Cout, 64) * cdiv(Cin, 64);
  int splits = cdiv(1024, tiles);
  if (splits > nsegs) splits = nsegs;
  return (size_t)splits * ((size_t)9 * Cout * Cin + Cout) * sizeof(float);
}

int umpr_conv3x3_wgrad(const float* gz, const float* x, float* dw, float* db, int N, int Cin, int Cout, int H, int W,
                       int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
  WgradParams p;
  p.gz = gz; p.x = x; p.N = N; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
  p.CW = W >= 32 ? 32 : W;
  p.R = W >= 32 ? 1 : (32 / W);
  UMPR_REQUIRE(p.R * p.CW <= 32 && (p.R + 2) * (p.CW + 2) <= 102, "wgrad: unsupported width %d", W);
  p.segs_y = cdiv(H, p.R);
  p.segs_x = cdiv(W, p.CW);
  p.nsegs = N * p.segs_y * p.segs_x;
  const int tiles = cdiv(Cout, 64) * cdiv(Cin, 64);
  int splits = cdiv(1024, tiles);
  if (splits > p.nsegs) splits = p.nsegs;
  const size_t per = ((size_t)9 * Cout * Cin + Cout) * sizeof(float);
  if ((size_t)splits * per > ws_bytes) splits = (int)(ws_bytes / per);
  UMPR_REQUIRE(splits >= 1, "wgrad: workspace too small (%zu bytes)", ws_bytes);
  p.segs_per_split = cdiv(p.nsegs, splits);
  splits = cdiv(p.nsegs, p.segs_per_split);
  p.slab = ws;
  p.bslab = db ? ws + (size_t)splits * 9 * Cout * Cin : nullptr;
  dim3 grid(cdiv(Cin, 64), cdiv(Cout, 64), splits);
  {
    UmprProfScope prof(UMPR_K_CONV_WGRAD, 2.0 * N * H * W * Cout * Cin * 9, s);
    conv3x3_wgrad_kernel<<<grid, 256, 0, s>>>(p);
  }
  UMPR_LAUNCH_CHECK("conv3x3_wgrad");
  const long total = (long)9 * Cout * Cin + (db ? Cout : 0);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  wgrad_reduce_kernel<<<blocks, 256, 0, s>>>(p.slab, p.bslab, splits, Cout, Cin, dw, db, accumulate);
  UMPR_LAUNCH_CHECK("wgrad_reduce");
  return 0;
}

int umpr_maxpool2_fwd_impl(const float* x, float* y, long planes, int H, int W, hipStream_t s) {
  UMPR_REQUIRE((H % 2) == 0 && (W % 2) == 0, "maxpool2: odd extent %dx%d", H, W);
  const long total = planes * (H / 2) * (W / 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  maxpool2_fwd_kernel<<<blocks, 256, 0, s>>>(x, y, planes, H, W);
  UMPR_LAUNCH_CHECK("maxpool2_fwd");
  return 0;
}

int umpr_maxpool2_bwd_relu_impl(const float* x, const float* gy, float* gx, long planes, int H, int W, hipStream_t s) {
  UMPR_REQUIRE((H % 2) == 0 && (W % 2) == 0, "maxpool2: odd extent %dx%d", H, W);
  const long total = planes * (H / 2) * (W / 2);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  maxpool2_bwd_relu_kernel<<<blocks, 256, 0, s>>>(x, gy, gx, planes, H, W);
  UMPR_LAUNCH_CHECK("maxpool2_bwd_relu");
  return 0;
}
