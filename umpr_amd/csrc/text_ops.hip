// Small fused text-path kernels of UMPR for gfx950 (all HBM/latency bound, one wave per sentence or one
// workgroup per sample; reductions by wavefront shuffles, fixed summation order = bitwise reproducible):
//   S-Net pooling (src/model.py:75-80), C-Net head (src/model.py:118-125), control gate incl. SS-Net
//   (src/model.py:186-197,142-143), visual head + fusion + losses (src/model.py:218-228,268-277).
#include "umpr_common.h"
#include "umpr_internal.h"

namespace {

constexpr int D = 128;  // 2u
constexpr int AT = 64;  // self_atte_size

__device__ float block_sum4(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// -------------------------------------------------------------------------------------------------- S-Net
struct SnetFwdParams {
  const float* X;    // [B][S][L][128]
  const float* U;    // [B][S][L][64] = tanh(X Ms^T)
  const float* Ws;   // [64]
  const float* word_soft; int wl;  // [B][S][wl]
  float* P;          // [B][S][L]   softmax over tokens (saved)
  float* wsum;       // [B][S]
  float* self_atte;  // [B][S][128]
  float* senti; long ld_senti;  // senti[b*ld + c]
  int S, L;
};

// one wave per sentence (B*S waves over the grid); the per-sample sum over sentences is snet_senti_kernel
// NL = ceil(L / 64): lane l % 64 holds position l in slot l / 64 (L <= 64 * NL; the reference's max_sent_length is 20, but
// review_level='review' makes whole reviews the "sentences", src/dataset.py:24 - supported up to 256 tokens)
template <int NL>
__global__ __launch_bounds__(256) void snet_pool_fwd_kernel(SnetFwdParams p, long nsent) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long sent = (long)blockIdx.x * 4 + wave;
  if (sent >= nsent) return;
  const float w = p.Ws[lane];
  const float* X = p.X + sent * p.L * D;
  const float* U = p.U + sent * p.L * AT;
  float e[NL];  // lane l % 64 holds e[l / 64]
#pragma unroll
  for (int k = 0; k < NL; ++k) e[k] = -INFINITY;
  int l = 0;
  for (; l + 3 < p.L; l += 4) {   // four independent reductions in flight (l .. l+3 share a slot: 64 is a multiple of 4)
    const float v0 = wave_sum(w * U[l * AT + lane]);
    const float v1 = wave_sum(w * U[(l + 1) * AT + lane]);
    const float v2 = wave_sum(w * U[(l + 2) * AT + lane]);
    const float v3 = wave_sum(w * U[(l + 3) * AT + lane]);
    const int ll = l & 63;
#pragma unroll
    for (int k = 0; k < NL; ++k)
      if (k == (l >> 6)) e[k] = lane == ll ? v0 : lane == ll + 1 ? v1 : lane == ll + 2 ? v2 : lane == ll + 3 ? v3 : e[k];
  }
  for (; l < p.L; ++l) {
    const float v = wave_sum(w * U[l * AT + lane]);
#pragma unroll
    for (int k = 0; k < NL; ++k)
      if (k == (l >> 6) && lane == (l & 63)) e[k] = v;
  }
  float mx = e[0];
#pragma unroll
  for (int k = 1; k < NL; ++k) mx = fmaxf(mx, e[k]);
  const float m = wave_max(mx);
  float ex[NL], zs = 0.f;
#pragma unroll
  for (int k = 0; k < NL; ++k) { ex[k] = 64 * k + lane < p.L ? expf(e[k] - m) : 0.f; zs += ex[k]; }
  const float z = wave_sum(zs);
  float pr[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    pr[k] = ex[k] / z;
    if (64 * k + lane < p.L) p.P[sent * p.L + 64 * k + lane] = pr[k];
  }
  float a0 = 0.f, a1 = 0.f;
  for (l = 0; l < p.L; ++l) {
    float src = pr[0];
#pragma unroll
    for (int k = 1; k < NL; ++k) src = (l >> 6) == k ? pr[k] : src;
    const float pl = __shfl(src, l & 63, 64);
    a0 += pl * X[l * D + lane];
    a1 += pl * X[l * D + 64 + lane];
  }
  p.self_atte[sent * D + lane] = a0;
  p.self_atte[sent * D + 64 + lane] = a1;
  float ws = 0.f;
  for (int i = lane; i < p.wl; i += 64) ws += p.word_soft[sent * p.wl + i];
  ws = wave_sum(ws);
  if (lane == 0) p.wsum[sent] = ws;
}

// senti[b] = sum_s wsum[b][s] * self_atte[b][s]   (fixed order)
__global__ __launch_bounds__(128) void snet_senti_kernel(const float* __restrict__ wsum, const float* __restrict__ self_atte,
                                                         float* __restrict__ senti, long ld_senti, int S) {
  const int b = blockIdx.x, d = threadIdx.x;
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += wsum[(long)b * S + s] * self_atte[((long)b * S + s) * D + d];
  senti[(long)b * ld_senti + d] = v;
}

struct SnetBwdParams {
  const float* X; const float* U; const float* Ws; const float* P; const float* wsum; const float* self_atte;
  const float* d_senti; long ld_ds;   // [B][128] strided rows
  const float* d_self_atte;           // [B][S][128] or null
  float* dX;                          // [B][S][L][128]  (= p[l] * d_sa; the GEMM dPre Ms accumulates onto it)
  float* dPre;                        // [B][S][L][64]
  float* dWs_part;                    // [B][64]
  float* d_word_soft; int wl;         // [B][S][wl] or null
  int S, L;
};

// one wave per sentence; dWs_part[workgroup][64] = the four waves' partial sums in wave order
template <int NL>
__global__ __launch_bounds__(256) void snet_pool_bwd_kernel(SnetBwdParams p, long nsent) {
  __shared__ float part[4][AT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long sent = (long)blockIdx.x * 4 + wave;
  float dws = 0.f;
  if (sent < nsent) {
    const long b = sent / p.S;
    const float w = p.Ws[lane];
    const float ds0 = p.d_senti[b * p.ld_ds + lane], ds1 = p.d_senti[b * p.ld_ds + 64 + lane];
    const float* X = p.X + sent * p.L * D;
    const float* U = p.U + sent * p.L * AT;
    const float ws = p.wsum[sent];
    float g0 = ws * ds0, g1 = ws * ds1;  // d self_atte
    if (p.d_self_atte) { g0 += p.d_self_atte[sent * D + lane]; g1 += p.d_self_atte[sent * D + 64 + lane]; }
    const float dwsum = wave_sum(p.self_atte[sent * D + lane] * ds0 + p.self_atte[sent * D + 64 + lane] * ds1);
    if (p.d_word_soft)
      for (int i = lane; i < p.wl; i += 64) p.d_word_soft[sent * p.wl + i] = dwsum;
    float pr[NL], dp[NL];  // lane l % 64 holds position l in slot l / 64
#pragma unroll
    for (int k = 0; k < NL; ++k) { pr[k] = 64 * k + lane < p.L ? p.P[sent * p.L + 64 * k + lane] : 0.f; dp[k] = 0.f; }
    int l = 0;
    for (; l + 3 < p.L; l += 4) {   // four independent reductions in flight
      const float v0 = wave_sum(X[l * D + lane] * g0 + X[l * D + 64 + lane] * g1);
      const float v1 = wave_sum(X[(l + 1) * D + lane] * g0 + X[(l + 1) * D + 64 + lane] * g1);
      const float v2 = wave_sum(X[(l + 2) * D + lane] * g0 + X[(l + 2) * D + 64 + lane] * g1);
      const float v3 = wave_sum(X[(l + 3) * D + lane] * g0 + X[(l + 3) * D + 64 + lane] * g1);
      const int ll = l & 63;
#pragma unroll
      for (int k = 0; k < NL; ++k)
        if (k == (l >> 6)) dp[k] = lane == ll ? v0 : lane == ll + 1 ? v1 : lane == ll + 2 ? v2 : lane == ll + 3 ? v3 : dp[k];
    }
    for (; l < p.L; ++l) {
      const float v = wave_sum(X[l * D + lane] * g0 + X[l * D + 64 + lane] * g1);
#pragma unroll
      for (int k = 0; k < NL; ++k)
        if (k == (l >> 6) && lane == (l & 63)) dp[k] = v;
    }
    float pd = 0.f;
#pragma unroll
    for (int k = 0; k < NL; ++k) pd += pr[k] * dp[k];
    const float dot = wave_sum(pd);
    float de[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) de[k] = pr[k] * (dp[k] - dot);
    for (l = 0; l < p.L; ++l) {
      float sp = pr[0], sd = de[0];
#pragma unroll
      for (int k = 1; k < NL; ++k) { sp = (l >> 6) == k ? pr[k] : sp; sd = (l >> 6) == k ? de[k] : sd; }
      const float pl = __shfl(sp, l & 63, 64), del = __shfl(sd, l & 63, 64);
      float* dx = p.dX + (sent * p.L + l) * D;
      dx[lane] = pl * g0; dx[64 + lane] = pl * g1;
      const float u = U[l * AT + lane];
      p.dPre[(sent * p.L + l) * AT + lane] = w * del * (1.f - u * u);
      dws += del * u;
    }
  }
  part[wave][lane] = dws;
  __syncthreads();
  if (tid < AT) p.dWs_part[(long)blockIdx.x * AT + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// -------------------------------------------------------------------------------------------------- C-Net
// Conv1d as a GEMM over a sliding-window view of X (UmprGemm::winA / winB): no im2col / col2im buffers.  The window's
// column order is (tap j, channel c), the parameter's is (c, j): three small re-orderings of the [KC][D][KS] weight.
//   mode 0: Wp [kc][j*D + c]   = Wc[kc][c*KS + j]                 forward:  Y = win(X) Wp^T
//   mode 1: Wq [j*KC + kc][c]  = Wc[kc][c*KS + (KS-1-j)]          backward: dX = win(dY) Wq
//   mode 2: dWc[kc][c*KS + j] (+)= dWp[kc][j*D + c]               backward: dWp = dY^T win(X)
__global__ void cnet_weight_order_kernel(const float* __restrict__ src, float* __restrict__ dst, int KC, int Dc, int KS,
                                         int mode, int accumulate) {
  const int total = KC * Dc * KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i % KS, c = (i / KS) % Dc, kc = i / (KS * Dc);   // i = index into Wc / dWc
    if (mode == 0) dst[(long)kc * Dc * KS + j * Dc + c] = src[i];
    else if (mode == 1) dst[((long)(KS - 1 - j) * KC + kc) * Dc + c] = src[i];
    else { const float v = src[(long)kc * Dc * KS + j * Dc + c]; dst[i] = accumulate ? dst[i] + v : v; }
  }
}

struct CnetHeadFwdParams {
  const float* Y;    // [B][S][L][KC] = relu(conv)
  const float* Wl;   // [V][KC]
  const float* bl;   // [V]
  float thr;
  float* cmax; int* argl;  // [B][S][KC]
  float* sp;               // [B][S][V]  sigmoid before threshold
  float* view_p;           // [B][S][V]
  float* final_;           // [B][V]
  int S, L, KC, V;
  int Lout;                // valid conv positions: L + 2 pad - KS + 1 (= L for odd KS, L - 1 for even, like nn.Conv1d)
};

// one wave per sentence over the whole grid (was: one workgroup per sample walking S / 4 sentences per wave - 64
// workgroups reading the 20 MB of Y at 0.2 TB/s); the per-sample sum over sentences is cnet_final_kernel
__global__ __launch_bounds__(256) void cnet_head_fwd_kernel(CnetHeadFwdParams p, long nsent) {
  extern __shared__ float sm[];  // cm[4][KC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long sent = (long)blockIdx.x * 4 + wave;
  if (sent >= nsent) return;
  float* cm = sm + wave * p.KC;
  const float* Y = p.Y + sent * p.L * p.KC;
  for (int k = lane; k < p.KC; k += 64) {
    float m = Y[k]; int a = 0;
    for (int l = 1; l < p.Lout; ++l) {
      const float y = Y[l * p.KC + k];
      if (y > m) { m = y; a = l; }
    }
    cm[k] = m;
    p.cmax[sent * p.KC + k] = m; p.argl[sent * p.KC + k] = a;
  }
  // cm is written and read by this wave only: a wave executes in lock-step, no barrier needed
  for (int v = 0; v < p.V; ++v) {
    float a = 0.f;
    for (int k = lane; k < p.KC; k += 64) a += p.Wl[v * p.KC + k] * cm[k];
    a = wave_sum(a) + p.bl[v];
    const float sg = sigmoidf_(a);
    const float vp = sg < p.thr ? 0.f : sg;
    if (lane == 0) { p.sp[sent * p.V + v] = sg; p.view_p[sent * p.V + v] = vp; }
  }
}

// final[b][v] = sum_s view_p[b][s][v]^2   (fixed order)
__global__ void cnet_final_kernel(const float* __restrict__ view_p, float* __restrict__ final_, int S, int V) {
  const int b = blockIdx.x;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) { const float vp = view_p[((long)b * S + s) * V + v]; a += vp * vp; }
    final_[(long)b * V + v] = a;
  }
}

struct CnetHeadBwdParams {
  const float* cmax; const int* argl; const float* sp; const float* view_p; const float* Wl;
  const float* d_final;   // [B][V] or null
  const float* d_view_p;  // [B][S][V] or null
  float* dY;              // [B][S][L][KC] pre-zeroed
  float* dWl_part;        // [B][V][KC]
  float* dbl_part;        // [B][V]
  int S, L, KC, V;
};

__global__ __launch_bounds__(256) void cnet_head_bwd_kernel(CnetHeadBwdParams p) {
  extern __shared__ float sm[];  // dW[4][V*KC], db[4][V], dsig[4][V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int VK = p.V * p.KC;
  float* dW = sm + wave * VK;
  float* db = sm + 4 * VK + wave * p.V;
  float* dsg = sm + 4 * VK + 4 * p.V + wave * p.V;
  for (int i = lane; i < VK; i += 64) dW[i] = 0.f;
  for (int v = lane; v < p.V; v += 64) db[v] = 0.f;
  __syncthreads();
  for (int s0 = 0; s0 < p.S; s0 += 4) {
    const int s = s0 + wave;
    const bool on = s < p.S;
    const long sent = (long)b * p.S + s;
    if (on) {
      for (int v = lane; v < p.V; v += 64) {
        const float vp = p.view_p[sent * p.V + v], sg = p.sp[sent * p.V + v];
        float g = 0.f;
        if (p.d_view_p) g += p.d_view_p[sent * p.V + v];
        if (p.d_final) g += 2.f * vp * p.d_final[(long)b * p.V + v];
        const float d = vp > 0.f ? g * sg * (1.f - sg) : 0.f;  // thresholded entries carry no gradient
        dsg[v] = d;
        db[v] += d;
      }
    }
    __syncthreads();
    if (on) {
      for (int k = lane; k < p.KC; k += 64) {
        const float cmv = p.cmax[sent * p.KC + k];
        float dc = 0.f;
        for (int v = 0; v < p.V; ++v) {
          dc += dsg[v] * p.Wl[v * p.KC + k];
          dW[v * p.KC + k] += dsg[v] * cmv;
        }
        if (cmv > 0.f) p.dY[(sent * p.L + p.argl[sent * p.KC + k]) * p.KC + k] = dc;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  for (int i = tid; i < VK; i += 256) p.dWl_part[(long)b * VK + i] = sm[i] + sm[VK + i] + sm[2 * VK + i] + sm[3 * VK + i];
  const float* dbb = sm + 4 * VK;
  for (int v = tid; v < p.V; v += 256)
    p.dbl_part[(long)b * p.V + v] = dbb[v] + dbb[p.V + v] + dbb[2 * p.V + v] + dbb[3 * p.V + v];
}

// -------------------------------------------------------------------------------------------------- gate
struct GateFwdParams {
  const float* sa;       // [B][S][128] self attention of the ui review (control S-Net)
  const float* w; const float* bias;   // SS-Net linear [128], [1]
  const float* view_p;   // [B][S][V]
  const float* c_out;    // [B][V]
  float* senti;          // [B][S]
  float* vs;             // [B][V]
  float* prefer_pos; float* prefer_neg;  // [B][V]
  int S, V;
};

__global__ __launch_bounds__(64) void gate_fwd_kernel(GateFwdParams p) {
  extern __shared__ float sm[];  // senti[S]
  const int lane = threadIdx.x, b = blockIdx.x;
  const float w0 = p.w[lane], w1 = p.w[64 + lane], bb = p.bias[0];
  for (int s = 0; s < p.S; ++s) {
    const float* sa = p.sa + ((long)b * p.S + s) * D;
    const float v = wave_sum(sa[lane] * w0 + sa[64 + lane] * w1) + bb;
    const float sg = sigmoidf_(v);
    if (lane == 0) { sm[s] = sg; p.senti[(long)b * p.S + s] = sg; }
  }
  __syncthreads();
  for (int v = lane; v < p.V; v += 64) {
    float num = 0.f, den = 0.f;
    for (int s = 0; s < p.S; ++s) {
      const float vp = p.view_p[((long)b * p.S + s) * p.V + v];
      num += sm[s] * vp * vp; den += vp * vp;
    }
    const float vsc = num / (den + 1e-4f);
    const float qp = vsc > 0.5f ? 1.f : 0.f;
    const float qpos = vsc < 0.5f ? 0.f : 4.f * (vsc - 0.5f) * (vsc - 0.5f);
    const float qneg = vsc > 0.5f ? 0.f : 4.f * (0.5f - vsc) * (0.5f - vsc);
    const float co = p.c_out[(long)b * p.V + v];
    p.vs[(long)b * p.V + v] = vsc;
    p.prefer_pos[(long)b * p.V + v] = co * qp * qpos;
    p.prefer_neg[(long)b * p.V + v] = co * (1.f - qp) * qneg;
  }
}

struct GateBwdParams {
  const float* sa; const float* w; const float* view_p; const float* c_out; const float* senti; const float* vs;
  const float* d_pp; const float* d_pn;  // [B][V]
  float* d_sa;       // [B][S][128]
  float* d_view_p;   // [B][S][V]
  float* d_c_out;    // [B][V]
  float* dw_part;    // [B][128]
  float* db_part;    // [B]
  int S, V;
};

__global__ __launch_bounds__(64) void gate_bwd_kernel(GateBwdParams p) {
  extern __shared__ float sm[];  // dnum[V], dden[V], dsenti[S]
  const int lane = threadIdx.x, b = blockIdx.x;
  float* dnum = sm; float* dden = sm + p.V; float* dse = sm + 2 * p.V;
  for (int v = lane; v < p.V; v += 64) {
    float num = 0.f, den = 0.f;
    for (int s = 0; s < p.S; ++s) {
      const float vp = p.view_p[((long)b * p.S + s) * p.V + v];
      num += p.senti[(long)b * p.S + s] * vp * vp; den += vp * vp;
    }
    den += 1e-4f;
    const float vsc = p.vs[(long)b * p.V + v];
    const float qp = vsc > 0.5f ? 1.f : 0.f;
    const float qpos = vsc < 0.5f ? 0.f : 4.f * (vsc - 0.5f) * (vsc - 0.5f);
    const float qneg = vsc > 0.5f ? 0.f : 4.f * (0.5f - vsc) * (0.5f - vsc);
    const float dqpos = vsc < 0.5f ? 0.f : 8.f * (vsc - 0.5f);
    const float dqneg = vsc > 0.5f ? 0.f : -8.f * (0.5f - vsc);
    const float co = p.c_out[(long)b * p.V + v];
    const float gpp = p.d_pp[(long)b * p.V + v], gpn = p.d_pn[(long)b * p.V + v];
    p.d_c_out[(long)b * p.V + v] = gpp * qp * qpos + gpn * (1.f - qp) * qneg;
    const float dvs = gpp * co * qp * dqpos + gpn * co * (1.f - qp) * dqneg;
    dnum[v] = dvs / den;
    dden[v] = -dvs * num / (den * den);
  }
  __syncthreads();
  for (int s = lane; s < p.S; s += 64) {
    float d = 0.f;
    const float se = p.senti[(long)b * p.S + s];
    for (int v = 0; v < p.V; ++v) {
      const float vp = p.view_p[((long)b * p.S + s) * p.V + v];
      d += dnum[v] * vp * vp;
      p.d_view_p[((long)b * p.S + s) * p.V + v] = 2.f * vp * (se * dnum[v] + dden[v]);
    }
    dse[s] = d * se * (1.f - se);
  }
  __syncthreads();
  const float w0 = p.w[lane], w1 = p.w[64 + lane];
  float dw0 = 0.f, dw1 = 0.f, dbs = 0.f;
  for (int s = 0; s < p.S; ++s) {
    const float d = dse[s];
    const float* sa = p.sa + ((long)b * p.S + s) * D;
    float* dsa = p.d_sa + ((long)b * p.S + s) * D;
    dw0 += d * sa[lane]; dw1 += d * sa[64 + lane]; dbs += d;
    dsa[lane] = d * w0; dsa[64 + lane] = d * w1;
  }
  p.dw_part[(long)b * D + lane] = dw0; p.dw_part[(long)b * D + 64 + lane] = dw1;
  if (lane == 0) p.db_part[b] = dbs;
}

// -------------------------------------------------------------------------------------------------- head
using HeadParams = UmprHead;

// pos/neg view embeddings and image embeddings: one wave per dot product of length F, over the whole grid (the single
// workgroup of head_fwd_kernel walked the 2V + BV products four at a time: 140 us on the step's critical path)
__global__ __launch_bounds__(256) void head_emb_kernel(HeadParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = p.B, V = p.V;
  const int ndot = 2 * V + B * V;
  {
    {
      const int d = blockIdx.x * 4 + wave;
      if (d >= ndot) return;
      float a = 0.f;
      if (d < 2 * V) {
        const float* src = (d < V ? p.pos_v + (long)d * p.F : p.neg_v + (long)(d - V) * p.F);
        for (int i = lane; i < p.F; i += 64) a += src[i] * p.lw[i];
      } else {
        const int bv = d - 2 * V;
        for (int i = lane; i < p.F; i += 64) {
          float m = 0.f;
          for (int q = 0; q < p.P; ++q) m += p.vgg[((long)bv * p.P + q) * p.F + i];
          a += (m / (float)p.P) * p.lw[i];
        }
      }
      a = wave_sum(a) + p.lb[0];
      if (lane == 0) {
        if (d < 2 * V) p.posneg_emb[d] = a; else p.img_emb[d - 2 * V] = a;
      }
    }
  }
}

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadParams p) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, V = p.V;
  if (V > 0) {   // posneg_emb / img_emb come from head_emb_kernel
    for (int i = tid; i < B * V; i += 256) {
      const int v = i % V;
      const float ie = p.img_emb[i];
      p.pos_match[i] = tanhf(fabsf(p.posneg_emb[v] - ie));
      p.neg_match[i] = tanhf(fabsf(p.posneg_emb[V + v] - ie));
    }
    __syncthreads();
  }
  float se = 0.f;
  for (int b = wave; b < B; b += 4) {
    float a = p.rr[(long)b * D + lane] * p.fw[lane] + p.rr[(long)b * D + 64 + lane] * p.fw[64 + lane];
    for (int v = lane; v < V; v += 64) {
      const float cc = p.c_u[b * V + v] * p.c_i[b * V + v];
      a += cc * (1.f - p.pos_match[b * V + v]) * p.fw[D + v] + cc * (1.f - p.neg_match[b * V + v]) * p.fw[D + V + v];
    }
    a = wave_sum(a) + p.fb[0];
    const float pr = fmaxf(a, 0.f);
    if (lane == 0) {
      p.z[b] = a; p.pred[b] = pr;
      const float e = pr - p.labels[b];
      se += e * e;
    }
  }
  const float loss_r = block_sum4(se, red) / (float)B;
  float lv = 0.f;
  if (V > 0) {
    // mean over the VxV matrix prefer_pos^T pos_match + prefer_neg^T neg_match (contraction over the batch)
    for (int i = tid; i < V * V; i += 256) {
      const int v1 = i / V, v2 = i % V;
      float a = 0.f;
      for (int b = 0; b < B; ++b)
        a += p.pp[b * V + v1] * p.pos_match[b * V + v2] + p.pn[b * V + v1] * p.neg_match[b * V + v2];
      lv += a;
    }
    lv = block_sum4(lv, red) / (float)(V * V);
  }
  if (tid == 0) {
    p.loss[0] = loss_r + lv * p.rate; p.loss[1] = loss_r; p.loss[2] = lv;
  }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadParams p) {
  extern __shared__ float sm[];  // dz[B], dimg[B*V], ddp[B*V], ddn[B*V], dpos[V], dneg[V]
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int B = p.B, V = p.V, FW = D + 2 * V;
  float* dz = sm; float* dimg = sm + B; float* ddps = dimg + B * V; float* ddns = ddps + B * V;
  float* dpos = ddns + B * V; float* dneg = dpos + V;
  // gridDim.x workgroups: every one repeats the O(B V) prefix in its own LDS, workgroup 0 alone writes the small outputs,
  // and the 1000-wide loop over the VGG features (64 dependent iterations per element) is sliced over the workgroups
  const bool first = blockIdx.x == 0;
  const float gl = p.d_loss[0];
  for (int b = tid; b < B; b += 256) {
    float g = gl * 2.f * (p.pred[b] - p.labels[b]) / (float)B;
    if (p.d_pred) g += p.d_pred[b];
    dz[b] = p.z[b] > 0.f ? g : 0.f;
  }
  __syncthreads();
  // fusion layer
  for (int i = tid; first && i < FW; i += 256) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) {
      float f;
      if (i < D) f = p.rr[(long)b * D + i];
      else if (i < D + V) { const int v = i - D; f = p.c_u[b * V + v] * p.c_i[b * V + v] * (1.f - p.pos_match[b * V + v]); }
      else { const int v = i - D - V; f = p.c_u[b * V + v] * p.c_i[b * V + v] * (1.f - p.neg_match[b * V + v]); }
      a += dz[b] * f;
    }
    p.d_fw[i] = a;
  }
  {
    float a = 0.f;
    for (int b = tid; b < B; b += 256) a += dz[b];
    a = block_sum4(a, red);
    if (first && tid == 0) p.d_fb[0] = a;
  }
  for (int i = tid; first && i < B * D; i += 256) p.d_rr[i] = dz[i / D] * p.fw[i % D];
  if (V == 0) return;
  const float sv = gl * p.rate / (float)(V * V);
  for (int i = tid; i < B * V; i += 256) {
    const int b = i / V, v = i % V;
    const float cu = p.c_u[i], ci = p.c_i[i], pm = p.pos_match[i], nm = p.neg_match[i];
    const float dfp = dz[b] * p.fw[D + v], dfn = dz[b] * p.fw[D + V + v];
    if (first) {
      p.d_cu[i] = dfp * ci * (1.f - pm) + dfn * ci * (1.f - nm);
      p.d_ci[i] = dfp * cu * (1.f - pm) + dfn * cu * (1.f - nm);
    }
    float spm = 0.f, snm = 0.f, spp = 0.f, spn = 0.f;
    for (int q = 0; q < V; ++q) {
      spm += p.pos_match[b * V + q]; snm += p.neg_match[b * V + q];
      spp += p.pp[b * V + q]; spn += p.pn[b * V + q];
    }
    if (first) { p.d_pp[i] = sv * spm; p.d_pn[i] = sv * snm; }
    const float dpm = -dfp * cu * ci + sv * spp;
    const float dnm = -dfn * cu * ci + sv * spn;
    const float ie = p.img_emb[i];
    const float dp_ = p.posneg_emb[v] - ie, dn_ = p.posneg_emb[V + v] - ie;
    const float sgp = dp_ > 0.f ? 1.f : (dp_ < 0.f ? -1.f : 0.f), sgn = dn_ > 0.f ? 1.f : (dn_ < 0.f ? -1.f : 0.f);
    const float ddp = dpm * (1.f - pm * pm) * sgp, ddn = dnm * (1.f - nm * nm) * sgn;
    dimg[i] = -(ddp + ddn);
    ddps[i] = ddp;  // d(pos_emb - img_emb)
    ddns[i] = ddn;
  }
  __syncthreads();
  for (int v = tid; v < V; v += 256) {
    float a = 0.f, c = 0.f;
    for (int b = 0; b < B; ++b) { a += ddps[b * V + v]; c += ddns[b * V + v]; }
    dpos[v] = a; dneg[v] = c;
  }
  __syncthreads();
  for (int i = blockIdx.x * 256 + tid; i < p.F; i += gridDim.x * 256) {
    const float lwi = p.lw[i];
    float a = 0.f;
    for (int bv = 0; bv < B * V; ++bv) {
      float m = 0.f;
      for (int q = 0; q < p.P; ++q) m += p.vgg[((long)bv * p.P + q) * p.F + i];
      a += dimg[bv] * (m / (float)p.P);
      const float g = dimg[bv] * lwi / (float)p.P;
      for (int q = 0; q < p.P; ++q) p.d_vgg[((long)bv * p.P + q) * p.F + i] = g;
    }
    for (int v = 0; v < V; ++v) {
      a += dpos[v] * p.pos_v[(long)v * p.F + i] + dneg[v] * p.neg_v[(long)v * p.F + i];
      p.d_pos_v[(long)v * p.F + i] = dpos[v] * lwi;
      p.d_neg_v[(long)v * p.F + i] = dneg[v] * lwi;
    }
    p.d_lw[i] = a;
  }
  {
    float a = 0.f;
    for (int i = tid; i < B * V; i += 256) a += dimg[i];
    for (int v = tid; v < V; v += 256) a += dpos[v] + dneg[v];
    a = block_sum4(a, red);
    if (first && tid == 0) p.d_lb[0] = a;
  }
}

// -------------------------------------------------------------------------------------------------- misc
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    gx[i] = gy[i] * (1.f - y[i] * y[i]);
}
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    gx[i] = y[i] > 0.f ? gy[i] : 0.f;
}
__global__ void fill_kernel(float* __restrict__ p, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
// partial[chunk][j] = sum over rows of the chunk.  256 threads = 4 row groups x 64 columns: four independent chains of 64
// loads per column instead of one chain of 256 (the kernel is bound by memory latency: the serial version took 147 us
// for a 12 MB matrix), combined in a fixed order through LDS.
__global__ void colsum_stage1_kernel(const float* __restrict__ src, long rows, int cols, long ld, int rows_per_chunk,
                                     float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const long r0 = (long)blockIdx.y * rows_per_chunk;
  const long r1 = min(rows, r0 + rows_per_chunk);
  float v0 = 0.f, v1 = 0.f;
  if (j < cols) {
    long r = r0 + grp;
    for (; r + 4 < r1; r += 8) { v0 += src[r * ld + j]; v1 += src[(r + 4) * ld + j]; }
    if (r < r1) v0 += src[r * ld + j];
  }
  red[grp][c] = v0 + v1;
  __syncthreads();
  if (grp == 0 && j < cols) part[(long)blockIdx.y * cols + j] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ void colsum_stage2_kernel(const float* __restrict__ part, int chunks, int cols, float* __restrict__ dst,
                                     int accumulate) {
  __shared__ float red[16][64];                         // blockDim.x = 64 * groups, groups <= 16
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6, ngrp = blockDim.x >> 6;
  const int j = blockIdx.x * 64 + c;
  float v = 0.f;
  if (j < cols)
    for (int q = grp; q < chunks; q += ngrp) v += part[(long)q * cols + j];
  red[grp][c] = v;
  __syncthreads();
  if (grp == 0 && j < cols) {
    float t;
    if (ngrp == 4) t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    else { t = 0.f; for (int g = 0; g < ngrp; ++g) t += red[g][c]; }
    dst[j] = accumulate ? dst[j] + t : t;
  }
}

// dropout: keep-mask from a counter hash (seed, element index); y = x * mask / (1-p)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                   long n, float p, uint64_t seed, int gen) {
  const float scale = 1.f / (1.f - p);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint8_t m;
    if (gen) {
      const uint32_t h = hash32((uint32_t)i * 0x9E3779B9U + (uint32_t)seed) ^ hash32((uint32_t)(i >> 32) + (uint32_t)(seed >> 32) + 0x85ebca6bU);
      const float u = (float)(hash32(h) >> 8) * (1.0f / 16777216.0f);
      m = u >= p ? 1 : 0;
      mask[i] = m;
    } else {
      m = mask[i];
    }
    y[i] = m ? x[i] * scale : 0.f;
  }
}
// gx = gy * mask/(1-p) * [a > 0]   (a = ReLU output feeding the dropout; a may be null)
__global__ void dropout_bwd_kernel(const float* __restrict__ gy, const uint8_t* __restrict__ mask,
                                   const float* __restrict__ a, float* __restrict__ gx, long n, float p) {
  const float scale = 1.f / (1.f - p);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float g = mask ? (mask[i] ? gy[i] * scale : 0.f) : gy[i];
    if (a && !(a[i] > 0.f)) g = 0.f;
    gx[i] = g;
  }
}

// Adam with coupled L2 exactly as torch.optim.Adam applies it (main.py:22-25):
//   g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float gscale, float wd, float b1, float b2, float eps,
                            float step_size, float inv_bc2_sqrt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] * gscale + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

inline int nblocks(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------ review merge
// src/model.py:150-151, 158: tanh(linear_u(repr_u) + linear_i(repr_i)); repr [B][256], W [128][256], out [B][128].  Batch-sized
// and latency-bound: the two register-streaming fc launches per direction each walked their whole reduction in ONE workgroup
// (15-17 us per launch at B = 32, eight launches forward + backward); these three kernels spread the same sums over 32-64
// workgroups with every load of a thread independent of the one before.
constexpr int MD = 128, MK = 256;

// grid (MD / 4, cdiv(B, 32)); thread: row m = tid >> 3, eighth ks = tid & 7 of both reductions, four output columns
__global__ __launch_bounds__(256) void merge_fwd_kernel(const float* __restrict__ ru, const float* __restrict__ ri,
                                                        const float* __restrict__ Wu, const float* __restrict__ Wi, int B,
                                                        float* __restrict__ out, float* __restrict__ out2) {
  const int tid = threadIdx.x, ks = tid & 7, m = blockIdx.y * 32 + (tid >> 3), n0 = blockIdx.x * 4;
  const int mc = m < B ? m : B - 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4* xr = reinterpret_cast<const float4*>((q ? ri : ru) + (long)mc * MK);
    const float4* wr = reinterpret_cast<const float4*>((q ? Wi : Wu) + (long)n0 * MK);
    float4 x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = xr[j * 8 + ks];      // the 8 lanes of a row read 128 contiguous bytes per j
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 w = wr[c * (MK / 4) + j * 8 + ks];
        acc[c] += x[j].x * w.x + x[j].y * w.y + x[j].z * w.z + x[j].w * w.w;
      }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    acc[c] += __shfl_xor(acc[c], 1); acc[c] += __shfl_xor(acc[c], 2); acc[c] += __shfl_xor(acc[c], 4);
  }
  if (ks == 0 && m < B) {
    float4 v = make_float4(tanhf(acc[0]), tanhf(acc[1]), tanhf(acc[2]), tanhf(acc[3]));
    *reinterpret_cast<float4*>(out + (long)m * MD + n0) = v;
    if (out2) *reinterpret_cast<float4*>(out2 + (long)m * MD + n0) = v;
  }
}

// d_repr_s[m][k] = sum_n dpre[m][n] W_s[n][k],  dpre = d_out (1 - out^2).  grid (cdiv(B, 4), 2 sides); thread: float4 column
// k4 = tid & 63, reduction quarter nq = tid >> 6 (32 n), four rows m; the quarters meet in LDS.
// NOT the default (UMPR_MERGE_DX=1 selects it): next to the bf16 weight-gradient kernels of the VGG backward (another stream) this
// kernel returned, in about one launch of three, sums that were off by a few per cent in the columns of lanes 48..63 and there in
// the .x / .z components only - with bit-identical inputs, and correct when run again on the quiescent device
// (tools/check_reproducible.py, round 3).  The ISA is clean (barriers, waitcnts, LDS extents checked); a variant that keeps the
// quarters in one wave and joins them by shuffles (no LDS exchange) showed the same, a plain one-row loop with dpre in LDS did not:
// what the failing variants share is hipcc's software-pipelined 8-deep loop (global_load_dwordx4 into registers a v_pk_fma_f32 has
// just read).  Unexplained beyond that; variant 2 below is what runs (+4 us per UMPR-R step).
__global__ __launch_bounds__(256) void merge_bwd_dx_kernel(const float* __restrict__ out, const float* __restrict__ d_out,
                                                           const float* __restrict__ Wu, const float* __restrict__ Wi, int B,
                                                           float* __restrict__ dru, float* __restrict__ dri) {
  __shared__ float dp[4][MD];
  __shared__ float4 part[3][4][64];
  const int tid = threadIdx.x, k4 = tid & 63, nq = tid >> 6, m0 = blockIdx.x * 4;
  const float* W = blockIdx.y ? Wi : Wu;
  float* dr = blockIdx.y ? dri : dru;
  for (int e = tid; e < 4 * MD; e += 256) {
    const int m = m0 + e / MD, n = e % MD;
    float v = 0.f;
    if (m < B) { const float y = out[(long)m * MD + n]; v = d_out[(long)m * MD + n] * (1.f - y * y); }
    dp[e / MD][n] = v;
  }
  __syncthreads();
  float4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* wp = reinterpret_cast<const float4*>(W + (long)(nq * 32) * MK) + k4;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    const float4 w = wp[(long)i * (MK / 4)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = dp[r][nq * 32 + i];
      acc[r].x += d * w.x; acc[r].y += d * w.y; acc[r].z += d * w.z; acc[r].w += d * w.w;
    }
  }
  if (nq > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[nq - 1][r][k4] = acc[r];
  }
  __syncthreads();
  if (nq == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float4 v = acc[r];
#pragma unroll
      for (int q = 0; q < 3; ++q) { const float4 o = part[q][r][k4]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      if (m0 + r < B) reinterpret_cast<float4*>(dr + (long)(m0 + r) * MK)[k4] = v;
    }
  }
}

// variant 2 (the default): the same sums, one thread per (row, float4 column), a plain loop, no LDS
__global__ __launch_bounds__(256) void merge_bwd_dx_v2_kernel(const float* __restrict__ out, const float* __restrict__ d_out,
                                                              const float* __restrict__ Wu, const float* __restrict__ Wi, int B,
                                                              float* __restrict__ dru, float* __restrict__ dri) {
  const int tid = threadIdx.x, k4 = tid & 63, m = blockIdx.x * 4 + (tid >> 6);
  if (m >= B) return;
  const float4* W = reinterpret_cast<const float4*>(blockIdx.y ? Wi : Wu) + k4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < MD; ++n) {
    const float y = out[(long)m * MD + n];
    const float d = d_out[(long)m * MD + n] * (1.f - y * y);
    const float4 w = W[(long)n * (MK / 4)];
    acc.x += d * w.x; acc.y += d * w.y; acc.z += d * w.z; acc.w += d * w.w;
  }
  reinterpret_cast<float4*>((blockIdx.y ? dri : dru) + (long)m * MK)[k4] = acc;
}

// dW_s[n][k] = sum_m dpre[m][n] repr_s[m][k].  grid (MD / 4, 2 sides); thread: float4 column k4 = tid & 63 of row n0 + (tid >> 6)
__global__ __launch_bounds__(256) void merge_bwd_dw_kernel(const float* __restrict__ out, const float* __restrict__ d_out,
                                                           const float* __restrict__ ru, const float* __restrict__ ri, int B,
                                                           float* __restrict__ dWu, float* __restrict__ dWi) {
  __shared__ float dp[128][4];
  const int tid = threadIdx.x, k4 = tid & 63, nn = tid >> 6, n0 = blockIdx.x * 4;
  const float* R = blockIdx.y ? ri : ru;
  float* dW = blockIdx.y ? dWi : dWu;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int mb = 0; mb < B; mb += 128) {          // B <= 128 in practice: one pass
    const int mcnt = min(128, B - mb);
    __syncthreads();
    for (int e = tid; e < mcnt * 4; e += 256) {
      const int m = mb + (e >> 2), n = n0 + (e & 3);
      const float y = out[(long)m * MD + n];
      dp[e >> 2][e & 3] = d_out[(long)m * MD + n] * (1.f - y * y);
    }
    __syncthreads();
    const float4* rp = reinterpret_cast<const float4*>(R + (long)mb * MK) + k4;
#pragma unroll 8
    for (int m = 0; m < mcnt; ++m) {
      const float4 x = rp[(long)m * (MK / 4)];
      const float d = dp[m][nn];
      acc.x += d * x.x; acc.y += d * x.y; acc.z += d * x.z; acc.w += d * x.w;
    }
  }
  reinterpret_cast<float4*>(dW + (long)(n0 + nn) * MK)[k4] = acc;
}

}  // namespace

// ---- internal host wrappers --------------------------------------------------------------------------------
// up to 8 copies / accumulations, or 8 column sums, in ONE launch (blockIdx.y = segment): the small bookkeeping of a
// GRU call (stacking the two directions' input weights; reducing the per-tile slabs of six gradients) was 4 + 8 launches
struct MultiCopy { const float* src[8]; float* dst[8]; long n[8]; int accumulate[8]; };
__global__ void multi_copy_kernel(MultiCopy mc) {
  const int q = blockIdx.y;
  const float* __restrict__ src = mc.src[q];
  float* __restrict__ dst = mc.dst[q];
  const long n = mc.n[q];
  const int acc = mc.accumulate[q];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = acc ? dst[i] + src[i] : src[i];
}
struct MultiColsum { const float* src[8]; float* dst[8]; int rows[8]; long cols[8]; long row_stride[8]; int accumulate[8]; };
__global__ void multi_colsum_rows_kernel(MultiColsum mc) {
  const int q = blockIdx.y;
  const float* __restrict__ src = mc.src[q];
  float* __restrict__ dst = mc.dst[q];
  const int rows = mc.rows[q];
  const long cols = mc.cols[q], rs = mc.row_stride[q];
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < cols; j += (long)gridDim.x * blockDim.x) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;   // four chains in flight, combined in a fixed order (as colsum_rows)
    int i = 0;
    for (; i + 3 < rows; i += 4) {
      v0 += src[(long)i * rs + j]; v1 += src[(long)(i + 1) * rs + j];
      v2 += src[(long)(i + 2) * rs + j]; v3 += src[(long)(i + 3) * rs + j];
    }
    for (; i < rows; ++i) v0 += src[(long)i * rs + j];
    const float v = (v0 + v1) + (v2 + v3);
    dst[j] = mc.accumulate[q] ? dst[j] + v : v;
  }
}

__global__ void copy_or_add_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = accumulate ? dst[i] + src[i] : src[i];
}
int umpr_multi_copy(const float* const* src, float* const* dst, const long* n, const int* accumulate, int nseg, hipStream_t s) {
  UMPR_REQUIRE(nseg >= 1 && nseg <= 8, "multi_copy: %d segments", nseg);
  MultiCopy mc;
  long mx = 0;
  for (int q = 0; q < nseg; ++q) { mc.src[q] = src[q]; mc.dst[q] = dst[q]; mc.n[q] = n[q]; mc.accumulate[q] = accumulate ? accumulate[q] : 0; if (n[q] > mx) mx = n[q]; }
  multi_copy_kernel<<<dim3(nblocks(mx), nseg), 256, 0, s>>>(mc);
  UMPR_LAUNCH_CHECK("multi_copy");
  return 0;
}
int umpr_multi_colsum_rows(const float* const* src, const int* rows, const long* cols, const long* row_stride, float* const* dst,
                           const int* accumulate, int nseg, hipStream_t s) {
  UMPR_REQUIRE(nseg >= 1 && nseg <= 8, "multi_colsum_rows: %d segments", nseg);
  MultiColsum mc;
  long mx = 0;
  for (int q = 0; q < nseg; ++q) {
    mc.src[q] = src[q]; mc.dst[q] = dst[q]; mc.rows[q] = rows[q]; mc.cols[q] = cols[q]; mc.row_stride[q] = row_stride[q];
    mc.accumulate[q] = accumulate ? accumulate[q] : 0;
    if (cols[q] > mx) mx = cols[q];
  }
  int blocks = (int)((mx + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  multi_colsum_rows_kernel<<<dim3(blocks, nseg), 256, 0, s>>>(mc);
  UMPR_LAUNCH_CHECK("multi_colsum_rows");
  return 0;
}
int umpr_copy_or_add(const float* src, float* dst, long n, int accumulate, hipStream_t s) {
  copy_or_add_kernel<<<nblocks(n), 256, 0, s>>>(src, dst, n, accumulate);
  UMPR_LAUNCH_CHECK("copy_or_add");
  return 0;
}
int umpr_fill(float* p, long n, float v, hipStream_t s) {
  fill_kernel<<<nblocks(n), 256, 0, s>>>(p, n, v);
  UMPR_LAUNCH_CHECK("fill");
  return 0;
}
int umpr_review_merge_fwd_impl(const float* ru, const float* ri, const float* Wu, const float* Wi, int B, float* out, float* out2,
                               hipStream_t s) {
  merge_fwd_kernel<<<dim3(MD / 4, cdiv(B, 32)), 256, 0, s>>>(ru, ri, Wu, Wi, B, out, out2);
  UMPR_LAUNCH_CHECK("merge_fwd");
  return 0;
}
int umpr_review_merge_bwd_impl(const float* ru, const float* ri, const float* Wu, const float* Wi, const float* out,
                               const float* d_out, int B, float* dru, float* dri, float* dWu, float* dWi, hipStream_t s) {
  static const int dx = umpr_env_int("UMPR_MERGE_DX", 2);     // 1: the pipelined kernel above (see its comment)
  if (dx == 2) merge_bwd_dx_v2_kernel<<<dim3(cdiv(B, 4), 2), 256, 0, s>>>(out, d_out, Wu, Wi, B, dru, dri);
  else merge_bwd_dx_kernel<<<dim3(cdiv(B, 4), 2), 256, 0, s>>>(out, d_out, Wu, Wi, B, dru, dri);
  UMPR_LAUNCH_CHECK("merge_bwd_dx");
  merge_bwd_dw_kernel<<<dim3(MD / 4, 2), 256, 0, s>>>(out, d_out, ru, ri, B, dWu, dWi);
  UMPR_LAUNCH_CHECK("merge_bwd_dw");
  return 0;
}
int umpr_tanh_bwd(const float* y, const float* gy, float* gx, long n, hipStream_t s) {
  tanh_bwd_kernel<<<nblocks(n), 256, 0, s>>>(y, gy, gx, n);
  UMPR_LAUNCH_CHECK("tanh_bwd");
  return 0;
}
int umpr_relu_bwd(const float* y, const float* gy, float* gx, long n, hipStream_t s) {
  relu_bwd_kernel<<<nblocks(n), 256, 0, s>>>(y, gy, gx, n);
  UMPR_LAUNCH_CHECK("relu_bwd");
  return 0;
}
size_t umpr_colsum_ws_bytes(long rows, int cols) { return (size_t)cdiv(rows, 256) * cols * sizeof(float); }
int umpr_colsum(const float* src, long rows, int cols, long ld, float* dst, int accumulate, float* ws, size_t ws_bytes,
                hipStream_t s) {
  const int chunks = cdiv(rows, 256);
  UMPR_REQUIRE(ws_bytes >= (size_t)chunks * cols * sizeof(float), "colsum: workspace too small");
  colsum_stage1_kernel<<<dim3(cdiv(cols, 64), chunks), 256, 0, s>>>(src, rows, cols, ld, 256, ws);
  UMPR_LAUNCH_CHECK("colsum1");
  colsum_stage2_kernel<<<cdiv(cols, 64), 256, 0, s>>>(ws, chunks, cols, dst, accumulate);
  UMPR_LAUNCH_CHECK("colsum2");
  return 0;
}
int umpr_dropout_fwd_impl(const float* x, float* y, uint8_t* mask, long n, float p, uint64_t seed, int gen, hipStream_t s) {
  dropout_fwd_kernel<<<nblocks(n), 256, 0, s>>>(x, y, mask, n, p, seed, gen);
  UMPR_LAUNCH_CHECK("dropout_fwd");
  return 0;
}
int umpr_dropout_bwd_impl(const float* gy, const uint8_t* mask, const float* a, float* gx, long n, float p, hipStream_t s) {
  dropout_bwd_kernel<<<nblocks(n), 256, 0, s>>>(gy, mask, a, gx, n, p);
  UMPR_LAUNCH_CHECK("dropout_bwd");
  return 0;
}
// evaluate_mse (src/evaluate.py:12-13): acc[0] += sum (pred - label)^2, acc[1] += n.  One workgroup, fixed summation
// order (thread strides, then a shuffle tree per wave, then the waves in order), float64 accumulator across batches.
__global__ void __launch_bounds__(256) sq_err_accumulate_kernel(const float* __restrict__ pred, const float* __restrict__ label,
                                                                long n, double* __restrict__ acc) {
  __shared__ double part[4];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float d = pred[i] - label[i];   // fp32 difference and square like torch's mse_loss(reduction='sum') elements
    s += (double)(d * d);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    acc[0] += ((part[0] + part[1]) + part[2]) + part[3];
    acc[1] += (double)n;
  }
}
int umpr_sq_err_accumulate_impl(const float* pred, const float* label, long n, double* acc, hipStream_t s) {
  sq_err_accumulate_kernel<<<1, 256, 0, s>>>(pred, label, n, acc);
  UMPR_LAUNCH_CHECK("sq_err_accumulate");
  return 0;
}
// The same update with its per-step scalars read from DEVICE memory (hyper[0..3] = grad_scale, step_size = lr / (1 - b1^t),
// 1 / sqrt(1 - b2^t), weight_decay as the caller will use them for the NEXT launch): what a captured hipGraph of a training step
// launches - the node's kernel arguments are frozen at capture, the bias corrections change every step.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long n, float b1, float b2, float eps, const float* __restrict__ hyper) {
  const float gscale = hyper[0], step_size = hyper[1], inv_bc2_sqrt = hyper[2], wd = hyper[3];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] * gscale + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}
int umpr_adam_dev_impl(float* p, const float* g, float* m, float* v, long n, float b1, float b2, float eps, const float* hyper,
                       hipStream_t s) {
  adam_dev_kernel<<<nblocks(n, 8192), 256, 0, s>>>(p, g, m, v, n, b1, b2, eps, hyper);
  UMPR_LAUNCH_CHECK("adam_dev");
  return 0;
}
int umpr_adam_impl(float* p, const float* g, float* m, float* v, long n, float gscale, float wd, float b1, float b2,
                   float eps, float step_size, float inv_bc2_sqrt, hipStream_t s) {
  adam_kernel<<<nblocks(n, 8192), 256, 0, s>>>(p, g, m, v, n, gscale, wd, b1, b2, eps, step_size, inv_bc2_sqrt);
  UMPR_LAUNCH_CHECK("adam");
  return 0;
}

// ---- S-Net ---------------------------------------------------------------------------------------------------
int umpr_snet_fwd_impl(const float* X, const float* Ms, const float* Ws, const float* word_soft, int wl, int B, int S,
                       int L, float* U, float* P, float* wsum, float* self_atte, float* senti, long ld_senti,
                       hipStream_t s) {
  UMPR_REQUIRE(L >= 1 && L <= 256, "snet: sentence length %d outside 1..256", L);
  UmprGemm g;
  g.A = X; g.lda = D; g.B = Ms; g.ldb = D; g.transB = true; g.C = U; g.ldc = AT; g.M = B * S * L; g.N = AT; g.K = D;
  g.act = UMPR_ACT_TANH;
  if (int rc = umpr_gemm(g, s)) return rc;
  SnetFwdParams p{X, U, Ws, word_soft, wl, P, wsum, self_atte, senti, ld_senti, S, L};
  const long nsent = (long)B * S;
  const unsigned fgrid = (unsigned)((nsent + 3) / 4);
  if (L <= 64) snet_pool_fwd_kernel<1><<<fgrid, 256, 0, s>>>(p, nsent);
  else if (L <= 128) snet_pool_fwd_kernel<2><<<fgrid, 256, 0, s>>>(p, nsent);
  else snet_pool_fwd_kernel<4><<<fgrid, 256, 0, s>>>(p, nsent);
  UMPR_LAUNCH_CHECK("snet_pool_fwd");
  snet_senti_kernel<<<B, D, 0, s>>>(wsum, self_atte, senti, ld_senti, S);
  UMPR_LAUNCH_CHECK("snet_senti");
  return 0;
}

size_t umpr_snet_bwd_ws_bytes_impl(int B, int S, int L) {
  return ((size_t)B * S * L * AT + ((size_t)B * S + 3) / 4 * AT + (size_t)512 * AT * D) * sizeof(float);
}

int umpr_snet_bwd_impl(const float* X, const float* Ms, const float* Ws, const float* U, const float* P,
                       const float* wsum, const float* self_atte, const float* d_senti, long ld_ds,
                       const float* d_self_atte, int B, int S, int L, int wl, float* dX, float* dMs, float* dWs,
                       float* d_word_soft, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(ws_bytes >= umpr_snet_bwd_ws_bytes_impl(B, S, L), "snet_bwd: workspace too small");
  const long nsent = (long)B * S;
  const int nwg = (int)((nsent + 3) / 4);
  float* dPre = ws; float* dWs_part = ws + (size_t)B * S * L * AT; float* slab = dWs_part + (size_t)nwg * AT;
  SnetBwdParams p{X, U, Ws, P, wsum, self_atte, d_senti, ld_ds, d_self_atte, dX, dPre, dWs_part, d_word_soft, wl, S, L};
  if (L <= 64) snet_pool_bwd_kernel<1><<<nwg, 256, 0, s>>>(p, nsent);
  else if (L <= 128) snet_pool_bwd_kernel<2><<<nwg, 256, 0, s>>>(p, nsent);
  else snet_pool_bwd_kernel<4><<<nwg, 256, 0, s>>>(p, nsent);
  UMPR_LAUNCH_CHECK("snet_pool_bwd");
  const int R = B * S * L;
  UmprGemm g;  // dX += dPre Ms
  g.A = dPre; g.lda = AT; g.B = Ms; g.ldb = D; g.C = dX; g.ldc = D; g.M = R; g.N = D; g.K = AT; g.accumulate = true;
  if (int rc = umpr_gemm(g, s)) return rc;
  UmprGemm h;  // dMs = dPre^T X
  h.A = dPre; h.lda = AT; h.transA = true; h.B = X; h.ldb = D; h.C = dMs; h.ldc = D; h.M = AT; h.N = D; h.K = R;
  h.split_k = 0; h.ws = slab; h.ws_bytes = (size_t)512 * AT * D * sizeof(float);
  if (int rc = umpr_gemm(h, s)) return rc;
  colsum_stage2_kernel<<<1, 1024, 0, s>>>(dWs_part, nwg, AT, dWs, 0);
  UMPR_LAUNCH_CHECK("snet_dWs");
  return 0;
}

// ---- C-Net head ------------------------------------------------------------------------------------------------
// forward scratch: the window-ordered weight [KC][KS*D] (KC <= 512)
size_t umpr_cnet_fwd_ws_bytes(int B, int S, int L, int KS) { (void)B; (void)S; (void)L; return (size_t)512 * D * KS * sizeof(float); }

int umpr_cnet_head_fwd_impl(const float* X, const float* Wc, const float* bc, const float* Wl, const float* bl,
                            float thr, int B, int S, int L, int KC, int KS, int V, float* Y, float* cmax, int* argl,
                            float* sp, float* view_p, float* final_, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(ws_bytes >= umpr_cnet_fwd_ws_bytes(B, S, L, KS), "cnet: workspace too small");
  const long R = (long)B * S * L;
  // nn.Conv1d(padding=(KS-1)/2) (src/model.py:93): an even KS yields L - 1 positions; the window GEMM computes L (the last one
  // reads one zero past the sentence) and the head takes its maximum over the valid ones
  const int pad = (KS - 1) / 2, Lout = L + 2 * pad - KS + 1;
  UMPR_REQUIRE(KS >= 1 && Lout >= 1 && KC <= 512, "cnet: kernel size %d on sentences of %d tokens, or more than 512 filters (%d)", KS, L, KC);
  float* Wp = ws;                                        // [KC][KS*D] in window order
  cnet_weight_order_kernel<<<nblocks((long)KC * D * KS), 256, 0, s>>>(Wc, Wp, KC, D, KS, 0, 0);
  UMPR_LAUNCH_CHECK("cnet_weight_order");
  UmprGemm g;   // Y = relu(win(X) Wp^T + bc): X read in place through the sliding window
  g.A = X; g.lda = D; g.winA_L = L; g.winA_D = D; g.winA_pad = pad;
  g.B = Wp; g.ldb = D * KS; g.transB = true; g.C = Y; g.ldc = KC; g.M = (int)R; g.N = KC;
  g.K = D * KS; g.bias = bc; g.bias_mode = 1; g.act = UMPR_ACT_RELU;
  if (int rc = umpr_gemm(g, s)) return rc;
  CnetHeadFwdParams p{Y, Wl, bl, thr, cmax, argl, sp, view_p, final_, S, L, KC, V, Lout};
  const long nsent = (long)B * S;
  cnet_head_fwd_kernel<<<(unsigned)((nsent + 3) / 4), 256, 4 * KC * sizeof(float), s>>>(p, nsent);
  UMPR_LAUNCH_CHECK("cnet_head_fwd");
  cnet_final_kernel<<<B, 64, 0, s>>>(view_p, final_, S, V);
  UMPR_LAUNCH_CHECK("cnet_final");
  return 0;
}

size_t umpr_cnet_bwd_ws_bytes(int B, int S, int L, int KC, int KS, int V) {
  const size_t R = (size_t)B * S * L;
  // dY + dWp + Wq (window-ordered weight gradient / transposed weight) + head partials + split-K slab
  return (R * KC + 2 * (size_t)KC * D * KS + (size_t)B * V * KC + (size_t)B * V + (size_t)cdiv(R, 256) * KC +
          (size_t)512 * KC * D * KS) * sizeof(float);
}

int umpr_cnet_head_bwd_impl(const float* X, const float* Wc, const float* Wl, const float* cmax, const int* argl,
                            const float* sp, const float* view_p, const float* d_final, const float* d_view_p, int B,
                            int S, int L, int KC, int KS, int V, float* dX, int accumulate_dX, int accumulate_w, float* dWc,
                            float* dbc, float* dWl, float* dbl, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(ws_bytes >= umpr_cnet_bwd_ws_bytes(B, S, L, KC, KS, V), "cnet_bwd: workspace too small");
  const long R = (long)B * S * L;
  const int CK = D * KS;
  float* dY = ws;
  float* dWp = dY + R * KC;                              // [KC][CK] in window order
  float* Wq = dWp + (size_t)KC * CK;                     // [KS*KC][D]
  float* dWl_part = Wq + (size_t)KC * CK;
  float* dbl_part = dWl_part + (size_t)B * V * KC;
  float* cs = dbl_part + (size_t)B * V;
  float* slab = cs + (size_t)cdiv(R, 256) * KC;
  if (hipMemsetAsync(dY, 0, (size_t)R * KC * sizeof(float), s) != hipSuccess) { umpr_set_error("cnet_bwd: memset"); return -2; }
  CnetHeadBwdParams p{cmax, argl, sp, view_p, Wl, d_final, d_view_p, dY, dWl_part, dbl_part, S, L, KC, V};
  cnet_head_bwd_kernel<<<B, 256, (4 * V * KC + 8 * V) * sizeof(float), s>>>(p);
  UMPR_LAUNCH_CHECK("cnet_head_bwd");
  colsum_stage2_kernel<<<cdiv(V * KC, 64), 256, 0, s>>>(dWl_part, B, V * KC, dWl, accumulate_w);
  colsum_stage2_kernel<<<cdiv(V, 64), 256, 0, s>>>(dbl_part, B, V, dbl, accumulate_w);
  UMPR_LAUNCH_CHECK("cnet_dWl");
  if (int rc = umpr_colsum(dY, R, KC, KC, dbc, accumulate_w, cs, (size_t)cdiv(R, 256) * KC * sizeof(float), s)) return rc;
  UMPR_REQUIRE(KS >= 1 && (KC & 3) == 0, "cnet_bwd: kernel size %d / %d filters", KS, KC);
  const int pad = (KS - 1) / 2;
  UmprGemm h;  // dWp[KC][CK] = dY^T win(X)
  h.A = dY; h.lda = KC; h.transA = true; h.B = X; h.ldb = D; h.winB_L = L; h.winB_D = D; h.winB_pad = pad;
  h.C = dWp; h.ldc = CK; h.M = KC; h.N = CK; h.K = (int)R;
  h.split_k = 0; h.ws = slab; h.ws_bytes = (size_t)512 * KC * CK * sizeof(float);
  if (int rc = umpr_gemm(h, s)) return rc;
  cnet_weight_order_kernel<<<nblocks((long)KC * CK), 256, 0, s>>>(dWp, dWc, KC, D, KS, 2, accumulate_w);
  cnet_weight_order_kernel<<<nblocks((long)KC * CK), 256, 0, s>>>(Wc, Wq, KC, D, KS, 1, 0);
  UMPR_LAUNCH_CHECK("cnet_weight_order");
  UmprGemm g;  // dX (+)= win(dY) Wq: the transposed convolution, again through the window
  // y[j] = sum_t w_t x[j + t - pad]  =>  dx[i] = sum_j' dy[i + j' - (KS - 1 - pad)] w_{KS-1-j'}: the window's pad is KS - 1 - pad
  // (= pad for odd KS)
  g.A = dY; g.lda = KC; g.winA_L = L; g.winA_D = KC; g.winA_pad = KS - 1 - pad;
  g.B = Wq; g.ldb = D; g.C = dX; g.ldc = D; g.M = (int)R; g.N = D; g.K = KS * KC; g.accumulate = accumulate_dX != 0;
  if (int rc = umpr_gemm(g, s)) return rc;
  return 0;
}

// ---- gate ------------------------------------------------------------------------------------------------------
int umpr_gate_fwd_impl(const float* sa, const float* w, const float* bias, const float* view_p, const float* c_out,
                       int B, int S, int V, float* senti, float* vs, float* pp, float* pn, hipStream_t s) {
  GateFwdParams p{sa, w, bias, view_p, c_out, senti, vs, pp, pn, S, V};
  gate_fwd_kernel<<<B, 64, S * sizeof(float), s>>>(p);
  UMPR_LAUNCH_CHECK("gate_fwd");
  return 0;
}
size_t umpr_gate_bwd_ws_bytes(int B) { return (size_t)B * (D + 1) * sizeof(float); }
int umpr_gate_bwd_impl(const float* sa, const float* w, const float* view_p, const float* c_out, const float* senti,
                       const float* vs, const float* d_pp, const float* d_pn, int B, int S, int V, float* d_sa,
                       float* d_view_p, float* d_c_out, float* dw, float* db, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(ws_bytes >= umpr_gate_bwd_ws_bytes(B), "gate_bwd: workspace too small");
  float* dw_part = ws; float* db_part = ws + (size_t)B * D;
  GateBwdParams p{sa, w, view_p, c_out, senti, vs, d_pp, d_pn, d_sa, d_view_p, d_c_out, dw_part, db_part, S, V};
  gate_bwd_kernel<<<B, 64, (2 * V + S) * sizeof(float), s>>>(p);
  UMPR_LAUNCH_CHECK("gate_bwd");
  colsum_stage2_kernel<<<2, 256, 0, s>>>(dw_part, B, D, dw, 0);
  colsum_stage2_kernel<<<1, 256, 0, s>>>(db_part, B, 1, db, 0);
  UMPR_LAUNCH_CHECK("gate_dw");
  return 0;
}

// ---- head ------------------------------------------------------------------------------------------------------
int umpr_head_launch(const UmprHead& p, int backward, hipStream_t s) {
  if (!backward) {
    if (p.V > 0) {
      head_emb_kernel<<<(2 * p.V + p.B * p.V + 3) / 4, 256, 0, s>>>(p);
      UMPR_LAUNCH_CHECK("head_emb");
    }
    head_fwd_kernel<<<1, 256, 0, s>>>(p);
    UMPR_LAUNCH_CHECK("head_fwd");
  } else {
    const size_t sh = ((size_t)p.B + 3 * (size_t)p.B * p.V + 2 * p.V) * sizeof(float);
    head_bwd_kernel<<<p.V > 0 ? (p.F + 255) / 256 : 1, 256, sh, s>>>(p);
    UMPR_LAUNCH_CHECK("head_bwd");
  }
  return 0;
}

// ---- R-Net pre-training head (pretrain/pretrain_rnet.py:147-169): sigmoid(Linear(256 -> 1)) + BCELoss(mean) ------
namespace {

// one wave per sample: z = att . w + b, p = sigmoid(z), term = -(t log p + (1 - t) log(1 - p)) with torch's -100 clamp
__global__ void bce_head_fwd_kernel(const float* __restrict__ att, long ld, const float* __restrict__ w,
                                    const float* __restrict__ b, const float* __restrict__ target, int B, int K,
                                    float* __restrict__ result, float* __restrict__ terms) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += att[(long)row * ld + k] * w[k];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    const float z = acc + b[0];
    const float p = 1.f / (1.f + expf(-z));
    const float t = target[row];
    result[row] = p;
    terms[row] = -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(log1pf(-p), -100.f));
  }
}

// single workgroup, fixed summation order: loss = mean(terms)
__global__ void bce_mean_kernel(const float* __restrict__ terms, int B, float* __restrict__ loss) {
  __shared__ float part[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) a += terms[i];
  part[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0] / (float)B;
}

// dz[b] = d_loss/B * (p - t)/max(p(1-p), 1e-12) * p(1-p)  (+ d_result[b] * p(1-p));  d_att[b][k] = dz[b] w[k]
__global__ void bce_head_bwd_rows_kernel(const float* __restrict__ w, const float* __restrict__ result,
                                         const float* __restrict__ target, const float* __restrict__ d_result,
                                         const float* __restrict__ d_loss, int B, int K, float* __restrict__ dz,
                                         float* __restrict__ d_att, long ld_d) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const float p = result[row], t = target[row];
  const float pq = p * (1.f - p);
  float g = d_loss[0] / (float)B * (p - t) / fmaxf(pq, 1e-12f);
  if (d_result) g += d_result[row];
  const float z = g * pq;
  if (lane == 0) dz[row] = z;
  for (int k = lane; k < K; k += 64) d_att[(long)row * ld_d + k] = z * w[k];
}

// dw[k] = sum_b dz[b] att[b][k] (thread per k, b ascending), db = sum_b dz[b] (thread K)
__global__ void bce_head_bwd_w_kernel(const float* __restrict__ att, long ld, const float* __restrict__ dz, int B,
                                      int K, float* __restrict__ dw, float* __restrict__ db) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > K) return;
  float a = 0.f;
  if (k < K) {
    for (int b = 0; b < B; ++b) a += dz[b] * att[(long)b * ld + k];
    dw[k] = a;
  } else {
    for (int b = 0; b < B; ++b) a += dz[b];
    db[0] = a;
  }
}

}  // namespace

int umpr_bce_head_fwd_impl(const float* att, long ld, const float* w, const float* b, const float* target, int B, int K,
                           float* result, float* loss, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(B > 0 && K > 0 && ld >= K, "bce_head_fwd: bad shape B=%d K=%d ld=%ld", B, K, ld);
  UMPR_REQUIRE(ws_bytes >= (size_t)B * sizeof(float), "bce_head_fwd: workspace too small");
  bce_head_fwd_kernel<<<(B + 3) / 4, 256, 0, s>>>(att, ld, w, b, target, B, K, result, ws);
  UMPR_LAUNCH_CHECK("bce_head_fwd");
  bce_mean_kernel<<<1, 256, 0, s>>>(ws, B, loss);
  UMPR_LAUNCH_CHECK("bce_mean");
  return 0;
}

int umpr_bce_head_bwd_impl(const float* att, long ld, const float* w, const float* result, const float* target,
                           const float* d_result, const float* d_loss, int B, int K, float* d_att, long ld_d, float* dw,
                           float* db, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(B > 0 && K > 0 && ld >= K && ld_d >= K, "bce_head_bwd: bad shape");
  UMPR_REQUIRE(ws_bytes >= (size_t)B * sizeof(float), "bce_head_bwd: workspace too small");
  bce_head_bwd_rows_kernel<<<(B + 3) / 4, 256, 0, s>>>(w, result, target, d_result, d_loss, B, K, ws, d_att, ld_d);
  UMPR_LAUNCH_CHECK("bce_head_bwd_rows");
  bce_head_bwd_w_kernel<<<(K + 1 + 63) / 64, 64, 0, s>>>(att, ld, ws, B, K, dw, db);
  UMPR_LAUNCH_CHECK("bce_head_bwd_w");
  return 0;
}
