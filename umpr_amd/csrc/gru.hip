// Bidirectional variable-length GRU (hidden 64) for gfx950: recurrent forward and BPTT.
//
// Reference semantics: ImprovedRnn.forward, src/model.py:12-21 = pack_padded_sequence -> nn.GRU -> pad_packed_sequence
// (zeros past each length) -> an extra gather by unsorted_indices.  Here the host passes the permutation
// (dst_row[n] = sorted_indices[n]: input row n is written to output row dst_row[n]) and lengths as int32 device
// arrays; `order` groups sequences of similar length into one workgroup (64 sequences x one direction).
//
// Per step the workgroup computes gh[seq][gate] = H[seq][64] * W_hh^T[64][192] on v_mfma_f32_32x32x2_f32 with the
// hidden state and W_hh staged in LDS; D lanes run along the hidden unit, so the gx loads, the out/saved stores and
// the dgx stores are 128-B coalesced segments.  Gate order (r,z,n); n = tanh(gx_n + r*(W_hn h + b_hn)).
#include "umpr_common.h"
#include "umpr_internal.h"

namespace {

constexpr int H = 64;      // hidden size (config.gru_size)
constexpr int G3 = 192;    // 3*H
constexpr int TS = 64;     // sequences per workgroup
constexpr int LDH = 65;    // Hs[k][LDH]

struct GruFwdParams {
  const float* gx;       // [N][L][384] = W_ih x + b_ih for (fwd | reverse)
  const float* whh[2];   // [192][64]
  const float* bhh[2];   // [192]
  const int* lengths;    // [N]
  const int* order;      // [N]
  const int* dst_row;    // [N]
  float* out;            // [N][L][128], pre-zeroed
  float* saved;          // [2][N][L][4][64] (r, z, n, W_hn h + b_hn) or null
  int N, L;
};

__global__ __launch_bounds__(256) void gru_fwd_kernel(GruFwdParams p) {
  __shared__ float Ws[H * G3];        // Ws[k][gate col] = W_hh[gate col][k]
  __shared__ float Hs[H * LDH];       // Hs[k][seq]
  __shared__ int s_n[TS], s_len[TS], s_dst[TS];
  __shared__ int s_maxlen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave >> 1, wh = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int dir = blockIdx.y;
  const int tile = blockIdx.x;

  if (tid == 0) s_maxlen = 0;
  __syncthreads();
  if (tid < TS) {
    const int pos = tile * TS + tid;
    int n = -1, len = 0, dst = 0;
    if (pos < p.N) {
      n = p.order[pos];
      len = p.lengths[n];
      if (len > p.L) len = p.L;
      dst = p.dst_row[n];
    }
    s_n[tid] = n; s_len[tid] = len; s_dst[tid] = dst;
    atomicMax(&s_maxlen, len);
  }
  const float* whh = p.whh[dir];
  for (int e = tid; e < G3 * H; e += 256) {
    const int j = e / H, k = e % H;  // global is [j][k], k contiguous
    Ws[k * G3 + j] = whh[e];
  }
  for (int e = tid; e < H * LDH; e += 256) Hs[e] = 0.f;
  __syncthreads();
  const int maxlen = s_maxlen;
  const int hid = wh * 32 + l31;
  float bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) bh[g] = p.bhh[dir][g * H + hid];
  float hreg[16];
  int nreg[16], lreg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    hreg[r] = 0.f;
    const int seq = ws * 32 + mfma_row(r, lane);
    nreg[r] = s_n[seq];
    lreg[r] = s_len[seq];
  }

  for (int step = 0; step < maxlen; ++step) {
    const int t = dir == 0 ? step : maxlen - 1 - step;
    float gxr[3][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // unconditional (clamped) loads: a per-lane `act ? load : 0` makes hipcc branch and wait per load
      const bool act = t < lreg[r];
      const float* g = p.gx + (act ? ((long)nreg[r] * p.L + t) * 384 + dir * G3 + hid : 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) gxr[q][r] = g[q * H];
    }
    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = bh[q];
#pragma unroll 8
    for (int kk = 0; kk < H / 2; ++kk) {
      const int k = 2 * kk + kh;
      const float a = Hs[k * LDH + ws * 32 + l31];
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = mfma32(a, Ws[k * G3 + q * H + hid], acc[q]);
    }
    __syncthreads();  // every wave has finished reading Hs for this step
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool act = t < lreg[r];
      if (act) {
        const float rr = sigmoidf_(gxr[0][r] + acc[0][r]);
        const float zz = sigmoidf_(gxr[1][r] + acc[1][r]);
        const float hn = acc[2][r];
        const float nn = tanhf(gxr[2][r] + rr * hn);
        const float hnew = (1.f - zz) * nn + zz * hreg[r];
        hreg[r] = hnew;
        const int seq = ws * 32 + mfma_row(r, lane);
        p.out[((long)s_dst[seq] * p.L + t) * 128 + dir * H + hid] = hnew;
        if (p.saved) {
          float* sv = p.saved + ((((long)dir * p.N + nreg[r]) * p.L + t) * 4) * H + hid;
          sv[0] = rr; sv[H] = zz; sv[2 * H] = nn; sv[3 * H] = hn;
        }
        Hs[hid * LDH + seq] = hnew;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
struct GruBwdParams {
  const float* dout;     // [N][L][128] gradient of the (permuted) output
  const float* out;      // [N][L][128] forward output (source of h_{t-1})
  const float* saved;    // [2][N][L][4][64]
  const float* whh[2];
  const int* lengths;
  const int* order;
  const int* dst_row;
  float* dgx;            // [N][L][384], pre-zeroed
  float* dwhh_slab;      // [tiles][2][192][64]
  float* dbias_slab;     // [tiles][2][2][192]  (db_ih, db_hh)
  int N, L;
};

constexpr int LDG = 65;  // DG[gate][LDG]

__global__ __launch_bounds__(256) void gru_bwd_kernel(GruBwdParams p) {
  __shared__ float Ws[G3 * H];     // Ws[gate][hid] = W_hh (natural layout)
  __shared__ float DG[G3 * LDG];   // DG[gate][seq]
  __shared__ float HP[TS * H];     // HP[seq][hid] = h_{prev}
  __shared__ int s_n[TS], s_len[TS], s_dst[TS];
  __shared__ int s_maxlen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave >> 1, wh = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int dir = blockIdx.y, tile = blockIdx.x;

  if (tid == 0) s_maxlen = 0;
  __syncthreads();
  if (tid < TS) {
    const int pos = tile * TS + tid;
    int n = -1, len = 0, dst = 0;
    if (pos < p.N) {
      n = p.order[pos];
      len = p.lengths[n];
      if (len > p.L) len = p.L;
      dst = p.dst_row[n];
    }
    s_n[tid] = n; s_len[tid] = len; s_dst[tid] = dst;
    atomicMax(&s_maxlen, len);
  }
  for (int e = tid; e < G3 * H; e += 256) Ws[e] = p.whh[dir][e];
  __syncthreads();
  const int maxlen = s_maxlen;
  const int hid = wh * 32 + l31;
  float dh[16];
  int nreg[16], lreg[16], dreg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dh[r] = 0.f;
    const int seq = ws * 32 + mfma_row(r, lane);
    nreg[r] = s_n[seq]; lreg[r] = s_len[seq]; dreg[r] = s_dst[seq];
  }
  f32x16 accw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
  float sbi[3] = {0.f, 0.f, 0.f}, sbh[3] = {0.f, 0.f, 0.f};

  for (int step = 0; step < maxlen; ++step) {
    // reverse of the forward order
    const int t = dir == 0 ? maxlen - 1 - step : step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    float dcarry[16];
    // all loads of the step first, unconditionally (clamped addresses), then the gate math with selects
    float ldo[16], lhp[16], lsv[4][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool act = t < lreg[r];
      const bool hasp = act && tp >= 0 && tp < lreg[r];
      const long orow = ((long)dreg[r] * p.L) * 128 + dir * H + hid;
      ldo[r] = p.dout[act ? orow + (long)t * 128 : 0];
      lhp[r] = p.out[hasp ? orow + (long)tp * 128 : 0];
      const float* sv = p.saved + (act ? ((((long)dir * p.N + nreg[r]) * p.L + t) * 4) * H + hid : 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) lsv[q][r] = sv[q * H];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int seq = ws * 32 + mfma_row(r, lane);
      const bool act = t < lreg[r];
      const bool hasp = act && tp >= 0 && tp < lreg[r];
      float drp = 0.f, dzp = 0.f, dghn = 0.f;
      const float hp = hasp ? lhp[r] : 0.f;
      dcarry[r] = dh[r];
      if (act) {
        const float dtot = dh[r] + ldo[r];
        const float rr = lsv[0][r], zz = lsv[1][r], nn = lsv[2][r], hn = lsv[3][r];
        const float dnp = dtot * (1.f - zz) * (1.f - nn * nn);
        dzp = dtot * (hp - nn) * zz * (1.f - zz);
        drp = dnp * hn * rr * (1.f - rr);
        dghn = dnp * rr;
        float* g = p.dgx + ((long)nreg[r] * p.L + t) * 384 + dir * G3 + hid;
        g[0] = drp; g[H] = dzp; g[2 * H] = dnp;
        dcarry[r] = dtot * zz;
        sbi[0] += drp; sbi[1] += dzp; sbi[2] += dnp;
        sbh[0] += drp; sbh[1] += dzp; sbh[2] += dghn;
      }
      DG[(hid)*LDG + seq] = drp;
      DG[(H + hid) * LDG + seq] = dzp;
      DG[(2 * H + hid) * LDG + seq] = dghn;
      HP[seq * H + hid] = hp;
    }
    __syncthreads();
    // dh_prev[seq][hid] += sum_gate dgh[seq][gate] * W_hh[gate][hid]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < G3 / 2; ++kk) {
      const int k = 2 * kk + kh;
      acc = mfma32(DG[k * LDG + ws * 32 + l31], Ws[k * H + hid], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = dcarry[r] + acc[r];  // inactive rows: dgh = 0 -> acc = 0, carry = dh
    // dW_hh[gate][hid] += sum_seq dgh[seq][gate] * h_prev[seq][hid]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = wave * 3 + i;
      const int gt = idx >> 1, ht = idx & 1;
#pragma unroll 8
      for (int kk = 0; kk < TS / 2; ++kk) {
        const int k = 2 * kk + kh;
        accw[i] = mfma32(DG[(gt * 32 + l31) * LDG + k], HP[k * H + ht * 32 + l31], accw[i]);
      }
    }
    __syncthreads();
  }

  float* slab = p.dwhh_slab + ((long)tile * 2 + dir) * G3 * H;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = wave * 3 + i;
    const int gt = idx >> 1, ht = idx & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(gt * 32 + mfma_row(r, lane)) * H + ht * 32 + l31] = accw[i][r];
  }
  // bias gradients: 4 threads (2 sequence halves x 2 lane halves) hold partial sums for each (gate, hid).  Summed in
  // a fixed order through LDS (DG is free after the loop's last barrier) - float atomics would make the result depend
  // on arrival order.
  {
    float* part = DG + (ws * 2 + kh) * (2 * G3);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      part[q * H + hid] = sbi[q];
      part[G3 + q * H + hid] = sbh[q];
    }
  }
  __syncthreads();
  float* bs = p.dbias_slab + ((long)tile * 2 + dir) * 2 * G3;
  for (int e = tid; e < 2 * G3; e += 256)
    bs[e] = ((DG[e] + DG[2 * G3 + e]) + DG[4 * G3 + e]) + DG[6 * G3 + e];
}


// ---------------------------------------------------------------------------------------------------------
// 16-sequence tiles on v_mfma_f32_16x16x4_f32 (the default).  One workgroup = 16 sequences x one direction, 4 waves,
// wave w owns hidden units 16w..16w+15 of all three gates.  Four times as many workgroups as the 64-sequence kernels
// above (N = 2560 sentences: 320 instead of 80), W_hh lives in registers for the whole sequence (48 VGPRs per lane:
// it is the same for every step), the hidden state is double-buffered in LDS (one barrier per step) and the next
// step's gx / saved gates are loaded while the current step's MFMAs run.
//   A lane l -> A[row = l&15][k-group l>>4], B lane l -> B[k-group l>>4][col = l&15], D reg r -> D[row = 4*(l>>4)+r][col = l&15].
// The k index a lane group feeds in k-step kk is free as long as A and B agree: group g takes k = 16g + kk (forward)
// so that a lane's 16 values of H / W_hh are contiguous (ds_read_b128 / float4 loads).
constexpr int TS2 = 16;
constexpr int LDH2 = 68;    // Hs[seq][LDH2]: 16 lanes x b128 at stride 68 floats touch 64 distinct banks
constexpr int LDG2 = 196;   // DGs[seq][LDG2]

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void gru_fwd16_kernel(GruFwdParams p) {
  __shared__ __attribute__((aligned(16))) float Hs[2][TS2 * LDH2];
  __shared__ int s_n[TS2], s_len[TS2], s_dst[TS2];
  __shared__ int s_maxlen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int dir = blockIdx.y, tile = blockIdx.x;

  if (tid == 0) s_maxlen = 0;
  __syncthreads();
  if (tid < TS2) {
    const int pos = tile * TS2 + tid;
    int n = -1, len = 0, dst = 0;
    if (pos < p.N) {
      n = p.order[pos];
      len = p.lengths[n];
      if (len > p.L) len = p.L;
      dst = p.dst_row[n];
    }
    s_n[tid] = n; s_len[tid] = len; s_dst[tid] = dst;
    atomicMax(&s_maxlen, len);
  }
  for (int e = tid; e < 2 * TS2 * LDH2; e += 256) (&Hs[0][0])[e] = 0.f;
  const int hid = wave * 16 + c;
  float w[3][16];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float4* src = reinterpret_cast<const float4*>(p.whh[dir] + (long)(q * H + hid) * H + 16 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = src[i];
      w[q][4 * i] = v.x; w[q][4 * i + 1] = v.y; w[q][4 * i + 2] = v.z; w[q][4 * i + 3] = v.w;
    }
  }
  float bh[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) bh[q] = p.bhh[dir][q * H + hid];
  __syncthreads();
  const int maxlen = s_maxlen;
  float hreg[4];
  int nreg[4], lreg[4], dreg[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    hreg[r] = 0.f;
    nreg[r] = s_n[4 * g + r]; lreg[r] = s_len[4 * g + r]; dreg[r] = s_dst[4 * g + r];
  }
  // positions past a sequence's length read as zero (pad_packed_sequence): written here instead of a memset of the whole
  // tensor before the launch (dst_row is a permutation, so every output row belongs to exactly one sequence)
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (nreg[r] >= 0)
      for (int t = lreg[r]; t < p.L; ++t) p.out[((long)dreg[r] * p.L + t) * 128 + dir * H + hid] = 0.f;
  auto load_gx = [&](int t, float (&dst)[3][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // unconditional (clamped) loads: a per-lane `act ? load : 0` makes hipcc branch and wait per load
      const bool act = t >= 0 && t < lreg[r];
      const float* gp = p.gx + (act ? ((long)nreg[r] * p.L + t) * 384 + dir * G3 + hid : 0);
#pragma unroll
      for (int q = 0; q < 3; ++q) dst[q][r] = gp[q * H];
    }
  };
  float gxn[3][4];
  load_gx(dir == 0 ? 0 : maxlen - 1, gxn);

  for (int step = 0; step < maxlen; ++step) {
    const int t = dir == 0 ? step : maxlen - 1 - step;
    const int tn = step + 1 < maxlen ? (dir == 0 ? t + 1 : t - 1) : -1;
    float gxr[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) gxr[q][r] = gxn[q][r];
    load_gx(tn, gxn);                                   // in flight under this step's MFMAs
    const float* hs = Hs[step & 1];
    float av[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(hs + c * LDH2 + 16 * g + 4 * i);
      av[4 * i] = v.x; av[4 * i + 1] = v.y; av[4 * i + 2] = v.z; av[4 * i + 3] = v.w;
    }
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = bh[q];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] = mfma16(av[kk], w[q][kk], acc[q]);
    float* hn_s = Hs[(step + 1) & 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = t < lreg[r];
      if (act) {
        const float rr = sigmoidf_(gxr[0][r] + acc[0][r]);
        const float zz = sigmoidf_(gxr[1][r] + acc[1][r]);
        const float hn = acc[2][r];
        const float nn = tanhf(gxr[2][r] + rr * hn);
        const float hnew = (1.f - zz) * nn + zz * hreg[r];
        hreg[r] = hnew;
        p.out[((long)dreg[r] * p.L + t) * 128 + dir * H + hid] = hnew;
        if (p.saved) {
          float* sv = p.saved + ((((long)dir * p.N + nreg[r]) * p.L + t) * 4) * H + hid;
          sv[0] = rr; sv[H] = zz; sv[2 * H] = nn; sv[3 * H] = hn;
        }
      }
      hn_s[(4 * g + r) * LDH2 + hid] = hreg[r];         // inactive rows carry their state into the other buffer
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gru_bwd16_kernel(GruBwdParams p) {
  __shared__ __attribute__((aligned(16))) float DGs[2][TS2 * LDG2];   // dgh[seq][gate]
  __shared__ __attribute__((aligned(16))) float HPs[2][TS2 * LDH2];   // h_prev[seq][hid]
  __shared__ int s_n[TS2], s_len[TS2], s_dst[TS2];
  __shared__ int s_maxlen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int dir = blockIdx.y, tile = blockIdx.x;

  if (tid == 0) s_maxlen = 0;
  __syncthreads();
  if (tid < TS2) {
    const int pos = tile * TS2 + tid;
    int n = -1, len = 0, dst = 0;
    if (pos < p.N) {
      n = p.order[pos];
      len = p.lengths[n];
      if (len > p.L) len = p.L;
      dst = p.dst_row[n];
    }
    s_n[tid] = n; s_len[tid] = len; s_dst[tid] = dst;
    atomicMax(&s_maxlen, len);
  }
  const int hid = wave * 16 + c;
  // dh_prev[seq][hid] = sum_j dgh[seq][j] W_hh[j][hid]: lane group g takes j = 48g + kk, its 48 weights stay in registers
  float w1[48];
#pragma unroll
  for (int kk = 0; kk < 48; ++kk) w1[kk] = p.whh[dir][(long)(48 * g + kk) * H + hid];
  __syncthreads();
  const int maxlen = s_maxlen;
  float dh[4];
  int nreg[4], lreg[4], dreg[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dh[r] = 0.f;
    nreg[r] = s_n[4 * g + r]; lreg[r] = s_len[4 * g + r]; dreg[r] = s_dst[4 * g + r];
  }
  // no gradient reaches the input projections of padded positions: zeros, written here instead of a memset before the launch
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (nreg[r] >= 0)
      for (int t = lreg[r]; t < p.L; ++t) {
        float* gp = p.dgx + ((long)nreg[r] * p.L + t) * 384 + dir * G3 + hid;
        gp[0] = 0.f; gp[H] = 0.f; gp[2 * H] = 0.f;
      }
  // dW_hh[j][hid'] += sum_seq dgh[seq][j] h_prev[seq][hid'], computed transposed: rows hid' (tile ht), columns this
  // wave's j = q*64 + hid; k-step kk of group g is sequence 4g + kk = the row this lane computed itself (register kk)
  f32x4 accw[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int ht = 0; ht < 4; ++ht)
#pragma unroll
      for (int r = 0; r < 4; ++r) accw[q][ht][r] = 0.f;
  float sbi[3] = {0.f, 0.f, 0.f}, sbh[3] = {0.f, 0.f, 0.f};

  struct StepLoads { float ldo[4], lhp[4], lsv[4][4]; };
  auto load_step = [&](int t, int tp, StepLoads& d) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = t >= 0 && t < lreg[r];
      const bool hasp = act && tp >= 0 && tp < lreg[r];
      const long orow = ((long)dreg[r] * p.L) * 128 + dir * H + hid;
      d.ldo[r] = p.dout[act ? orow + (long)t * 128 : 0];
      d.lhp[r] = p.out[hasp ? orow + (long)tp * 128 : 0];
      const float* sv = p.saved + (act ? ((((long)dir * p.N + nreg[r]) * p.L + t) * 4) * H + hid : 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) d.lsv[q][r] = sv[q * H];
    }
  };
  StepLoads nx;
  {
    const int t0 = dir == 0 ? maxlen - 1 : 0;
    load_step(maxlen > 0 ? t0 : -1, dir == 0 ? t0 - 1 : t0 + 1, nx);
  }

  for (int step = 0; step < maxlen; ++step) {
    // reverse of the forward order
    const int t = dir == 0 ? maxlen - 1 - step : step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const StepLoads cur = nx;
    {
      const int t2 = step + 1 < maxlen ? (dir == 0 ? t - 1 : t + 1) : -1;
      load_step(t2, dir == 0 ? t2 - 1 : t2 + 1, nx);
    }
    float* dgs = DGs[step & 1];
    float* hps = HPs[step & 1];
    float dcarry[4], dv[3][4], hpv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool act = t < lreg[r];
      const bool hasp = act && tp >= 0 && tp < lreg[r];
      float drp = 0.f, dzp = 0.f, dghn = 0.f;
      const float hp = hasp ? cur.lhp[r] : 0.f;
      dcarry[r] = dh[r];
      if (act) {
        const float dtot = dh[r] + cur.ldo[r];
        const float rr = cur.lsv[0][r], zz = cur.lsv[1][r], nn = cur.lsv[2][r], hn = cur.lsv[3][r];
        const float dnp = dtot * (1.f - zz) * (1.f - nn * nn);
        dzp = dtot * (hp - nn) * zz * (1.f - zz);
        drp = dnp * hn * rr * (1.f - rr);
        dghn = dnp * rr;
        float* gp = p.dgx + ((long)nreg[r] * p.L + t) * 384 + dir * G3 + hid;
        gp[0] = drp; gp[H] = dzp; gp[2 * H] = dnp;
        dcarry[r] = dtot * zz;
        sbi[0] += drp; sbi[1] += dzp; sbi[2] += dnp;
        sbh[0] += drp; sbh[1] += dzp; sbh[2] += dghn;
      }
      dv[0][r] = drp; dv[1][r] = dzp; dv[2][r] = dghn; hpv[r] = hp;
      float* row = dgs + (4 * g + r) * LDG2 + hid;
      row[0] = drp; row[H] = dzp; row[2 * H] = dghn;
      hps[(4 * g + r) * LDH2 + hid] = hp;
    }
    __syncthreads();
    // dh_prev: three accumulator chains over j = 48g + kk
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(dgs + c * LDG2 + 48 * g + 4 * i);
      acc[0] = mfma16(v.x, w1[4 * i], acc[0]);
      acc[1] = mfma16(v.y, w1[4 * i + 1], acc[1]);
      acc[2] = mfma16(v.z, w1[4 * i + 2], acc[2]);
      acc[0] = mfma16(v.w, w1[4 * i + 3], acc[0]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = dcarry[r] + ((acc[0][r] + acc[1][r]) + acc[2][r]);  // inactive rows: acc = 0
    // dW_hh
#pragma unroll
    for (int ht = 0; ht < 4; ++ht) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float a = hps[(4 * g + kk) * LDH2 + 16 * ht + c];
#pragma unroll
        for (int q = 0; q < 3; ++q) accw[q][ht] = mfma16(a, dv[q][kk], accw[q][ht]);
      }
    }
    // no second barrier: the next step writes the other DGs / HPs buffer, and the step after that is behind the
    // next step's barrier, which every wave reaches only after these reads
  }

  float* slab = p.dwhh_slab + ((long)tile * 2 + dir) * G3 * H;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int ht = 0; ht < 4; ++ht)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(long)(q * H + hid) * H + 16 * ht + 4 * g + r] = accw[q][ht][r];
  // bias gradients: the four lane groups hold partial sums (4 sequences each) of the same (gate, hid): fixed-order sum
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    sbi[q] += __shfl_xor(sbi[q], 16, 64); sbi[q] += __shfl_xor(sbi[q], 32, 64);
    sbh[q] += __shfl_xor(sbh[q], 16, 64); sbh[q] += __shfl_xor(sbh[q], 32, 64);
  }
  if (g == 0) {
    float* bs = p.dbias_slab + ((long)tile * 2 + dir) * 2 * G3;
#pragma unroll
    for (int q = 0; q < 3; ++q) { bs[q * H + hid] = sbi[q]; bs[G3 + q * H + hid] = sbh[q]; }
  }
}

// dst[j] (+)= sum_i src[i][j]   (rows x cols, fixed order)
__global__ void colsum_rows_kernel(const float* __restrict__ src, int rows, long cols, long row_stride,
                                   float* __restrict__ dst, int accumulate) {
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < cols; j += (long)gridDim.x * blockDim.x) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;   // four chains in flight, combined in a fixed order
    int i = 0;
    for (; i + 3 < rows; i += 4) {
      v0 += src[(long)i * row_stride + j]; v1 += src[(long)(i + 1) * row_stride + j];
      v2 += src[(long)(i + 2) * row_stride + j]; v3 += src[(long)(i + 3) * row_stride + j];
    }
    for (; i < rows; ++i) v0 += src[(long)i * row_stride + j];
    const float v = (v0 + v1) + (v2 + v3);
    dst[j] = accumulate ? dst[j] + v : v;
  }
}

}  // namespace

int umpr_colsum_rows(const float* src, int rows, long cols, long row_stride, float* dst, int accumulate, hipStream_t s) {
  int blocks = (int)((cols + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  colsum_rows_kernel<<<blocks, 256, 0, s>>>(src, rows, cols, row_stride, dst, accumulate);
  UMPR_LAUNCH_CHECK("colsum_rows");
  return 0;
}

// UMPR_GRU_V1=1: the 64-sequence kernels (A/B runs)
static const bool g_gru_v1 = (umpr_env_int("UMPR_GRU_V1", 0) == 1);

int umpr_gru_recurrent_fwd(const float* gx, const float* whh_f, const float* bhh_f, const float* whh_r,
                           const float* bhh_r, const int* lengths, const int* order, const int* dst_row, float* out,
                           float* saved, int N, int L, hipStream_t s) {
  GruFwdParams p;
  p.gx = gx; p.whh[0] = whh_f; p.whh[1] = whh_r; p.bhh[0] = bhh_f; p.bhh[1] = bhh_r;
  p.lengths = lengths; p.order = order; p.dst_row = dst_row; p.out = out; p.saved = saved; p.N = N; p.L = L;
  if (g_gru_v1 && hipMemsetAsync(out, 0, (size_t)N * L * 128 * sizeof(float), s) != hipSuccess) {   // gru_fwd16 zeroes the tails itself
    umpr_set_error("gru_fwd: memset failed");
    return -2;
  }
  dim3 grid(umpr_gru_tiles(N), 2);
  // family 3 counts BYTES (HBM/latency-bound kernel): per token and both directions gx 384 floats read, out 128
  // written, saved gates 2*4*64 written when training
  UmprProfScope prof(UMPR_K_GRU, 4.0 * N * L * (384 + 128 + (saved ? 512 : 0)), s);
  if (g_gru_v1) gru_fwd_kernel<<<grid, 256, 0, s>>>(p);
  else gru_fwd16_kernel<<<grid, 256, 0, s>>>(p);
  UMPR_LAUNCH_CHECK("gru_fwd");
  return 0;
}

int umpr_gru_tiles(int N) { return cdiv(N, g_gru_v1 ? TS : TS2); }

int umpr_gru_bptt(const float* dout, const float* out, const float* saved, const float* whh_f, const float* whh_r,
                  const int* lengths, const int* order, const int* dst_row, float* dgx, float* dwhh_slab,
                  float* dbias_slab, int N, int L, hipStream_t s) {
  GruBwdParams p;
  p.dout = dout; p.out = out; p.saved = saved; p.whh[0] = whh_f; p.whh[1] = whh_r;
  p.lengths = lengths; p.order = order; p.dst_row = dst_row; p.dgx = dgx; p.dwhh_slab = dwhh_slab;
  p.dbias_slab = dbias_slab; p.N = N; p.L = L;
  if (g_gru_v1 && hipMemsetAsync(dgx, 0, (size_t)N * L * 384 * sizeof(float), s) != hipSuccess) {   // gru_bwd16 zeroes the tails itself
    umpr_set_error("gru_bwd: memset failed");
    return -2;
  }
  dim3 grid(umpr_gru_tiles(N), 2);
  // bytes per token: dout 128 + out 128 + saved 512 read, dgx 384 written
  UmprProfScope prof(UMPR_K_GRU, 4.0 * N * L * (128 + 128 + 512 + 384), s);
  if (g_gru_v1) gru_bwd_kernel<<<grid, 256, 0, s>>>(p);
  else gru_bwd16_kernel<<<grid, 256, 0, s>>>(p);
  UMPR_LAUNCH_CHECK("gru_bwd");
  return 0;
}
