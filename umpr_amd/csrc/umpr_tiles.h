// Register-staged operand tiles shared by the GEMM and the implicit-GEMM convolution kernels.
#pragma once
#include "umpr_common.h"

constexpr int BK = 16;  // k-depth of one LDS stage (8 v_mfma_f32_32x32x2_f32 steps)
constexpr int KFLUSH = 8;  // k-tiles per MFMA accumulation chain before folding into the running total

// One operand tile (TILE x BK) held in registers between the global load and the LDS store.
// KCONTIG: global rows are the tile's m/n index, contiguous along k.  else: global rows are k, contiguous along m/n.
// BKT: k-depth of the stage (16 everywhere except the bf16 GEMM, whose stages are bound by the latency of their global loads:
// 32 there halves the number of round trips).
template <int TILE, bool KCONTIG, int BKT = BK>
struct TileRegs {
  static constexpr int NV = TILE * BKT / 4 / 256;
  static constexpr int KQ = BKT / 4;                   // float4 per k-contiguous row of the stage
  static constexpr int LD = KCONTIG ? TILE + 2 : TILE + 4;
  float4 r[NV];

  // Branch-free AND wait-free at load time: every lane always issues its loads (out-of-range lanes read element 0 of
  // the operand); validity is kept as a bit mask and the zero-select happens in store(), i.e. after the MFMAs of the
  // current tile.  A per-lane `if (valid) v = load`, or a select right behind the load, makes hipcc wait vmcnt(0)
  // per load: the loads then complete one L2 round trip after the other (rocprofv3: SQ_WAIT_ANY was 43% of the wave
  // lifetime).  vec_ok (wave-uniform) promises 16-B alignment and extents that are multiples of 4.
  unsigned okmask;
  // gathered rows of a k-contiguous operand are the same for every k-stage: looked up once (rows_cached), not once per stage
  // in front of the dependent data load (the embedding gather of the GRU input projection: two global round trips per stage)
  long grow[NV];
  bool rows_cached = false;
  __device__ __forceinline__ void cache_rows(const int64_t* __restrict__ gather, int mn0, int MN, int tid) {
    if (!KCONTIG || !gather) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int gm = mn0 + (tid + v * 256) / KQ;
      grow[v] = gather[gm < MN ? gm : 0] | (gm < MN ? 0L : -1L);
    }
    rows_cached = true;
  }
  // winL > 0: sliding-window operand (UmprGemm::winA / winB) - the operand's row index (mn when KCONTIG, k otherwise) is a
  // sentence position, element (row, col) lives at base[(row - winPad) * ld + col] and exists iff
  // 0 <= row % winL + col / winD - winPad < winL.  A float4 never straddles a window step (winD % 4 == 0).
  __device__ __forceinline__ void load(const float* __restrict__ base, long ld, const int64_t* __restrict__ gather,
                                       int mn0, int MN, int k0, int kend, int vec_ok, int tid, int winL = 0, int winD = 1,
                                       int winPad = 0) {
    okmask = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      int o0, o1, o2, o3;
      const float* p;
      if (KCONTIG) {
        const int mn = e / KQ, kq = e % KQ;
        const int gm = mn0 + mn, gk = k0 + 4 * kq;
        long row = gm < MN ? (long)gm : -1;
        if (gather) row = rows_cached ? grow[v] : (gather[gm < MN ? gm : 0] | (gm < MN ? 0L : -1L));
        bool rv = row >= 0;
        if (winL) { const int tpos = gm % winL + gk / winD - winPad; rv = rv && tpos >= 0 && tpos < winL; row -= winPad; }
        p = base + (rv ? row * ld + gk : 0);
        o0 = rv && gk < kend; o1 = rv && gk + 1 < kend; o2 = rv && gk + 2 < kend; o3 = rv && gk + 3 < kend;
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        const int gk = k0 + k, gm = mn0 + 4 * q;
        long row = gk < kend ? (long)gk : -1;
        if (gather) row = gather[gk < kend ? gk : 0] | (gk < kend ? 0L : -1L);
        bool rv = row >= 0;
        if (winL) { const int tpos = gk % winL + gm / winD - winPad; rv = rv && tpos >= 0 && tpos < winL; row -= winPad; }
        p = base + (rv ? row * ld + gm : 0);
        o0 = rv && gm < MN; o1 = rv && gm + 1 < MN; o2 = rv && gm + 2 < MN; o3 = rv && gm + 3 < MN;
      }
      if (vec_ok) {
        r[v] = *reinterpret_cast<const float4*>(o0 ? p : base);
        o1 = o2 = o3 = o0;
      } else {
        r[v].x = *(o0 ? p : base); r[v].y = *(o1 ? p + 1 : base); r[v].z = *(o2 ? p + 2 : base); r[v].w = *(o3 ? p + 3 : base);
      }
      okmask |= (unsigned)(o0 | (o1 << 1) | (o2 << 2) | (o3 << 3)) << (4 * v);
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      const unsigned m = okmask >> (4 * v);
      const float4 val = make_float4((m & 1) ? r[v].x : 0.f, (m & 2) ? r[v].y : 0.f, (m & 4) ? r[v].z : 0.f, (m & 8) ? r[v].w : 0.f);
      if (KCONTIG) {
        const int mn = e / KQ, kq = e % KQ;
        S[(4 * kq + 0) * LD + mn] = val.x;
        S[(4 * kq + 1) * LD + mn] = val.y;
        S[(4 * kq + 2) * LD + mn] = val.z;
        S[(4 * kq + 3) * LD + mn] = val.w;
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        *reinterpret_cast<float4*>(&S[k * LD + 4 * q]) = val;
      }
    }
  }

  // bf16 stage image for v_mfma_f32_32x32x16_bf16: S[row][LDB16] bf16, row = the tile's m / n index, 16 k per row (32 B)
  // padded to 48 B - 16 lanes x ds_read_b128 at a 48-B stride touch 64 distinct banks
  // An operand whose global rows are k (KCONTIG = false) keeps that order in LDS - S[k][LDT16] bf16, one 8-B store per float4
  // (the row-major image would need four scattered 2-B stores that land 16-way on four banks) - and its MFMA fragments come
  // from ds_read_b64_tr_b16.  Row pitch 320 B: the 4 rows x 2 column groups a half-wave reads fall into 64 distinct banks.
  static constexpr int LDB16 = BKT + 8;               // 48-B rows for 16 k, 80-B rows for 32 k: both conflict-free for b128 reads
  static constexpr int LDT16 = 160;
  static constexpr int B16_ELEMS = KCONTIG ? TILE * LDB16 : BKT * LDT16;
  __device__ __forceinline__ void store_b16(__bf16* __restrict__ S, int tid) const {
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      const unsigned m = okmask >> (4 * v);
      const float4 val = make_float4((m & 1) ? r[v].x : 0.f, (m & 2) ? r[v].y : 0.f, (m & 4) ? r[v].z : 0.f, (m & 8) ? r[v].w : 0.f);
      if (KCONTIG) {
        const int mn = e / KQ, kq = e % KQ;
        bf16x4_ o;
        o[0] = (__bf16)val.x; o[1] = (__bf16)val.y; o[2] = (__bf16)val.z; o[3] = (__bf16)val.w;
        *reinterpret_cast<bf16x4_*>(&S[mn * LDB16 + 4 * kq]) = o;
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        bf16x4_ o;
        o[0] = (__bf16)val.x; o[1] = (__bf16)val.y; o[2] = (__bf16)val.z; o[3] = (__bf16)val.w;
        *reinterpret_cast<bf16x4_*>(&S[k * LDT16 + 4 * q]) = o;
      }
    }
  }

  // MFMA fragment (8 consecutive k of row `row0 + (lane & 31)`, k-half lane >> 5) of the stage image written by store_b16
  typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
  typedef __bf16 bf16x4t_ __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ bf16x8_ frag_b16(const __bf16* __restrict__ S, int row0, int lane, int ks = 0) {
    if (KCONTIG) {
      return *reinterpret_cast<const bf16x8_*>(&S[(row0 + (lane & 31)) * LDB16 + 16 * ks + 8 * (lane >> 5)]);
    } else {
      const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
      const __bf16* a = &S[(16 * ks + 8 * (g >> 1) + q4) * LDT16 + row0 + 16 * (g & 1) + 4 * p4];
      typedef __attribute__((address_space(3))) void* lds_void_t;
      const bf16x4t_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4t_ __attribute__((address_space(3)))*)(lds_void_t)(a));
      const bf16x4t_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4t_ __attribute__((address_space(3)))*)(lds_void_t)(a + 4 * LDT16));
      bf16x8_ r;
#pragma unroll
      for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
      return r;
    }
  }
};

