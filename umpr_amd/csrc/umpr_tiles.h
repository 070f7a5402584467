// Register-staged operand tiles shared by the GEMM and the implicit-GEMM convolution kernels.
#pragma once
#include "umpr_common.h"

constexpr int BK = 16;  // k-depth of one LDS stage (8 v_mfma_f32_32x32x2_f32 steps)
constexpr int KFLUSH = 8;  // k-tiles per MFMA accumulation chain before folding into the running total

// One operand tile (TILE x BK) held in registers between the global load and the LDS store.
// KCONTIG: global rows are the tile's m/n index, contiguous along k.  else: global rows are k, contiguous along m/n.
template <int TILE, bool KCONTIG>
struct TileRegs {
  static constexpr int NV = TILE * BK / 4 / 256;
  static constexpr int LD = KCONTIG ? TILE + 2 : TILE + 4;
  float4 r[NV];

  // Branch-free AND wait-free at load time: every lane always issues its loads (out-of-range lanes read element 0 of
  // the operand); validity is kept as a bit mask and the zero-select happens in store(), i.e. after the MFMAs of the
  // current tile.  A per-lane `if (valid) v = load`, or a select right behind the load, makes hipcc wait vmcnt(0)
  // per load: the loads then complete one L2 round trip after the other (rocprofv3: SQ_WAIT_ANY was 43% of the wave
  // lifetime).  vec_ok (wave-uniform) promises 16-B alignment and extents that are multiples of 4.
  unsigned okmask;
  __device__ __forceinline__ void load(const float* __restrict__ base, long ld, const int64_t* __restrict__ gather,
                                       int mn0, int MN, int k0, int kend, int vec_ok, int tid) {
    okmask = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      int o0, o1, o2, o3;
      const float* p;
      if (KCONTIG) {
        const int mn = e >> 2, kq = e & 3;
        const int gm = mn0 + mn, gk = k0 + 4 * kq;
        long row = gm < MN ? (long)gm : -1;
        if (gather) row = gather[gm < MN ? gm : 0] | (gm < MN ? 0L : -1L);
        const bool rv = row >= 0;
        p = base + (rv ? row * ld + gk : 0);
        o0 = rv && gk < kend; o1 = rv && gk + 1 < kend; o2 = rv && gk + 2 < kend; o3 = rv && gk + 3 < kend;
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        const int gk = k0 + k, gm = mn0 + 4 * q;
        long row = gk < kend ? (long)gk : -1;
        if (gather) row = gather[gk < kend ? gk : 0] | (gk < kend ? 0L : -1L);
        const bool rv = row >= 0;
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
        const int mn = e >> 2, kq = e & 3;
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
  static constexpr int LDB16 = 24;
  __device__ __forceinline__ void store_b16(__bf16* __restrict__ S, int tid) const {
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      const unsigned m = okmask >> (4 * v);
      const float4 val = make_float4((m & 1) ? r[v].x : 0.f, (m & 2) ? r[v].y : 0.f, (m & 4) ? r[v].z : 0.f, (m & 8) ? r[v].w : 0.f);
      if (KCONTIG) {
        const int mn = e >> 2, kq = e & 3;
        bf16x4_ o;
        o[0] = (__bf16)val.x; o[1] = (__bf16)val.y; o[2] = (__bf16)val.z; o[3] = (__bf16)val.w;
        *reinterpret_cast<bf16x4_*>(&S[mn * LDB16 + 4 * kq]) = o;
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        S[(4 * q + 0) * LDB16 + k] = (__bf16)val.x;
        S[(4 * q + 1) * LDB16 + k] = (__bf16)val.y;
        S[(4 * q + 2) * LDB16 + k] = (__bf16)val.z;
        S[(4 * q + 3) * LDB16 + k] = (__bf16)val.w;
      }
    }
  }
};

