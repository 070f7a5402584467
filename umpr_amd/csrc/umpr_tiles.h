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

  __device__ __forceinline__ void load(const float* __restrict__ base, long ld, const int64_t* __restrict__ gather,
                                       int mn0, int MN, int k0, int kend, int vec_ok, int tid) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KCONTIG) {
        const int mn = e >> 2, kq = e & 3;
        const int gm = mn0 + mn, gk = k0 + 4 * kq;
        long row = -1;
        if (gm < MN) row = gather ? (long)gather[gm] : (long)gm;
        if (row >= 0 && gk < kend) {
          const float* p = base + row * ld + gk;
          if (vec_ok && gk + 3 < kend) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = p[0];
            if (gk + 1 < kend) val.y = p[1];
            if (gk + 2 < kend) val.z = p[2];
            if (gk + 3 < kend) val.w = p[3];
          }
        }
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        const int gk = k0 + k, gm = mn0 + 4 * q;
        long row = -1;
        if (gk < kend) row = gather ? (long)gather[gk] : (long)gk;
        if (row >= 0 && gm < MN) {
          const float* p = base + row * ld + gm;
          if (vec_ok && gm + 3 < MN) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = p[0];
            if (gm + 1 < MN) val.y = p[1];
            if (gm + 2 < MN) val.z = p[2];
            if (gm + 3 < MN) val.w = p[3];
          }
        }
      }
      r[v] = val;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int e = tid + v * 256;
      if (KCONTIG) {
        const int mn = e >> 2, kq = e & 3;
        S[(4 * kq + 0) * LD + mn] = r[v].x;
        S[(4 * kq + 1) * LD + mn] = r[v].y;
        S[(4 * kq + 2) * LD + mn] = r[v].z;
        S[(4 * kq + 3) * LD + mn] = r[v].w;
      } else {
        constexpr int QPR = TILE / 4;
        const int k = e / QPR, q = e % QPR;
        *reinterpret_cast<float4*>(&S[k * LD + 4 * q]) = r[v];
      }
    }
  }
};

