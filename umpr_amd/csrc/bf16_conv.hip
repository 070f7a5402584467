// bf16 mixed-precision VGG16 convolution stack for gfx950 (BASELINE.json configs[4]): v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, activations and gradients in bf16, master weights / weight gradients in fp32.
//
// Activation layout "CB8-PF" (chosen for this hardware, not the reference's NCHW):
//   * channel blocks of 8: tensor = [C/8 planes][plane stride ps][8] bf16 - one pixel of one plane is 16 B, the unit of
//     ds_read_b128, of an MFMA k-half (8 consecutive k) and of one LDS-DMA lane;
//   * padded-flat pixels: image n, row y, column x lives at pixel P = (n*(H+1) + y + 1) * (W+1) + x + 1.  Column 0 of
//     every row and one row between (before, after) images are zero, so the 3x3 tap (dy,dx) of ANY pixel is pixel
//     P + dy*(W+1) + dx - no boundary handling anywhere.  Kernels produce pixels [0, ptot) incl. the zero pads; a
//     zeroed lead / tail guard around them keeps every tile's halo reads in bounds and finite.
// With this layout an operand tile is a handful of contiguous runs per plane: it goes global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging registers), lands conflict-free for the fragment
// reads ([plane][pixel] x 16 B: consecutive lanes = consecutive 16-B slots), the nine taps are an immediate offset on
// the fragment address (zero VALU in the loop), and the epilogue's bf16 stores are whole 512-B runs per wave-instruction.
//
//   conv_bf16_kernel   forward and data gradient: D[cout][pixel] = sum_{tap,c} Wp[tap][cout][c] * X[pixel + off(tap)][c],
//                      K-loop over (32-channel chunk, tap); weights for one step arrive 3 steps ahead in a 4-deep LDS ring
//                      (counted vmcnt), the input patch of a chunk serves its 9 taps.  1-D tiles of consecutive flat
//                      pixels on the narrow maps, 2-D tiles (smaller halo) on the 224 / 112 maps.
//   wgrad_bf16_kernel  weight gradient: dW[tap][co][ci] = sum_P dY[P][co] * X[P + off(tap)][ci], K = pixels; both MFMA
//                      operands are K-major in memory, so fragments come from ds_read_b64_tr_b16 (transposing LDS read);
//                      9 accumulator tiles per wave, split-K over pixel segments into fp32 slabs + deterministic reduce.
#include "umpr_common.h"
#include "umpr_internal.h"
#include <stdlib.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---- geometry ---------------------------------------------------------------------------------------------------
UmprPF umpr_pf(int N, int H, int W) {
  UmprPF g;
  g.N = N; g.H = H; g.W = W; g.RW = W + 1;
  g.rows = (long)N * (H + 1) + 1;
  g.ptot = g.rows * g.RW;
  g.lead = (long)align_up((size_t)g.RW + 2, 64);
  // tail: the last tile may start just below ptot and reads a patch of up to 18 map rows (2-D tiles) or 512 + 2 RW
  // pixels (1-D tiles) beyond its origin, rounded up to whole 64-pixel DMA pieces
  g.tail = (long)align_up((size_t)18 * g.RW + 512 + 2 * g.RW + 128, 64);
  g.ps = (long)align_up((size_t)(g.lead + g.ptot + g.tail), 64);
  return g;
}
size_t umpr_pf_bytes(const UmprPF& g, int C) { return (size_t)((C + 7) / 8) * g.ps * 16; }

namespace {

// LDS-DMA: 64 lanes x 16 B; LDS destination = M0 (wave-uniform byte address) + lane * 16, per-lane global source.
// Issued by inline asm so that hipcc neither waits vmcnt(0) behind it nor counts it; M0 is saved and restored.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_ptr_t)p; }

// pixel P of a map with row pitch RW and image pitch H+1 rows: a real pixel (not a zero pad)?
template <int RW>
__device__ __forceinline__ bool pf_real(long P, int H) {
  const unsigned row = (unsigned)P / (unsigned)RW;     // P < 2^32 for every map that fits the GPU
  const unsigned col = (unsigned)P - row * (unsigned)RW;
  return col != 0 && (row % (unsigned)(H + 1)) != 0;
}

// ------------------------------------------------------------------------------------------------------------------
// forward / data gradient
struct ConvB16Params {
  const bf16_t* x; long xps;     // input: pixel 0 of plane 0, plane stride in pixels
  const bf16_t* wp;              // packed weights [cout tile][chunk][tap][4 planes][BN][8]
  const float* bias;             // [M] or null
  const bf16_t* mask; long mps;  // dgrad: out *= (mask > 0), same layout as y; or null
  bf16_t* y; long yps;
  int C, M, H;                   // reduction channels (multiple of 32), output channels (multiple of BN), map height
  long ptot;                     // pixels [0, ptot) are stored
  long rows;                     // map rows incl. zero rows
  int relu, nct;
  long ntp;                      // pixel tiles
  long zlead, ztail;             // > 0: the kernel also zeroes the output's guards (zlead pixels before pixel 0, ztail
                                 // after ptot, per plane), a slice per pixel tile - saves a launch per layer in the backward
};

// guard pixels [g0, g1) of the zlead + ztail guard pixels of each of this channel tile's planes
template <int BN, int NT>
__device__ __forceinline__ void zero_guard_slice(const ConvB16Params& p, int ct, long pt, int tid) {
  const long G = p.zlead + p.ztail;
  if (G <= 0) return;
  const long g0 = pt * G / p.ntp, g1 = (pt + 1) * G / p.ntp;
  constexpr int PL = BN / 8;
  bf16x8 z;
#pragma unroll
  for (int q = 0; q < 8; ++q) z[q] = (bf16_t)0.f;
  for (long e = tid; e < (g1 - g0) * PL; e += NT) {
    const long gp = g0 + e / PL;
    const int pl = (int)(e % PL);
    const long P = gp < p.zlead ? gp - p.zlead : p.ptot + (gp - p.zlead);
    *reinterpret_cast<bf16x8*>(p.y + ((long)(ct * PL + pl) * p.yps + P) * 8) = z;
  }
}

// W: map width.  TR > 0: 2-D tile of TR rows x TC real columns; TR == 0: 1-D tile of TC consecutive flat pixels.
// BN output channels per workgroup; WP x WC waves (pixels x channels).
// ABL (ablation builds for timing only, results wrong): 1 = no per-step vmcnt wait / barrier, 2 = fragments read once,
// 3 = no DMA in the loop
template <int W, int TR, int TC, int BN, int WP, int WC, int D = 3, int ABL = 0>
__global__ __launch_bounds__(64 * WP * WC, (WP * WC == 4 ? 2 : 1)) void conv_bf16_kernel(ConvB16Params p) {
  constexpr int NW = WP * WC, NT = 64 * NW;
  constexpr bool TWO_D = TR > 0;
  constexpr int BM = TWO_D ? TR * TC : TC;
  constexpr int RW = W + 1;
  constexpr int LP = TC + 2;                      // 2-D patch row pitch (pixels)
  constexpr int TS = TWO_D ? LP : RW;             // LDS pixels between the rows a tap's dy selects
  constexpr int PATCH = TWO_D ? (TR + 2) * LP : BM + 2 * RW + 2;
  constexpr int PPP = (PATCH + 63) / 64;          // DMA pieces per plane
  constexpr int PPX = PPP * 64;                   // LDS pixels per plane
  constexpr int PPC = 4 * PPP;                    // patch pieces per 32-channel chunk
  constexpr int PPW = (PPC + NW - 1) / NW;        // ... per wave (surplus slots repeat a piece)
  constexpr int WPS = BN / 16;                    // weight pieces per step (BN x 64 B)
  constexpr int WPW = (WPS + NW - 1) / NW;
  constexpr int NB = D + 1;                       // weights arrive D steps ahead in a ring of NB stages
  constexpr int TPW = BM / WP / 32, TCW = BN / WC / 32;
  constexpr int WB = BN * 32, PB = 4 * PPX * 8;   // elements per weight stage / patch stage
  static_assert(BM % (32 * WP) == 0 && BN % (32 * WC) == 0 && (!TWO_D || (W % TC == 0 && TC % 16 == 0)), "tile shape");
  __shared__ __attribute__((aligned(1024))) bf16_t smem[NB * WB + 2 * PB];
  bf16_t* const Wb = smem;
  bf16_t* const Pb = smem + NB * WB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WC, wco = wave % WC;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware order: workgroups b, b+8, ... share an XCD.  The weights of one output-channel tile (up to 2.4 MB) are what
  // every pixel tile re-reads, so an XCD works on ONE channel tile where the tile count divides 8.
  const int xcd = blockIdx.x & 7;
  const long slot = blockIdx.x >> 3;
  int ct; long pt;
  if (p.nct <= 8 && (8 % p.nct) == 0) { const int g = 8 / p.nct; ct = xcd / g; pt = slot * g + (xcd % g); }
  else { ct = (int)(slot % p.nct); pt = (slot / p.nct) * 8 + xcd; }
  if (pt >= p.ntp) return;

  long Q0; int ctile = 0;
  if (TWO_D) {
    constexpr int CT = W / TC;
    const long band = pt / CT;
    ctile = (int)(pt - band * CT);
    Q0 = band * TR * RW + 1 + ctile * TC;
  } else {
    Q0 = pt * BM;
  }

  // ---- DMA plan of this wave
  const int nchunks = p.C / 32;
  const bf16_t* psrc[PPW]; unsigned pdst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = (wave + NW * i) % PPC;
    const int pl = q / PPP, pp = q - pl * PPP;
    int u = pp * 64 + lane;
    long gp;
    if (TWO_D) {
      if (u > PATCH - 1) u = PATCH - 1;
      const int a = u / LP, b = u - a * LP;
      gp = Q0 + (long)(a - 1) * RW + (b - 1);
    } else {
      gp = Q0 - RW - 1 + u;
    }
    psrc[i] = p.x + ((long)pl * p.xps + gp) * 8;
    pdst[i] = lds_addr(Pb) + (unsigned)(pl * PPX + pp * 64) * 16u;
  }
  const bf16_t* wsrc[WPW]; unsigned wdst[WPW];
  const bf16_t* wtile = p.wp + (long)ct * nchunks * 9 * WB;
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int q = (wave + NW * i) % WPS;
    wsrc[i] = wtile + q * 512 + lane * 8;
    wdst[i] = lds_addr(Wb) + (unsigned)q * 1024u;
  }
  const long chunk_stride = 4 * p.xps * 8;        // elements between 32-channel chunks of the input
  const int S = nchunks * 9;
  auto issue_w = [&](int s) {                     // stage of step s -> ring slot s % NB (steps past the end repeat the last)
    const int sc = s < S ? s : S - 1;
    const unsigned slot_off = (unsigned)(s % NB) * (WB * 2u);
#pragma unroll
    for (int i = 0; i < WPW; ++i) glds16(wsrc[i] + (long)sc * WB, wdst[i] + slot_off);
  };
  auto issue_p = [&](int c) {                     // patch of chunk c -> buffer c & 1
    const int cc = c < nchunks ? c : nchunks - 1;
    const unsigned buf_off = (unsigned)(c & 1) * (PB * 2u);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(psrc[i] + cc * chunk_stride, pdst[i] + buf_off);
  };

  // ---- fragment addresses (bytes inside a stage)
  unsigned pbo[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int k = wpx * (BM / WP) + j * 32 + r;
    const int idx = TWO_D ? (k / TC + 1) * LP + (k % TC) + 1 : k + RW + 1;
    pbo[j] = (unsigned)(h * PPX + idx) * 16u;
  }
  const unsigned wbo = (unsigned)(h * BN + wco * (BN / WC) + r) * 16u;

  // accumulators start at the bias: D[cout][pixel], register x of a lane is cout (x&3) + 8*(x>>2) + 4*h of the tile
  f32x16 acc[TCW][TPW];
#pragma unroll
  for (int i = 0; i < TCW; ++i) {
    f32x16 b0;
#pragma unroll
    for (int x = 0; x < 16; ++x)
      b0[x] = p.bias ? p.bias[ct * BN + wco * (BN / WC) + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h] : 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = b0;
  }

  issue_p(0);
#pragma unroll
  for (int d = 0; d < D; ++d) issue_w(d);
  wait_vm<(D - 1) * WPW>();
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const char* pb = reinterpret_cast<const char*>(Pb) + (c & 1) * (PB * 2);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = c * 9 + t;
      if (ABL != 3) {
        issue_w(s + D);
        if (t == 0) issue_p(c + 1);
      }
      const char* wb = reinterpret_cast<const char*>(Wb) + (s % NB) * (WB * 2);
      constexpr int dummy = 0; (void)dummy;
      const int toff = ((t / 3) - 1) * TS + (t % 3) - 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[TCW], b[TPW];
        if (ABL == 2) {
#pragma unroll
          for (int i = 0; i < TCW; ++i) { a[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Wb) + wbo + i * 512); asm volatile("" : "+v"(a[i])); }
#pragma unroll
          for (int j = 0; j < TPW; ++j) { b[j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(Pb) + pbo[j]); asm volatile("" : "+v"(b[j])); }
        } else {
#pragma unroll
        for (int i = 0; i < TCW; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wb + wbo + (ks * 2 * BN + i * 32) * 16);
#pragma unroll
        for (int j = 0; j < TPW; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + pbo[j] + (ks * 2 * PPX + toff) * 16);
        }
#pragma unroll
        for (int i = 0; i < TCW; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      // stage s+1 (and, from the fourth step of a chunk on, the next chunk's patch) must have landed; the younger DMAs
      // (D-1 weight stages, plus the patch while it may still fly) stay in flight across the barrier
      if (ABL != 1 && ABL != 3) {
        if (t < D) wait_vm<(D - 1) * WPW + PPW>(); else wait_vm<(D - 1) * WPW>();
      }
      if (ABL != 1) __syncthreads();
    }
  }
  wait_vm<0>();

  // ---- epilogue: ReLU / mask, zero at pads, bf16, 8-B stores that pair up to whole 16-B pixels across the half-waves
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int k = wpx * (BM / WP) + j * 32 + r;
    const long P = TWO_D ? Q0 + (long)(k / TC) * RW + (k % TC) : Q0 + k;
    if (P >= p.ptot) continue;
    const bool real = pf_real<RW>(P, p.H);
    // all mask words of this pixel first (independent loads in flight together), then the stores: loaded one by one
    // between the stores, each would wait out a full memory round trip (the compiler cannot move a load above a store
    // that may alias it) - that cost the 64-channel data gradient 250 us
    bf16x4 mk[TCW][4];
    if (p.mask) {
#pragma unroll
      for (int i = 0; i < TCW; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = ct * BN + wco * (BN / WC) + i * 32 + 8 * g + 4 * h;
          mk[i][g] = *reinterpret_cast<const bf16x4*>(p.mask + ((long)(co >> 3) * p.mps + P) * 8 + (co & 7));
        }
    }
#pragma unroll
    for (int i = 0; i < TCW; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = ct * BN + wco * (BN / WC) + i * 32 + 8 * g + 4 * h;
        const long o = ((long)(co >> 3) * p.yps + P) * 8 + (co & 7);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (float)mk[i][g][e] > 0.f ? v[e] : 0.f;
        }
        bf16x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (bf16_t)(real ? v[e] : 0.f);
        *reinterpret_cast<bf16x4*>(p.y + o) = out;
      }
    }
  }
  zero_guard_slice<BN, NT>(p, ct, pt, tid);
  if (TWO_D && ctile == 0) {   // column 0 (the zero pad) of this tile's rows: no tile computes it
    const long row0 = Q0 / RW;
    for (int e = tid; e < TR * (BN / 8); e += NT) {
      const int i = e % TR, cb = e / TR;
      const long P = (row0 + i) * RW;
      if (P < p.ptot) {
        bf16x8 z;
#pragma unroll
        for (int q = 0; q < 8; ++q) z[q] = (bf16_t)0.f;
        *reinterpret_cast<bf16x8*>(p.y + ((long)(ct * (BN / 8) + cb) * p.yps + P) * 8) = z;
      }
    }
  }
}

// The same kernel on v_mfma_f32_16x16x32_bf16: one MFMA covers the whole 32-channel chunk of a tap for a 16 px x 16 ch tile
// (lane group l >> 4 reads plane l >> 4).  Same LDS traffic and MFMA cycles per step as the 32x32x16 form; the guide
// measures a higher sustained clock for this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7).
// PP (8-wave workgroups only): ping-pong schedule.  The two waves a SIMD holds (w and w + 4) alternate a LOAD phase (DMA
// issue + the step's 12 fragment reads) and a COMPUTE phase (its 32 MFMAs) separated by s_barrier, the second half of the
// workgroup running one barrier behind the first, so one wave of every SIMD multiplies while its partner reads
// (MI355X_MICROARCH.md, "Two waves per SIMD").  Ring safety: a stage is overwritten one full step after its last read and
// every wave's reads are drained (lgkmcnt(0)) before the barrier that ends its load phase.
template <int W, int TR, int TC, int BN, int WP, int WC, int D = 3, bool PP = false>
__global__ __launch_bounds__(64 * WP * WC, (WP * WC == 4 ? (TR == 4 ? 3 : 2) : 1)) void conv_bf16_m16_kernel(ConvB16Params p) {
  constexpr int ABL = 0;
  static_assert(!PP || WP * WC == 8, "ping-pong needs two waves per SIMD in one workgroup");
  constexpr int NW = WP * WC, NT = 64 * NW;
  constexpr bool TWO_D = TR > 0;
  constexpr int BM = TWO_D ? TR * TC : TC;
  constexpr int RW = W + 1;
  constexpr int LP = TC + 2;                      // 2-D patch row pitch (pixels)
  constexpr int TS = TWO_D ? LP : RW;             // LDS pixels between the rows a tap's dy selects
  constexpr int PATCH = TWO_D ? (TR + 2) * LP : BM + 2 * RW + 2;
  constexpr int PPP = (PATCH + 63) / 64;          // DMA pieces per plane
  constexpr int PPX = PPP * 64;                   // LDS pixels per plane
  constexpr int PPC = 4 * PPP;                    // patch pieces per 32-channel chunk
  constexpr int PPW = (PPC + NW - 1) / NW;        // ... per wave (surplus slots repeat a piece)
  constexpr int WPS = BN / 16;                    // weight pieces per step (BN x 64 B)
  constexpr int WPW = (WPS + NW - 1) / NW;
  constexpr int NB = D + 1;                       // weights arrive D steps ahead in a ring of NB stages
  constexpr int TPW = BM / WP / 16, TCW = BN / WC / 16;   // 16 x 16 MFMA tiles per wave
  constexpr int WB = BN * 32, PB = 4 * PPX * 8;   // elements per weight stage / patch stage
  static_assert(BM % (32 * WP) == 0 && BN % (32 * WC) == 0 && (!TWO_D || (W % TC == 0 && TC % 16 == 0)), "tile shape");
  __shared__ __attribute__((aligned(1024))) bf16_t smem[NB * WB + 2 * PB];
  bf16_t* const Wb = smem;
  bf16_t* const Pb = smem + NB * WB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpx = wave / WC, wco = wave % WC;
  const int r = lane & 15, h = lane >> 4;                 // tile row / column index and plane (k-quarter) of this lane

  // XCD-aware order: workgroups b, b+8, ... share an XCD.  The weights of one output-channel tile (up to 2.4 MB) are what
  // every pixel tile re-reads, so an XCD works on ONE channel tile where the tile count divides 8.
  const int xcd = blockIdx.x & 7;
  const long slot = blockIdx.x >> 3;
  int ct; long pt;
  if (p.nct <= 8 && (8 % p.nct) == 0) { const int g = 8 / p.nct; ct = xcd / g; pt = slot * g + (xcd % g); }
  else { ct = (int)(slot % p.nct); pt = (slot / p.nct) * 8 + xcd; }
  if (pt >= p.ntp) return;

  long Q0; int ctile = 0;
  if (TWO_D) {
    constexpr int CT = W / TC;
    const long band = pt / CT;
    ctile = (int)(pt - band * CT);
    Q0 = band * TR * RW + 1 + ctile * TC;
  } else {
    Q0 = pt * BM;
  }

  // ---- DMA plan of this wave
  const int nchunks = p.C / 32;
  const bf16_t* psrc[PPW]; unsigned pdst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int q = (wave + NW * i) % PPC;
    const int pl = q / PPP, pp = q - pl * PPP;
    int u = pp * 64 + lane;
    long gp;
    if (TWO_D) {
      if (u > PATCH - 1) u = PATCH - 1;
      const int a = u / LP, b = u - a * LP;
      gp = Q0 + (long)(a - 1) * RW + (b - 1);
    } else {
      gp = Q0 - RW - 1 + u;
    }
    psrc[i] = p.x + ((long)pl * p.xps + gp) * 8;
    pdst[i] = lds_addr(Pb) + (unsigned)(pl * PPX + pp * 64) * 16u;
  }
  const bf16_t* wsrc[WPW]; unsigned wdst[WPW];
  const bf16_t* wtile = p.wp + (long)ct * nchunks * 9 * WB;
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int q = (wave + NW * i) % WPS;
    wsrc[i] = wtile + q * 512 + lane * 8;
    wdst[i] = lds_addr(Wb) + (unsigned)q * 1024u;
  }
  const long chunk_stride = 4 * p.xps * 8;        // elements between 32-channel chunks of the input
  const int S = nchunks * 9;
  auto issue_w = [&](int s) {                     // stage of step s -> ring slot s % NB (steps past the end repeat the last)
    const int sc = s < S ? s : S - 1;
    const unsigned slot_off = (unsigned)(s % NB) * (WB * 2u);
#pragma unroll
    for (int i = 0; i < WPW; ++i) glds16(wsrc[i] + (long)sc * WB, wdst[i] + slot_off);
  };
  auto issue_p = [&](int c) {                     // patch of chunk c -> buffer c & 1
    const int cc = c < nchunks ? c : nchunks - 1;
    const unsigned buf_off = (unsigned)(c & 1) * (PB * 2u);
#pragma unroll
    for (int i = 0; i < PPW; ++i) glds16(psrc[i] + cc * chunk_stride, pdst[i] + buf_off);
  };

  // ---- fragment addresses (bytes inside a stage)
  unsigned pbo[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int k = wpx * (BM / WP) + j * 16 + r;
    const int idx = TWO_D ? (k / TC + 1) * LP + (k % TC) + 1 : k + RW + 1;
    pbo[j] = (unsigned)(h * PPX + idx) * 16u;
  }
  const unsigned wbo = (unsigned)(h * BN + wco * (BN / WC) + r) * 16u;

  // accumulators start at the bias: D[cout][pixel], register x of a lane is cout 4*h + x of the 16-channel tile
  f32x4 acc[TCW][TPW];
#pragma unroll
  for (int i = 0; i < TCW; ++i) {
    f32x4 b0;
#pragma unroll
    for (int x = 0; x < 4; ++x) b0[x] = p.bias ? p.bias[ct * BN + wco * (BN / WC) + i * 16 + 4 * h + x] : 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[i][j] = b0;
  }

  issue_p(0);
#pragma unroll
  for (int d = 0; d < D; ++d) issue_w(d);
  wait_vm<(D - 1) * WPW>();
  __syncthreads();

  if (PP) {
    const bool late = wave >= NW / 2;                   // second-dispatched half: one barrier behind
    if (late) __builtin_amdgcn_s_barrier();
    for (int c = 0; c < nchunks; ++c) {
      const char* pb = reinterpret_cast<const char*>(Pb) + (c & 1) * (PB * 2);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int s = c * 9 + t;
        // ---- load phase
        issue_w(s + D);
        if (t == 0) issue_p(c + 1);
        const char* wb = reinterpret_cast<const char*>(Wb) + (s % NB) * (WB * 2);
        const int toff = ((t / 3) - 1) * TS + (t % 3) - 1;
        bf16x8 a[TCW], b[TPW];
#pragma unroll
        for (int i = 0; i < TCW; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wb + wbo + (i * 16) * 16);
#pragma unroll
        for (int j = 0; j < TPW; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + pbo[j] + toff * 16);
        if (t < D) wait_vm<(D - 1) * WPW + PPW>(); else wait_vm<(D - 1) * WPW>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute phase
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TCW; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!late) __builtin_amdgcn_s_barrier();
    wait_vm<0>();
  } else {
  for (int c = 0; c < nchunks; ++c) {
    const char* pb = reinterpret_cast<const char*>(Pb) + (c & 1) * (PB * 2);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int s = c * 9 + t;
      if (ABL != 3) {
        issue_w(s + D);
        if (t == 0) issue_p(c + 1);
      }
      const char* wb = reinterpret_cast<const char*>(Wb) + (s % NB) * (WB * 2);
      constexpr int dummy = 0; (void)dummy;
      const int toff = ((t / 3) - 1) * TS + (t % 3) - 1;
      {
        bf16x8 a[TCW], b[TPW];
#pragma unroll
        for (int i = 0; i < TCW; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wb + wbo + (i * 16) * 16);
#pragma unroll
        for (int j = 0; j < TPW; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pb + pbo[j] + toff * 16);
#pragma unroll
        for (int i = 0; i < TCW; ++i)
#pragma unroll
          for (int j = 0; j < TPW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      // stage s+1 (and, from the fourth step of a chunk on, the next chunk's patch) must have landed; the younger DMAs
      // (D-1 weight stages, plus the patch while it may still fly) stay in flight across the barrier
      if (ABL != 1 && ABL != 3) {
        if (t < D) wait_vm<(D - 1) * WPW + PPW>(); else wait_vm<(D - 1) * WPW>();
      }
      if (ABL != 1) __syncthreads();
    }
  }
  wait_vm<0>();
  }

  // ---- epilogue: ReLU / mask, zero at pads, bf16; a lane holds channels 4h .. 4h+3 of its pixel: 8-B stores
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int k = wpx * (BM / WP) + j * 16 + r;
    const long P = TWO_D ? Q0 + (long)(k / TC) * RW + (k % TC) : Q0 + k;
    if (P >= p.ptot) continue;
    const bool real = pf_real<RW>(P, p.H);
    bf16x4 mk[TCW];
    if (p.mask) {
#pragma unroll
      for (int i = 0; i < TCW; ++i) {
        const int co = ct * BN + wco * (BN / WC) + i * 16 + 4 * h;
        mk[i] = *reinterpret_cast<const bf16x4*>(p.mask + ((long)(co >> 3) * p.mps + P) * 8 + (co & 7));
      }
    }
#pragma unroll
    for (int i = 0; i < TCW; ++i) {
      const int co = ct * BN + wco * (BN / WC) + i * 16 + 4 * h;
      const long o = ((long)(co >> 3) * p.yps + P) * 8 + (co & 7);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)mk[i][e] > 0.f ? v[e] : 0.f;
      }
      bf16x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = (bf16_t)(real ? v[e] : 0.f);
      *reinterpret_cast<bf16x4*>(p.y + o) = out;
    }
  }
  zero_guard_slice<BN, NT>(p, ct, pt, tid);
  if (TWO_D && ctile == 0) {   // column 0 (the zero pad) of this tile's rows: no tile computes it
    const long row0 = Q0 / RW;
    for (int e = tid; e < TR * (BN / 8); e += NT) {
      const int i = e % TR, cb = e / TR;
      const long P = (row0 + i) * RW;
      if (P < p.ptot) {
        bf16x8 z;
#pragma unroll
        for (int q = 0; q < 8; ++q) z[q] = (bf16_t)0.f;
        *reinterpret_cast<bf16x8*>(p.y + ((long)(ct * (BN / 8) + cb) * p.yps + P) * 8) = z;
      }
    }
  }
}

// packed bf16 weights from the fp32 parameter w [Cout][Cin][3][3]:
//   wp[ct][chunk][tap][pl][m][e],  output channel o = ct*BN + m, reduction channel c = chunk*32 + pl*8 + e
//   forward  (transposed = 0): o = cout, c = cin, value w[o][c][tap]
//   dgrad    (transposed = 1): o = cin,  c = cout, value w[c][o][8 - tap]
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int M, int C, int Cin,
                                         int BN, int transposed) {
  const long total = (long)M * C * 9;
  const int nchunks = C / 32;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int e = (int)(idx & 7);
    long q = idx >> 3;
    const int m = (int)(q % BN); q /= BN;
    const int pl = (int)(q & 3); q >>= 2;
    const int tap = (int)(q % 9); q /= 9;
    const int chunk = (int)(q % nchunks);
    const int ct = (int)(q / nchunks);
    const int o = ct * BN + m, c = chunk * 32 + pl * 8 + e;
    const float v = transposed ? w[((long)c * Cin + o) * 9 + 8 - tap] : w[((long)o * Cin + c) * 9 + tap];
    wp[idx] = (bf16_t)v;
  }
}

// the same for up to 12 layers in one launch (all packs of a forward or a backward pass): blockIdx.y = layer
struct PackSet { const float* w[12]; bf16_t* wp[12]; int M[12], C[12], Cin[12], BN[12]; int transposed, n; };
// One workgroup = 16 output channels x 32 reduction channels x 9 taps, staged through LDS: the fp32 parameter is read in
// runs of 288 (forward) / 144 (transposed) consecutive floats and the packed image is written in runs of 256 B (16 m x 8 e).
// (The element-per-thread form above reads with a stride of 36 B per lane; all 24 packs of a step took 0.27 ms that way.)
__global__ __launch_bounds__(256) void pack_weights_bf16_multi_kernel(PackSet ps) {
  __shared__ float T[9 * 32 * 17];                      // T[tap][c_l][o_l], o_l padded to 17
  const int l = blockIdx.y;
  const int M = ps.M[l], C = ps.C[l], Cin = ps.Cin[l], BN = ps.BN[l];
  const int nchunks = C / 32;
  const int tile = blockIdx.x;
  if (tile >= (M / 16) * nchunks) return;
  const int ot = tile / nchunks, chunk = tile - ot * nchunks;
  const int o0 = ot * 16, c0 = chunk * 32;
  const float* __restrict__ w = ps.w[l];
  const int tid = threadIdx.x;
  if (!ps.transposed) {                                  // value(o, c, tap) = w[o][c][tap], o-rows of 288 floats
    for (int idx = tid; idx < 16 * 288; idx += 256) {
      const int o_l = idx / 288, rem = idx - o_l * 288;
      const int c_l = rem / 9, tap = rem - c_l * 9;
      T[(tap * 32 + c_l) * 17 + o_l] = w[((long)(o0 + o_l) * Cin + c0) * 9 + rem];
    }
  } else {                                               // value(o, c, tap) = w[c][o][8 - tap], c-rows of 144 floats
    for (int idx = tid; idx < 32 * 144; idx += 256) {
      const int c_l = idx / 144, rem = idx - c_l * 144;
      const int o_l = rem / 9, t = rem - o_l * 9;
      T[((8 - t) * 32 + c_l) * 17 + o_l] = w[((long)(c0 + c_l) * Cin + o0) * 9 + rem];
    }
  }
  __syncthreads();
  const int ct = o0 / BN, m0 = o0 - ct * BN;
  bf16_t* __restrict__ wp = ps.wp[l];
  for (int j = tid; j < 9 * 512; j += 256) {
    const int tap = j >> 9, r = j & 511;
    const int pl = r >> 7, m_l = (r & 127) >> 3, e = r & 7;
    const long dst = (((((long)ct * nchunks + chunk) * 9 + tap) * 4 + pl) * BN + m0 + m_l) * 8 + e;
    wp[dst] = (bf16_t)T[(tap * 32 + pl * 8 + e) * 17 + m_l];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient
struct WgradB16Params {
  const bf16_t* dy; long dps;    // [Cout/8][ps][8], zero at pads and in the guards
  const bf16_t* x; long xps;     // [Cin/8][ps][8]
  float* slab;                   // [splits][Cout][Cin][9]: the parameter's own order, summed by the linear reduce
  float* bslab;                  // [splits][Cout] bias-gradient partial sums (written by the ci-tile-0 workgroups) or null
  int Cin, Cout;
  long nseg;                     // pixel segments
  int segs_per_split, splits, ntiles, ncit;
};

// segment: 1-D (TR == 0): TC consecutive flat pixels; 2-D: TR rows x TC real columns.  Workgroup tile = 32*WCO output
// channels x 32*WCI input channels x 9 taps; KSW wave groups split the k-steps of every segment between them.
// (The first version of this kernel used v_mfma_f32_32x32x16_bf16 with 16-pixel k-steps; the 16x16x32 form below replaced it
// after the A/B in profiles/r02_k_m16_ab.txt: +2 %.)
// The weight gradient on v_mfma_f32_16x16x32_bf16: a k-step is 32 pixels, a wave's 32 x 32 channel tile is 2 x 2 MFMA
// tiles per tap (same LDS bytes and MFMA cycles as the 32x32x16 form, higher sustained clock).
template <int W, int TR, int TC, int WCO, int WCI, int KSW>
__global__ __launch_bounds__(64 * WCO * WCI * KSW) void wgrad_bf16_m16_kernel(WgradB16Params p) {
  constexpr int NW = WCO * WCI * KSW, NT = 64 * NW;
  constexpr bool TWO_D = TR > 0;
  constexpr int KP = TWO_D ? TR * TC : TC;        // pixels per segment
  constexpr int RW = W + 1;
  constexpr int LP = TC + 2;
  constexpr int TS = TWO_D ? LP : RW;
  constexpr int PATCH = TWO_D ? (TR + 2) * LP : KP + 2 * RW + 2;
  constexpr int PPP = (PATCH + 63) / 64;
  constexpr int PPX = PPP * 64 + 4;               // +4 pixels: plane stride off the 256-B bank period (tr reads)
  constexpr int KPX = KP + 4;
  constexpr int GPL = 4 * WCO, XPL = 4 * WCI;     // dY planes / x planes of the tile
  constexpr int GPC = GPL * (KP / 64), XPC = XPL * PPP;
  constexpr int NPC = GPC + XPC;                  // DMA pieces per segment
  constexpr int PCW = (NPC + NW - 1) / NW;
  constexpr int GB = GPL * KPX * 8, XB = XPL * PPX * 8;   // elements per stage
  constexpr int KSTEPS = KP / 32, KSL = KSTEPS / KSW;
  static_assert(KP % 64 == 0 && KSTEPS % KSW == 0 && (!TWO_D || (W % TC == 0 && TC % 16 == 0)), "segment shape");
  constexpr int RED = KSW > 1 ? (WCO * WCI * 9 * 16 * 64 + WCO * WCI * 32) * 2 : 0;   // fp32 exchange of the wave groups, in bf16 units
  constexpr int SMEM = 2 * (GB + XB) > RED ? 2 * (GB + XB) : RED;
  __shared__ __attribute__((aligned(1024))) bf16_t smem[SMEM];
  bf16_t* const Gs = smem;                         // [2][GB]
  bf16_t* const Xs = smem + 2 * GB;                // [2][XB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ksw = wave / (WCO * WCI), wt = wave % (WCO * WCI);
  const int wco = wt / WCI, wci = wt % WCI;
  // workgroups of one split read the same pixels: keep them on one XCD (splits is a multiple of 8 or the tail idles)
  const int xcd = blockIdx.x & 7;
  const long qq = blockIdx.x >> 3;
  const int tile = (int)(qq % p.ntiles);
  const int split = (int)(qq / p.ntiles) * 8 + xcd;
  if (split >= p.splits) return;
  const int cot = tile / p.ncit, cit = tile % p.ncit;
  const long sbeg = (long)split * p.segs_per_split;
  const long send = min(p.nseg, sbeg + p.segs_per_split);

  // ---- DMA plan: piece q of a segment; q < GPC: dY plane q / (KP/64), 64 pixels; else x plane, patch piece
  const bf16_t* src0[PCW]; unsigned dst0[PCW]; int isx[PCW];
#pragma unroll
  for (int i = 0; i < PCW; ++i) {
    const int q = (wave + NW * i) % NPC;
    if (q < GPC) {
      const int pl = q / (KP / 64), pp = q % (KP / 64);
      const int k = pp * 64 + lane;
      const long gp = TWO_D ? (long)(k / TC) * RW + (k % TC) : k;
      src0[i] = p.dy + ((long)(cot * GPL + pl) * p.dps + gp) * 8;
      dst0[i] = lds_addr(Gs) + (unsigned)(pl * KPX + pp * 64) * 16u;
      isx[i] = 0;
    } else {
      const int q2 = q - GPC;
      const int pl = q2 / PPP, pp = q2 % PPP;
      int u = pp * 64 + lane;
      long gp;
      if (TWO_D) {
        if (u > PATCH - 1) u = PATCH - 1;
        const int a = u / LP, b = u - a * LP;
        gp = (long)(a - 1) * RW + (b - 1);
      } else {
        gp = u - RW - 1;
      }
      src0[i] = p.x + ((long)(cit * XPL + pl) * p.xps + gp) * 8;
      dst0[i] = lds_addr(Xs) + (unsigned)(pl * PPX + pp * 64) * 16u;
      isx[i] = 1;
    }
  }
  auto seg_origin = [&](long seg) -> long {
    if (TWO_D) {
      constexpr int CT = W / TC;
      const long band = seg / CT;
      return band * TR * RW + 1 + (seg - band * CT) * TC;
    }
    return seg * KP;
  };
  auto issue = [&](long seg, int buf) {
    const long q0 = seg_origin(seg) * 8;
#pragma unroll
    for (int i = 0; i < PCW; ++i)
      glds16(src0[i] + q0, dst0[i] + (unsigned)buf * (isx[i] ? XB * 2u : GB * 2u));
  };

  // ---- fragment addresses.  16-lane group g handles pixels 8g .. 8g+7 of the 32-pixel k-step for 16 channels (two
  // planes); inside a group lane 4q+p supplies the address of k-row q, channels 4p..4p+3.  Wave group ksw works on the
  // k-steps kk*KSW + ksw: its offset is part of the lane base, so every k-step / tap offset below is an immediate.
  const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int kg = 8 * g + q4;                                          // + 4 for the second read of a k-step
  const int kx = ((TWO_D && TC == 16) ? (g >> 1) * LP + 8 * (g & 1) : 8 * g) + q4;
  const int kadv_g = ksw * 32;
  const int kadv_x = ksw * (TWO_D ? (TC == 16 ? 2 * LP : LP) : 32);
  const unsigned abase = (unsigned)((wco * 4 + (p4 >> 1)) * KPX + kg + kadv_g) * 16u + (p4 & 1) * 8u;   // + 2 planes per tile
  const unsigned bbase = (unsigned)((wci * 4 + (p4 >> 1)) * PPX + kx + kadv_x) * 16u + (p4 & 1) * 8u;

  f32x4 acc[9][2][2];                                                  // [tap][co tile][ci tile]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[t][u >> 1][u & 1][x] = 0.f;
  // bias gradient: an A fragment of a lane is 8 pixels of ONE output channel (row l & 15 of its tile) - the waves
  // with input-channel position 0 of the ci-tile-0 workgroups add them up on the side (16 VALU per 9 MFMAs)
  const bool do_bias = p.bslab != nullptr && cit == 0 && wci == 0;   // wave-uniform
  float bsum[2] = {0.f, 0.f};

  if (sbeg < send) issue(sbeg, 0);
  wait_vm<0>();
  __syncthreads();
  for (long seg = sbeg; seg < send; ++seg) {
    const int cur = (int)(seg - sbeg) & 1;
    issue(seg + 1 < send ? seg + 1 : seg, cur ^ 1);      // the last segment re-loads itself: uniform DMA count
    const char* gs = reinterpret_cast<const char*>(Gs) + cur * (GB * 2) + abase;
    const char* xs = reinterpret_cast<const char*>(Xs) + cur * (XB * 2) + bbase;
#pragma unroll
    for (int kk = 0; kk < KSL; ++kk) {
      const int ko = kk * KSW * 32;                      // first pixel of wave group 0's k-step in the dY image
      const int xo = TWO_D ? (ko / TC + 1) * LP + (ko % TC) + 1 : ko + RW + 1;   // ... and in the x patch (centre tap)
      bf16x8 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(lds_ptr_t)(gs + (i * 2 * KPX + ko) * 16));
        const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(lds_ptr_t)(gs + (i * 2 * KPX + ko + 4) * 16));
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[i][e] = a0[e]; a[i][4 + e] = a1[e]; }
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float s4 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) s4 += (float)a[i][e];
          bsum[i] += s4;
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) - 1) * TS + (t % 3) - 1;
        bf16x8 b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(lds_ptr_t)(xs + (i * 2 * PPX + xo + toff) * 16));
          const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(lds_ptr_t)(xs + (i * 2 * PPX + xo + toff + 4) * 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) { b[i][e] = b0[e]; b[i][4 + e] = b1[e]; }
        }
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
          for (int ib = 0; ib < 2; ++ib)
            acc[t][ia][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ia], b[ib], acc[t][ia][ib], 0, 0, 0);
      }
    }
    wait_vm<0>();
    __syncthreads();
  }

  // ---- combine the KSW wave groups through LDS, then store the slab tile: D[co][ci], lanes run along ci
#pragma unroll
  for (int i = 0; i < 2; ++i) {                          // the four k-quarters of a channel
    bsum[i] += __shfl_xor(bsum[i], 16, 64);
    bsum[i] += __shfl_xor(bsum[i], 32, 64);
  }
  if (KSW > 1) {
    float* red = reinterpret_cast<float*>(smem);       // [WCO*WCI][144][64] floats = 36 KB per wave, then [WCO*WCI][32]
    if (ksw == 1) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int x = 0; x < 4; ++x) red[((wt * 9 + t) * 16 + u * 4 + x) * 64 + lane] = acc[t][u >> 1][u & 1][x];
      if (lane < 16) {
        red[WCO * WCI * 9 * 16 * 64 + wt * 32 + lane] = bsum[0];
        red[WCO * WCI * 9 * 16 * 64 + wt * 32 + 16 + lane] = bsum[1];
      }
    }
    __syncthreads();
    if (ksw == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int x = 0; x < 4; ++x) acc[t][u >> 1][u & 1][x] += red[((wt * 9 + t) * 16 + u * 4 + x) * 64 + lane];
      bsum[0] += red[WCO * WCI * 9 * 16 * 64 + wt * 32 + (lane & 15)];
      bsum[1] += red[WCO * WCI * 9 * 16 * 64 + wt * 32 + 16 + (lane & 15)];
    }
  }
  if (ksw == 0) {
    const int r = lane & 15;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = cit * 32 * WCI + wci * 32 + (u & 1) * 16 + r;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int co = cot * 32 * WCO + wco * 32 + (u >> 1) * 16 + 4 * g + x;
        float* o = p.slab + (((long)split * p.Cout + co) * p.Cin + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t] = acc[t][u >> 1][u & 1][x];
      }
    }
    if (do_bias && lane < 16) {
      p.bslab[(long)split * p.Cout + cot * 32 * WCO + wco * 32 + lane] = bsum[0];
      p.bslab[(long)split * p.Cout + cot * 32 * WCO + wco * 32 + 16 + lane] = bsum[1];
    }
  }
}

// dW (+)= sum over splits of slab[split][...] in the parameter's own order (float4), db likewise from bslab
__global__ void wgrad_reduce_linear_kernel(const float* __restrict__ slab, const float* __restrict__ bslab, int splits,
                                           long per, int Cout, float* __restrict__ dw, float* __restrict__ db,
                                           int accumulate) {
  const long nv = per / 4;                               // per = 9 * Cout * Cin is a multiple of 4 (channels of 64)
  const float4* s4 = reinterpret_cast<const float4*>(slab);
  float4* d4 = reinterpret_cast<float4*>(dw);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv + (bslab ? Cout : 0); i += (long)gridDim.x * blockDim.x) {
    if (i < nv) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;   // two interleaved chains, fixed order
      int q = 0;
      for (; q + 1 < splits; q += 2) {
        const float4 u = s4[(long)q * nv + i], v = s4[(long)(q + 1) * nv + i];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
      }
      if (q < splits) { const float4 u = s4[(long)q * nv + i]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
      float4 r = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
      if (accumulate) { const float4 o = d4[i]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
      d4[i] = r;
    } else {
      const int co = (int)(i - nv);
      float v = 0.f;
      for (int q = 0; q < splits; ++q) v += bslab[(long)q * Cout + co];
      db[co] = accumulate ? db[co] + v : v;
    }
  }
}

// The same for layers with many splits (the 64- and 128-channel layers: one or two output tiles, so 128-256 splits): a
// thread of the kernel above walks all of them one dependent load after the other (200 us for the 36 864 weights of
// conv1_2, at the very end of the backward pass).  Here 1024 threads = 64 float4 columns x 16 split groups; a group sums the
// splits q = g, g + 16, ... in order, the groups are combined through LDS in a fixed order.
__global__ __launch_bounds__(1024) void wgrad_reduce_wide_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                                 int splits, long per, int Cout, float* __restrict__ dw,
                                                                 float* __restrict__ db, int accumulate) {
  __shared__ float4 red[16][64];
  const long nv = per / 4;
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + c;
  const long nvb = nv + (bslab ? (Cout + 3) / 4 : 0);     // bias columns follow, four channels per "column"
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < nv) {
    const float4* s4 = reinterpret_cast<const float4*>(slab);
    for (int q = g; q < splits; q += 16) { const float4 u = s4[(long)q * nv + i]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
  } else if (i < nvb) {
    const int co = (int)(i - nv) * 4;
    for (int q = g; q < splits; q += 16) {
      const float* b = bslab + (long)q * Cout + co;
      a.x += b[0]; if (co + 1 < Cout) a.y += b[1]; if (co + 2 < Cout) a.z += b[2]; if (co + 3 < Cout) a.w += b[3];
    }
  }
  red[g][c] = a;
  __syncthreads();
  if (g == 0 && i < nvb) {
    float4 r = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 u = red[k][c]; r.x += u.x; r.y += u.y; r.z += u.z; r.w += u.w; }
    if (i < nv) {
      float4* d4 = reinterpret_cast<float4*>(dw);
      if (accumulate) { const float4 o = d4[i]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
      d4[i] = r;
    } else {
      const int co = (int)(i - nv) * 4;
      const float v[4] = {r.x, r.y, r.z, r.w};
      for (int e = 0; e < 4 && co + e < Cout; ++e) db[co + e] = accumulate ? db[co + e] + v[e] : v[e];
    }
  }
}

// bias gradient partial sums: bslab[chunk][co] = sum over the chunk's pixels of dY[P][co]
__global__ __launch_bounds__(256) void bias_grad_bf16_kernel(const bf16_t* __restrict__ dy, long dps, long ptot,
                                                            int Cout, int nchunk, float* __restrict__ bslab) {
  const int cb = blockIdx.x, chunk = blockIdx.y;
  const long per = (ptot + nchunk - 1) / nchunk;
  const long beg = chunk * per, end = min(ptot, beg + per);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  const bf16_t* base = dy + (long)cb * dps * 8;
  for (long P = beg + threadIdx.x; P < end; P += 256) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + P * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
  }
  __shared__ float red[4][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = wave_sum(s[e]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x >> 6][e] = s[e];
  }
  __syncthreads();
  if (threadIdx.x < 8) bslab[(long)chunk * Cout + cb * 8 + threadIdx.x] =
      (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------------------------
// layout conversion, pooling, guards
// fp32 NCHW -> bf16 CB8-PF (zero pads written; channels past C are zero)
__global__ void nchw_to_cb8_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long yps, int C, int H, int W,
                                   long ptot, long lead) {
  const int RW = W + 1;
  const int cb = blockIdx.y;
  // Q runs over the whole plane [0, yps): pixel P = Q - lead; the guards (P < 0, P >= ptot) are written as zeros here
  for (long Q = blockIdx.x * (long)blockDim.x + threadIdx.x; Q < yps; Q += (long)gridDim.x * blockDim.x) {
    const long P = Q - lead;
    const bool inside = P >= 0 && P < ptot;
    const long row = inside ? P / RW : 0; const int col = inside ? (int)(P - row * RW) : 0;
    const int rr = (int)(row % (H + 1));
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
    if (col != 0 && rr != 0) {
      const long n = row / (H + 1);
      const long o = (n * C * H + (rr - 1)) * (long)W + (col - 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cb * 8 + e;
        if (c < C) v[e] = (bf16_t)x[o + (long)c * H * W];
      }
    }
    *reinterpret_cast<bf16x8*>(y + ((long)cb * yps + P) * 8) = v;
  }
}

// bf16 CB8-PF -> fp32 NCHW (real pixels only)
__global__ void cb8_to_nchw_kernel(const bf16_t* __restrict__ x, long xps, float* __restrict__ y, int N, int C, int H,
                                   int W) {
  const int RW = W + 1;
  const int cb = blockIdx.y;
  const long total = (long)N * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % W);
    const long t = i / W;
    const int yy = (int)(t % H);
    const long n = t / H;
    const long P = (n * (H + 1) + yy + 1) * RW + xx + 1;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)cb * xps + P) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cb * 8 + e;
      if (c < C) y[((n * C + c) * H + yy) * (long)W + xx] = (float)v[e];
    }
  }
}

__global__ void zero_guards_kernel(bf16_t* __restrict__ base, long ps, long lead, long ptot) {
  // base = start of plane 0 (not pixel 0): zero [0, lead) and [lead + ptot, ps) of every plane, 16 B per thread
  const int pl = blockIdx.y;
  const long tail0 = lead + ptot, n = lead + (ps - tail0);
  uint4 z = make_uint4(0, 0, 0, 0);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long P = i < lead ? i : tail0 + (i - lead);
    *reinterpret_cast<uint4*>(base + ((long)pl * ps + P) * 8) = z;
  }
}

// the same for up to 20 tensors in one launch (the activation arena of a forward pass): blockIdx.y = tensor
struct GuardSet { bf16_t* base[20]; long ps[20], lead[20], ptot[20]; int planes[20]; int n; };
__global__ void zero_guards_multi_kernel(GuardSet gs) {
  const int t = blockIdx.y;
  const long ps = gs.ps[t], lead = gs.lead[t], tail0 = lead + gs.ptot[t];
  const long per = lead + (ps - tail0), n = per * gs.planes[t];
  uint4 z = make_uint4(0, 0, 0, 0);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long pl = i / per, q = i - pl * per;
    const long P = q < lead ? q : tail0 + (q - lead);
    *reinterpret_cast<uint4*>(gs.base[t] + (pl * ps + P) * 8) = z;
  }
}

__device__ __forceinline__ bf16x8 max8(bf16x8 a, bf16x8 b) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (float)a[e] >= (float)b[e] ? a[e] : b[e];
  return r;
}

// 2x2/2 max pooling on CB8-PF maps; writes every output pixel incl. the zero pads
__global__ void maxpool2_bf16_fwd_kernel(const bf16_t* __restrict__ x, long xps, bf16_t* __restrict__ y, long yps, int H,
                                         int W, long ptot_out) {
  const int Ho = H / 2, Wo = W / 2, RWo = Wo + 1, RWi = W + 1;
  const int cb = blockIdx.y;
  for (long P = blockIdx.x * (long)blockDim.x + threadIdx.x; P < ptot_out; P += (long)gridDim.x * blockDim.x) {
    const long row = P / RWo; const int col = (int)(P - row * RWo);
    const int rr = (int)(row % (Ho + 1));
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
    if (col != 0 && rr != 0) {
      const long n = row / (Ho + 1);
      const long Pi = (n * (H + 1) + 2 * (rr - 1) + 1) * RWi + 2 * (col - 1) + 1;
      const bf16_t* s = x + ((long)cb * xps + Pi) * 8;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(s), b = *reinterpret_cast<const bf16x8*>(s + 8);
      const bf16x8 c = *reinterpret_cast<const bf16x8*>(s + (long)RWi * 8), d = *reinterpret_cast<const bf16x8*>(s + (long)RWi * 8 + 8);
      v = max8(max8(a, b), max8(c, d));
    }
    *reinterpret_cast<bf16x8*>(y + ((long)cb * yps + P) * 8) = v;
  }
}

// gx = pool backward routed to the FIRST maximum of each window (order (0,0),(0,1),(1,0),(1,1), like the fp32 kernel),
// zero where that maximum is not > 0 (ReLU of the activation that fed the pool); one thread per INPUT pixel, pads zero
// Backward of the 2x2 max-pool with the ReLU mask of the layer before it, one thread per pooled pixel: the window's four
// inputs are read ONCE (64 B), the pooled gradient once (16 B), and the four gradient pixels leave as two 32-B pieces.  (The
// first version ran one thread per INPUT pixel: every thread of a window fetched the whole window again and redid its argmax -
// 80 B through the texture path per 16 B written, 2.0 TB/s over the five pools of a batch-64 step.)  The guards of the gx
// plane are covered by the same index space: a guard pixel of the POOLED map owns the guard pixels of the input map that sit
// where its window would be (column 0 of its two rows; the zero row above an image; both for the corner), and the indices
// past the pooled map zero the lead / tail of the plane.
__global__ __launch_bounds__(256) void maxpool2_bf16_bwd_relu_win_kernel(const bf16_t* __restrict__ x, long xps,
                                                                         const bf16_t* __restrict__ gy, long gps,
                                                                         bf16_t* __restrict__ gx, long gxps, int H, int W,
                                                                         unsigned ptot_out, long ptot_in, long lead) {
  const unsigned Ho = H / 2, Wo = W / 2, RWo = Wo + 1, RWi = W + 1;
  const int cb = blockIdx.y;
  const unsigned extra = (unsigned)(gxps - ptot_in);        // lead + tail pixels of the plane
  bf16x8 zero;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero[e] = (bf16_t)0.f;
  bf16_t* gplane = gx + (long)cb * gxps * 8;                 // gx points at pixel 0; the plane starts `lead` pixels before it
  for (unsigned P = blockIdx.x * blockDim.x + threadIdx.x; P < ptot_out + extra; P += gridDim.x * blockDim.x) {
    if (P >= ptot_out) {                                     // plane guards: [-lead, 0) and [ptot_in, gxps - lead)
      const long q = P - ptot_out;
      const long pix = q < lead ? q - lead : ptot_in + (q - lead);
      *reinterpret_cast<bf16x8*>(gplane + pix * 8) = zero;
      continue;
    }
    const unsigned row = P / RWo, col = P - row * RWo;
    const unsigned n = row / (Ho + 1), rr = row - n * (Ho + 1);
    if (rr == 0) {                                           // the zero row above image n (or below the last image)
      bf16_t* d = gplane + ((long)n * (H + 1) * RWi) * 8;
      if (col == 0) *reinterpret_cast<bf16x8*>(d) = zero;
      else { *reinterpret_cast<bf16x8*>(d + (2 * col - 1) * 8) = zero; *reinterpret_cast<bf16x8*>(d + (2 * col) * 8) = zero; }
      continue;
    }
    const long r1 = (long)n * (H + 1) + 2 * (rr - 1) + 1;    // first of the window's two input rows
    if (col == 0) {
      *reinterpret_cast<bf16x8*>(gplane + (r1 * RWi) * 8) = zero;
      *reinterpret_cast<bf16x8*>(gplane + ((r1 + 1) * RWi) * 8) = zero;
      continue;
    }
    const long Pw = r1 * RWi + 2 * (col - 1) + 1;
    const bf16_t* s = x + ((long)cb * xps + Pw) * 8;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(s), b = *reinterpret_cast<const bf16x8*>(s + 8);
    const bf16x8 c = *reinterpret_cast<const bf16x8*>(s + (long)RWi * 8), d = *reinterpret_cast<const bf16x8*>(s + (long)RWi * 8 + 8);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(gy + ((long)cb * gps + P) * 8);
    bf16x8 o[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int arg = 0; float m = (float)a[e];                    // first maximum wins, as in the forward kernel and in ATen
      if ((float)b[e] > m) { m = (float)b[e]; arg = 1; }
      if ((float)c[e] > m) { m = (float)c[e]; arg = 2; }
      if ((float)d[e] > m) { m = (float)d[e]; arg = 3; }
      const bool live = m > 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q][e] = (live && arg == q) ? g[e] : (bf16_t)0.f;
    }
    bf16_t* dst = gplane + Pw * 8;
    *reinterpret_cast<bf16x8*>(dst) = o[0]; *reinterpret_cast<bf16x8*>(dst + 8) = o[1];
    *reinterpret_cast<bf16x8*>(dst + (long)RWi * 8) = o[2]; *reinterpret_cast<bf16x8*>(dst + (long)RWi * 8 + 8) = o[3];
  }
}

__global__ void maxpool2_bf16_bwd_relu_kernel(const bf16_t* __restrict__ x, long xps, const bf16_t* __restrict__ gy,
                                              long gps, bf16_t* __restrict__ gx, long gxps, int H, int W, long ptot_in,
                                              long lead) {
  const int Ho = H / 2, Wo = W / 2, RWo = Wo + 1, RWi = W + 1;
  const int cb = blockIdx.y;
  // Q runs over the whole plane of gx: pixel P = Q - lead; its guards are written as zeros here
  for (long Q = blockIdx.x * (long)blockDim.x + threadIdx.x; Q < gxps; Q += (long)gridDim.x * blockDim.x) {
    const long P = Q - lead;
    const bool inside = P >= 0 && P < ptot_in;
    const long row = inside ? P / RWi : 0; const int col = inside ? (int)(P - row * RWi) : 0;
    const int rr = (int)(row % (H + 1));
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (bf16_t)0.f;
    if (col != 0 && rr != 0) {
      const long n = row / (H + 1);
      const int yy = rr - 1, xx = col - 1;
      const int yo = yy >> 1, xo = xx >> 1, me = (yy & 1) * 2 + (xx & 1);
      const long Pw = (n * (H + 1) + 2 * yo + 1) * RWi + 2 * xo + 1;
      const bf16_t* s = x + ((long)cb * xps + Pw) * 8;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(s), b = *reinterpret_cast<const bf16x8*>(s + 8);
      const bf16x8 c = *reinterpret_cast<const bf16x8*>(s + (long)RWi * 8), d = *reinterpret_cast<const bf16x8*>(s + (long)RWi * 8 + 8);
      const long Po = (n * (Ho + 1) + yo + 1) * RWo + xo + 1;
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(gy + ((long)cb * gps + Po) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int arg = 0; float m = (float)a[e];
        if ((float)b[e] > m) { m = (float)b[e]; arg = 1; }
        if ((float)c[e] > m) { m = (float)c[e]; arg = 2; }
        if ((float)d[e] > m) { m = (float)d[e]; arg = 3; }
        out[e] = (m > 0.f && arg == me) ? g[e] : (bf16_t)0.f;
      }
    }
    *reinterpret_cast<bf16x8*>(gx + ((long)cb * gxps + P) * 8) = out;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// first VGG layer (3 input channels, K = 27 padded to 32): fp32 NCHW image in, bf16 CB8-PF out.  No LDS, no barriers:
// every wave keeps the 64 x 32 weight fragments in registers and walks over 32-pixel row segments; the im2col fragment
// (16 values per lane) is gathered straight from the image (38 MB at batch 64: L2 / Infinity-Cache resident).
struct Conv1B16Params { const float* x; const float* w; const float* bias; bf16_t* y; long yps; int N, H, W; };

__global__ __launch_bounds__(256) void conv1_bf16_fwd_kernel(Conv1B16Params p) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const int H = p.H, W = p.W, RW = W + 1;
  const long HW = (long)H * W;
  bf16x8 wa[2][2];
  int doff[16], ddy[16], ddx[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = 16 * (q >> 3) + 8 * h + (q & 7);           // k = c * 9 + tap; k >= 27: zero weight, any valid address
    const int kk = k < 27 ? k : 0;
    const int c = kk / 9, tap = kk - c * 9;
    ddy[q] = tap / 3 - 1; ddx[q] = tap % 3 - 1;
    doff[q] = c * (int)HW + ddy[q] * W + ddx[q];
#pragma unroll
    for (int i = 0; i < 2; ++i) wa[i][q >> 3][q & 7] = (bf16_t)(k < 27 ? p.w[(i * 32 + r) * 27 + k] : 0.f);
  }
  f32x16 b0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int x = 0; x < 16; ++x) b0[i][x] = p.bias ? p.bias[i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h] : 0.f;
  const int xt_n = W / 32;
  const long ntiles = (long)p.N * H * xt_n;
  for (long t = wave; t < ntiles; t += nwaves) {
    const int xt = (int)(t % xt_n);
    const long ny = t / xt_n;
    const int y = (int)(ny % H);
    const long n = ny / H;
    const int x = xt * 32 + r;
    const float* px = p.x + n * 3 * HW + (long)y * W + x;
    const bool interior = y > 0 && y < H - 1 && xt > 0 && xt < xt_n - 1;      // wave-uniform
    bf16x8 bf[2];
    if (interior) {
#pragma unroll
      for (int q = 0; q < 16; ++q) bf[q >> 3][q & 7] = (bf16_t)px[doff[q]];
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int yy = y + ddy[q], xx = x + ddx[q];
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float v = px[ok ? doff[q] : 0];
        bf[q >> 3][q & 7] = (bf16_t)(ok ? v : 0.f);
      }
    }
    const long P = (n * (H + 1) + y + 1) * RW + x + 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 acc = b0[i];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][0], bf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][1], bf[1], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = i * 32 + 8 * g + 4 * h;
        bf16x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = (bf16_t)fmaxf(acc[4 * g + e], 0.f);
        *reinterpret_cast<bf16x4*>(p.y + ((long)(co >> 3) * p.yps + P) * 8 + (co & 7)) = out;
      }
    }
  }
}

// zero pads of a CB8-PF tensor whose real pixels are written by a kernel that skips the pads: column 0 of every row and
// the zero rows before / between / after the images
__global__ void pf_zero_pads_kernel(bf16_t* __restrict__ y, long yps, int H, int W, long rows) {
  const int RW = W + 1, cb = blockIdx.y;
  uint4 z = make_uint4(0, 0, 0, 0);
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    bf16_t* base = y + ((long)cb * yps + row * RW) * 8;
    if (row % (H + 1) == 0) {
      for (int c = threadIdx.x; c < RW; c += blockDim.x) *reinterpret_cast<uint4*>(base + (long)c * 8) = z;
    } else if (threadIdx.x == 0) {
      *reinterpret_cast<uint4*>(base) = z;
    }
  }
}

// first layer's weight gradient: dW[co][c*9+tap] = sum_P dY[P][co] * x[P + off(tap)][c] with dY in bf16 CB8-PF and the
// fp32 image.  M = 64 output channels (two accumulator tiles), N = 27 (one tile of 32 columns), K = pixels in steps of
// 16 along a row.  Both fragments are gathered straight from memory (each dY element exactly once in the whole launch, the
// image from L2): no LDS in the loop; the waves of a workgroup are summed through LDS at the end, workgroups write
// split-K slabs [split][9][64][3] for the shared reduce kernel.
struct Wgrad1B16Params { const bf16_t* dy; long dps; const float* x; float* slab; float* bslab; int N, H, W; };

__global__ __launch_bounds__(256) void conv1_bf16_wgrad_kernel(Wgrad1B16Params p) {
  __shared__ float red[3][2][16][64];
  __shared__ float bred[4][64];
  float bs[2] = {0.f, 0.f};
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
  const long wave = (long)blockIdx.x * 4 + wv, nwaves = (long)gridDim.x * 4;
  const int H = p.H, W = p.W, RW = W + 1;
  const long HW = (long)H * W;
  const int col = r < 27 ? r : 0;
  const int c = col / 9, tap = col - c * 9, dy_ = tap / 3 - 1, dx_ = tap % 3 - 1;
  const bool colok = r < 27;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int x = 0; x < 16; ++x) acc[i][x] = 0.f;
  const int ks_n = W / 16;
  const long nsteps = (long)p.N * H * ks_n;
  for (long t = wave; t < nsteps; t += nwaves) {
    const int kx = (int)(t % ks_n);
    const long ny = t / ks_n;
    const int y = (int)(ny % H);
    const long n = ny / H;
    const int x0 = kx * 16 + 8 * h;                              // this lane half's 8 pixels x0 .. x0+7 of row y
    const long P = (n * (H + 1) + y + 1) * RW + x0 + 1;
    // A[row = co][k = pixel]: 8 consecutive pixels of channel co = i*32 + r: 2-byte gathers at a 16-B stride
    bf16x8 a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int co = i * 32 + r;
      const bf16_t* g = p.dy + ((long)(co >> 3) * p.dps + P) * 8 + (co & 7);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[i][j] = g[j * 8];
      float s8 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s8 += (float)a[i][j];
      bs[i] += s8;                                               // bias gradient of channel i*32 + r on the side
    }
    // B[k = pixel][col = (c, tap)]: the image at (y + dy, x0 + j + dx), zero outside
    bf16x8 b;
    const int yy = y + dy_;
    const bool rowok = colok && yy >= 0 && yy < H;
    const float* px = p.x + (n * 3 + c) * HW + (long)(rowok ? yy : 0) * W;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int xx = x0 + j + dx_;
      const bool ok = rowok && xx >= 0 && xx < W;
      const float v = px[ok ? xx : 0];
      b[j] = (bf16_t)(ok ? v : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[i], 0, 0, 0);
  }
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) red[wv - 1][i][x][lane] = acc[i][x];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) bs[i] += __shfl_xor(bs[i], 32, 64);
  if (lane < 32) { bred[wv][lane] = bs[0]; bred[wv][32 + lane] = bs[1]; }
  __syncthreads();
  if (p.bslab && threadIdx.x < 64)
    p.bslab[(long)blockIdx.x * 64 + threadIdx.x] = (bred[0][threadIdx.x] + bred[1][threadIdx.x]) + (bred[2][threadIdx.x] + bred[3][threadIdx.x]);
  if (wv == 0 && colok) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const float v = acc[i][x] + ((red[0][i][x][lane] + red[1][i][x][lane]) + red[2][i][x][lane]);
        const int co = i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        p.slab[(((long)blockIdx.x * 9 + tap) * 64 + co) * 3 + c] = v;
      }
  }
}

inline int grid_for(long n, int cap) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

// 1 (default): v_mfma_f32_16x16x32_bf16 form, 5-10 % faster per layer on MI355X (profiles/r02_k_m16_ab.txt); 0: 32x32x16 form
const int g_b16_m16 = umpr_env_int("UMPR_B16_M16", 1);

// UMPR_B16_PP: ping-pong schedule in the 8-wave forward / dgrad kernels.  2 (default): where it measured faster - the
// 256-channel tiles (wave tile 128 x 64, 32 MFMAs per phase: +8-10 %, 512->512@28 1158 TF) and the 1-D 128-channel tiles
// (+3-5 %); the 2-D 128-channel tiles of the 112 x 112 maps (16 MFMAs per phase) lose 15 % to the second barrier.
// 1: everywhere, 0: nowhere (profiles/r02_u_pp_ab.txt).
const int g_b16_pp = umpr_env_int("UMPR_B16_PP", 2);

template <int W, int TR, int TC, int BN, int WP, int WC, int D = 3, int ABL = 0>
void launch_conv(ConvB16Params p, hipStream_t s) {
  constexpr int BM = TR > 0 ? TR * TC : TC;
  if (TR > 0) p.ntp = ((p.rows + TR - 1) / TR) * (W / TC);
  else p.ntp = (p.ptot + BM - 1) / BM;
  p.nct = p.M / BN;
  long blocks;
  if (p.nct <= 8 && (8 % p.nct) == 0) { const int g = 8 / p.nct; blocks = ((p.ntp + g - 1) / g) * 8; }
  else blocks = ((p.ntp + 7) / 8) * 8 * p.nct;
  if (g_b16_m16 && ABL == 0) {
    if constexpr (WP * WC == 8) {
      if (g_b16_pp == 1 || (g_b16_pp == 2 && (BN == 256 || TR == 0))) {
        conv_bf16_m16_kernel<W, TR, TC, BN, WP, WC, D, true><<<dim3((unsigned)blocks), 64 * WP * WC, 0, s>>>(p);
        return;
      }
    }
    conv_bf16_m16_kernel<W, TR, TC, BN, WP, WC, D><<<dim3((unsigned)blocks), 64 * WP * WC, 0, s>>>(p);
  }
  else conv_bf16_kernel<W, TR, TC, BN, WP, WC, D, ABL><<<dim3((unsigned)blocks), 64 * WP * WC, 0, s>>>(p);
}

// tile choice per map width and output-channel count:
//   M % 256 == 0 : 256 pixels x 256 channels, 8 waves 2 x 4 (each 128 px x 64 ch), one workgroup per CU
//   M % 128 == 0 : 256 x 128, 8 waves 4 x 2 (64 x 64)
//   else (64)    : 256 x 64, 4 waves 4 x 1 (64 x 64), 66 KB of LDS: two workgroups per CU, so one's prologue / epilogue
//                  (K is only 576 deep there) hides behind the other's main loop
//   14 x 14 maps : 57 pixel tiles only - 128-channel tiles (4 x 57 = 228 workgroups) fill the 256 CUs, 256-channel
//                  tiles (114) would leave half of them idle
// UMPR_B16_CFG (A/B runs): tile for the layers with >= 256 output channels.  0: 256 px x 256 ch, 8 waves (one workgroup
// per CU); 1: 256 x 128, 4 waves of 128 x 64 (80 KB of LDS: two workgroups per CU, their barriers interleave);
// 2: 512 x 128, 8 waves of 128 x 64
const int g_b16_cfg = umpr_env_int("UMPR_B16_CFG", 0);

// UMPR_B16_T64=1: the 64-channel output tiles of the 224 / 112 maps take 512 pixels (8 waves, one workgroup per CU) instead of
// 256 (4 waves, two per CU): half the weight re-reads from L2 per output pixel (A/B)
const int g_b16_t64 = umpr_env_int("UMPR_B16_T64", 0);
// UMPR_B16_T3=1: conv1_2 (64 -> 64 at 224x224, K = 576: a workgroup lives for 18 steps) on 4 x 32 = 128-pixel tiles - 48 KB of
// LDS and <= 170 registers, THREE workgroups per CU, so that two others' main loops cover one's prologue / epilogue (A/B)
const int g_b16_t3 = umpr_env_int("UMPR_B16_T3", 0);

int conv_bn_for(int M, int W) {
  if (M % 256 == 0 && W != 14 && g_b16_cfg == 0) return 256;
  return M % 128 == 0 ? 128 : 64;
}

template <int W, int TR, int TC, int TR2, int TC2>
int dispatch_conv_w(const ConvB16Params& p, hipStream_t s) {
  const int BN = conv_bn_for(p.M, W);
  static const int deep = umpr_env_int("UMPR_B16_D", 3);
  static const int abl = umpr_env_int("UMPR_B16_ABL", 0);
  if (BN == 256 && W == 28 && abl == 1) launch_conv<W, TR, TC, 256, 2, 4, 3, 1>(p, s);
  else if (BN == 256 && W == 28 && abl == 2) launch_conv<W, TR, TC, 256, 2, 4, 3, 2>(p, s);
  else if (BN == 256 && W == 28 && abl == 3) launch_conv<W, TR, TC, 256, 2, 4, 3, 3>(p, s);
  else if (BN == 256 && deep == 5) launch_conv<W, TR, TC, 256, 2, 4, 5>(p, s);
  else if (BN == 256) launch_conv<W, TR, TC, 256, 2, 4>(p, s);
  else if (BN == 128 && p.M % 256 == 0 && W != 14 && g_b16_cfg == 1) launch_conv<W, TR, TC, 128, 2, 2>(p, s);
  else if (BN == 128 && p.M % 256 == 0 && W != 14 && g_b16_cfg == 2) launch_conv<W, TR2, TC2, 128, 4, 2>(p, s);
  else if (BN == 128) launch_conv<W, TR, TC, 128, 4, 2>(p, s);
  else if (p.M % 64 == 0 && g_b16_t64 && TR2 > 0) launch_conv<W, TR2, TC2, 64, 8, 1>(p, s);   // 512-pixel tiles, 8 waves
  else if (p.M % 64 == 0 && W == 224 && g_b16_t3) {   // 128-pixel tiles, three workgroups per CU
    if constexpr (W == 224) launch_conv<224, 4, 32, 64, 4, 1>(p, s);
  }
  else if (p.M % 64 == 0) launch_conv<W, TR, TC, 64, 4, 1>(p, s);
  else return -1;
  return 0;
}

constexpr int kWgradB16Wgs = 256;   // workgroups per launch the split-K factor aims at: one per CU (LDS admits one)

// segment / tile / split-K plan of a weight-gradient launch (shared by the workspace query and the launch)
struct WgradPlan { bool big; long nseg; int ntiles, ncit, splits, segs_per_split; };
WgradPlan wgrad_plan(const UmprPF& g, int Cin, int Cout) {
  WgradPlan q;
  q.big = (Cout % 128 == 0);                       // 128 co x 64 ci tiles; else 64 x 64 with two k-split wave groups
  const int tco = q.big ? 128 : 64, tci = 64;
  if (g.W == 224) q.nseg = ((g.rows + 3) / 4) * (224 / 32);        // 2-D segments 4 rows x 32 columns
  else if (g.W == 112) q.nseg = ((g.rows + 7) / 8) * (112 / 16);   // 8 x 16
  else q.nseg = (g.ptot + 127) / 128;                              // 128 consecutive flat pixels
  q.ncit = Cin / tci;
  q.ntiles = (Cout / tco) * q.ncit;
  int splits = (kWgradB16Wgs + q.ntiles - 1) / q.ntiles;
  splits = (splits + 7) / 8 * 8;
  if (splits > q.nseg) splits = (int)q.nseg;
  q.segs_per_split = (int)((q.nseg + splits - 1) / splits);
  q.splits = (int)((q.nseg + q.segs_per_split - 1) / q.segs_per_split);
  return q;
}

template <int W, int TR, int TC, int WCO, int WCI, int KSW>
void launch_wgrad(WgradB16Params p, const WgradPlan& q, hipStream_t s) {
  p.nseg = q.nseg; p.ncit = q.ncit; p.ntiles = q.ntiles; p.splits = q.splits; p.segs_per_split = q.segs_per_split;
  const long blocks = (long)((q.splits + 7) / 8) * 8 * q.ntiles;
  wgrad_bf16_m16_kernel<W, TR, TC, WCO, WCI, KSW><<<dim3((unsigned)blocks), 64 * WCO * WCI * KSW, 0, s>>>(p);
}

}  // namespace

// ---- internal host entry points ----------------------------------------------------------------------------------
size_t umpr_conv_bf16_pack_bytes(int Cin, int Cout) { return (size_t)Cin * Cout * 9 * sizeof(bf16_t) + 1024; }

// forward (transposed = 0): y[M = Cout] = relu?(conv(x[C = Cin]) + bias);  dgrad (transposed = 1): y[M = Cin] =
// conv^T(x[C = Cout]) * [mask > 0].  x, y, mask: pointers to the START of plane 0 (incl. the lead guard) of CB8-PF
// tensors of geometry g.  wpack: scratch of umpr_conv_bf16_pack_bytes.  The output's guards are zeroed here.

// packs the weights of n (<= 12) layers in one launch: layer i -> wpack + offsets[i] (each umpr_conv_bf16_pack_bytes
// apart at least), in the tile order umpr_conv_bf16_run picks for a W[i] x W[i] map
int umpr_conv_bf16_pack_all(const float* const* w, const int* Cin, const int* Cout, const int* W, int n, int transposed,
                            void* wpack, const size_t* offsets, hipStream_t s) {
  UMPR_REQUIRE(n >= 1 && n <= 12, "conv_bf16_pack_all: %d layers", n);
  PackSet ps;
  ps.n = n; ps.transposed = transposed;
  for (int i = 0; i < n; ++i) {
    const int M = transposed ? Cin[i] : Cout[i], C = transposed ? Cout[i] : Cin[i];
    UMPR_REQUIRE(C % 32 == 0 && M % 64 == 0, "conv_bf16_pack_all: channels (%d -> %d)", C, M);
    ps.w[i] = w[i]; ps.wp[i] = reinterpret_cast<bf16_t*>(static_cast<char*>(wpack) + offsets[i]);
    ps.M[i] = M; ps.C[i] = C; ps.Cin[i] = Cin[i]; ps.BN[i] = conv_bn_for(M, W[i]);
  }
  int maxt = 0;
  for (int i = 0; i < n; ++i) { const int t = (ps.M[i] / 16) * (ps.C[i] / 32); if (t > maxt) maxt = t; }
  pack_weights_bf16_multi_kernel<<<dim3(maxt, n), 256, 0, s>>>(ps);
  UMPR_LAUNCH_CHECK("pack_weights_bf16_multi");
  return 0;
}

int umpr_conv_bf16_run(const void* x, const float* w, int transposed, const float* bias, const void* mask, void* y,
                       const UmprPF& g, int Cin, int Cout, int relu, void* wpack, size_t wpack_bytes, hipStream_t s,
                       bool zero_guards) {
  const int M = transposed ? Cin : Cout, C = transposed ? Cout : Cin;
  UMPR_REQUIRE(C % 32 == 0 && M % 64 == 0, "conv_bf16: channels (%d -> %d) must be multiples of 32 / 64", C, M);
  UMPR_REQUIRE(g.H == g.W && (g.W == 224 || g.W == 112 || g.W == 56 || g.W == 28 || g.W == 14),
               "conv_bf16: map %dx%d is not a VGG16 map size", g.H, g.W);
  UMPR_REQUIRE(wpack_bytes >= umpr_conv_bf16_pack_bytes(Cin, Cout), "conv_bf16: weight scratch too small");
  const int BN = conv_bn_for(M, g.W);
  bf16_t* wp = static_cast<bf16_t*>(wpack);
  if (w) {   // w == nullptr: wpack already holds this layer's packed weights (umpr_conv_bf16_pack_all)
    pack_weights_bf16_kernel<<<grid_for((long)M * C * 9, 2048), 256, 0, s>>>(w, wp, M, C, Cin, BN, transposed);
    UMPR_LAUNCH_CHECK("pack_weights_bf16");
  }
  bf16_t* y0 = static_cast<bf16_t*>(y);
  ConvB16Params p;
  p.zlead = zero_guards ? g.lead : 0;
  p.ztail = zero_guards ? g.ps - g.lead - g.ptot : 0;
  p.x = static_cast<const bf16_t*>(x) + g.lead * 8; p.xps = g.ps;
  p.wp = wp; p.bias = bias;
  p.mask = mask ? static_cast<const bf16_t*>(mask) + g.lead * 8 : nullptr; p.mps = g.ps;
  p.y = y0 + g.lead * 8; p.yps = g.ps;
  p.C = C; p.M = M; p.H = g.H; p.ptot = g.ptot; p.rows = g.rows; p.relu = relu; p.nct = 0; p.ntp = 0;
  UmprProfScope prof(transposed ? UMPR_K_B16_DGRAD : UMPR_K_B16_FWD, 2.0 * (double)g.ptot * M * C * 9, s);
  int rc;
  switch (g.W) {
    case 224: rc = dispatch_conv_w<224, 8, 32, 16, 32>(p, s); break;
    case 112: rc = dispatch_conv_w<112, 16, 16, 32, 16>(p, s); break;
    case 56: rc = dispatch_conv_w<56, 0, 256, 0, 512>(p, s); break;
    case 28: rc = dispatch_conv_w<28, 0, 256, 0, 512>(p, s); break;
    default: rc = dispatch_conv_w<14, 0, 256, 0, 512>(p, s); break;
  }
  UMPR_REQUIRE(rc == 0, "conv_bf16: unsupported channel count %d", M);
  UMPR_LAUNCH_CHECK("conv_bf16");
  return 0;
}

size_t umpr_wgrad_bf16_ws_bytes(const UmprPF& g, int Cin, int Cout) {
  const WgradPlan q = wgrad_plan(g, Cin, Cout);
  return (size_t)q.splits * ((size_t)9 * Cout * Cin + Cout) * sizeof(float) + 1024;
}

// dw [Cout][Cin][3][3], db [Cout] (fp32, overwritten or accumulated) from dy, x in CB8-PF (start-of-plane pointers)
int umpr_wgrad_bf16_run(const void* dy, const void* x, float* dw, float* db, const UmprPF& g, int Cin, int Cout,
                        int accumulate, float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "wgrad_bf16: channels (%d, %d) must be multiples of 64", Cin, Cout);
  UMPR_REQUIRE(g.H == g.W && (g.W == 224 || g.W == 112 || g.W == 56 || g.W == 28 || g.W == 14),
               "wgrad_bf16: map %dx%d is not a VGG16 map size", g.H, g.W);
  UMPR_REQUIRE(ws_bytes >= umpr_wgrad_bf16_ws_bytes(g, Cin, Cout), "wgrad_bf16: workspace too small");
  WgradB16Params p;
  p.dy = static_cast<const bf16_t*>(dy) + g.lead * 8; p.dps = g.ps;
  p.x = static_cast<const bf16_t*>(x) + g.lead * 8; p.xps = g.ps;
  p.slab = ws; p.Cin = Cin; p.Cout = Cout;
  const WgradPlan q = wgrad_plan(g, Cin, Cout);
  p.bslab = db ? ws + (size_t)q.splits * 9 * Cout * Cin : nullptr;
  {
    UmprProfScope prof(UMPR_K_B16_WGRAD, 2.0 * (double)g.ptot * Cout * Cin * 9, s);
    if (q.big) {
      switch (g.W) {
        case 224: launch_wgrad<224, 4, 32, 4, 2, 1>(p, q, s); break;
        case 112: launch_wgrad<112, 8, 16, 4, 2, 1>(p, q, s); break;
        case 56: launch_wgrad<56, 0, 128, 4, 2, 1>(p, q, s); break;
        case 28: launch_wgrad<28, 0, 128, 4, 2, 1>(p, q, s); break;
        default: launch_wgrad<14, 0, 128, 4, 2, 1>(p, q, s); break;
      }
    } else {
      switch (g.W) {
        case 224: launch_wgrad<224, 4, 32, 2, 2, 2>(p, q, s); break;
        case 112: launch_wgrad<112, 8, 16, 2, 2, 2>(p, q, s); break;
        case 56: launch_wgrad<56, 0, 128, 2, 2, 2>(p, q, s); break;
        case 28: launch_wgrad<28, 0, 128, 2, 2, 2>(p, q, s); break;
        default: launch_wgrad<14, 0, 128, 2, 2, 2>(p, q, s); break;
      }
    }
  }
  UMPR_LAUNCH_CHECK("wgrad_bf16");
  const long per = (long)9 * Cout * Cin;
  if (q.splits >= 32) {
    const long cols = per / 4 + (db ? (Cout + 3) / 4 : 0);
    wgrad_reduce_wide_kernel<<<(unsigned)((cols + 63) / 64), 1024, 0, s>>>(ws, db ? p.bslab : nullptr, q.splits, per, Cout, dw,
                                                                          db, accumulate);
  } else {
    wgrad_reduce_linear_kernel<<<grid_for(per / 4 + Cout, 2048), 256, 0, s>>>(ws, db ? p.bslab : nullptr, q.splits, per, Cout,
                                                                              dw, db, accumulate);
  }
  UMPR_LAUNCH_CHECK("wgrad_reduce");
  return 0;
}

int umpr_nchw_to_cb8(const float* x, void* y, const UmprPF& g, int C, hipStream_t s) {
  bf16_t* y0 = static_cast<bf16_t*>(y);
  const int planes = (C + 7) / 8;
  nchw_to_cb8_kernel<<<dim3(grid_for(g.ps, 4096), planes), 256, 0, s>>>(x, y0 + g.lead * 8, g.ps, C, g.H, g.W, g.ptot, g.lead);
  UMPR_LAUNCH_CHECK("nchw_to_cb8");
  return 0;
}

int umpr_cb8_to_nchw(const void* x, float* y, const UmprPF& g, int C, hipStream_t s) {
  const int planes = (C + 7) / 8;
  cb8_to_nchw_kernel<<<dim3(grid_for((long)g.N * g.H * g.W, 4096), planes), 256, 0, s>>>(
      static_cast<const bf16_t*>(x) + g.lead * 8, g.ps, y, g.N, C, g.H, g.W);
  UMPR_LAUNCH_CHECK("cb8_to_nchw");
  return 0;
}

int umpr_maxpool2_bf16_fwd_run(const void* x, void* y, const UmprPF& gi, const UmprPF& go, int C, hipStream_t s,
                               bool zero_guards) {
  bf16_t* y0 = static_cast<bf16_t*>(y);
  if (zero_guards) zero_guards_kernel<<<dim3(grid_for(go.ps - go.ptot, 64), C / 8), 256, 0, s>>>(y0, go.ps, go.lead, go.ptot);
  maxpool2_bf16_fwd_kernel<<<dim3(grid_for(go.ptot, 4096), C / 8), 256, 0, s>>>(
      static_cast<const bf16_t*>(x) + gi.lead * 8, gi.ps, y0 + go.lead * 8, go.ps, gi.H, gi.W, go.ptot);
  UMPR_LAUNCH_CHECK("maxpool2_bf16_fwd");
  return 0;
}

int umpr_maxpool2_bf16_bwd_run(const void* x, const void* gy, void* gx, const UmprPF& gi, const UmprPF& go, int C,
                               hipStream_t s) {
  bf16_t* g0 = static_cast<bf16_t*>(gx);
  static const bool per_window = umpr_env_on("UMPR_B16_POOL_BWD_WIN");     // 0: one thread per input pixel (the first version)
  if (per_window && go.ptot + (gi.ps - gi.ptot) < (1l << 31))
    maxpool2_bf16_bwd_relu_win_kernel<<<dim3(grid_for(go.ptot + (gi.ps - gi.ptot), 4096), C / 8), 256, 0, s>>>(
        static_cast<const bf16_t*>(x) + gi.lead * 8, gi.ps, static_cast<const bf16_t*>(gy) + go.lead * 8, go.ps,
        g0 + gi.lead * 8, gi.ps, gi.H, gi.W, (unsigned)go.ptot, gi.ptot, gi.lead);
  else
    maxpool2_bf16_bwd_relu_kernel<<<dim3(grid_for(gi.ps, 4096), C / 8), 256, 0, s>>>(
        static_cast<const bf16_t*>(x) + gi.lead * 8, gi.ps, static_cast<const bf16_t*>(gy) + go.lead * 8, go.ps,
        g0 + gi.lead * 8, gi.ps, gi.H, gi.W, gi.ptot, gi.lead);
  UMPR_LAUNCH_CHECK("maxpool2_bf16_bwd");
  return 0;
}

// first VGG layer in the bf16 path: fp32 images [N][3][H][W] -> bf16 CB8-PF [64 channels], ReLU fused
int umpr_conv1_bf16_fwd(const float* x, const float* w, const float* bias, void* y, const UmprPF& g, hipStream_t s,
                        bool zero_guards) {
  UMPR_REQUIRE(g.W % 32 == 0, "conv1_bf16: width %d is not a multiple of 32", g.W);
  bf16_t* y0 = static_cast<bf16_t*>(y);
  if (zero_guards) zero_guards_kernel<<<dim3(grid_for(g.ps - g.ptot, 64), 8), 256, 0, s>>>(y0, g.ps, g.lead, g.ptot);
  pf_zero_pads_kernel<<<dim3((unsigned)(g.rows < 4096 ? g.rows : 4096), 8), 64, 0, s>>>(y0 + g.lead * 8, g.ps, g.H, g.W, g.rows);
  Conv1B16Params p{x, w, bias, y0 + g.lead * 8, g.ps, g.N, g.H, g.W};
  UmprProfScope prof(UMPR_K_B16_FWD, 2.0 * (double)g.N * g.H * g.W * 64 * 27, s);
  conv1_bf16_fwd_kernel<<<2048, 256, 0, s>>>(p);
  UMPR_LAUNCH_CHECK("conv1_bf16_fwd");
  return 0;
}

// dw[co][c][tap] (+)= sum over the splits of slab[split][tap][co][c], db likewise: one wave per output element (the
// generic reduce walks the 512 slabs with one thread per element: 138 us for 1792 outputs), fixed order
__global__ __launch_bounds__(256) void wgrad1_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                            int splits, float* __restrict__ dw, float* __restrict__ db,
                                                            int accumulate) {
  constexpr int per = 9 * 64 * 3;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= per + (bslab ? 64 : 0)) return;
  float v = 0.f;
  if (i < per) {
    for (int q = lane; q < splits; q += 64) v += slab[(long)q * per + i];
  } else {
    for (int q = lane; q < splits; q += 64) v += bslab[(long)q * 64 + (i - per)];
  }
  v = wave_sum(v);
  if (lane == 0) {
    if (i < per) {
      const int c = i % 3, r = i / 3, co = r % 64, t = r / 64;
      float* d = dw + ((long)co * 3 + c) * 9 + t;
      *d = accumulate ? *d + v : v;
    } else {
      float* d = db + (i - per);
      *d = accumulate ? *d + v : v;
    }
  }
}

constexpr int kWgrad1Splits = 512;
size_t umpr_conv1_bf16_wgrad_ws_bytes() { return (size_t)kWgrad1Splits * (9 * 64 * 3 + 64) * sizeof(float) + 1024; }

// dw [64][3][3][3], db [64] from dY (bf16 CB8-PF, 64 channels) and the fp32 images
int umpr_conv1_bf16_wgrad(const void* dy, const float* x, float* dw, float* db, const UmprPF& g, int accumulate,
                          float* ws, size_t ws_bytes, hipStream_t s) {
  UMPR_REQUIRE(g.W % 16 == 0 && ws_bytes >= umpr_conv1_bf16_wgrad_ws_bytes(), "conv1_bf16_wgrad: bad width / workspace");
  const bf16_t* d0 = static_cast<const bf16_t*>(dy) + g.lead * 8;
  float* bslab = db ? ws + (size_t)kWgrad1Splits * 9 * 64 * 3 : nullptr;
  Wgrad1B16Params p{d0, g.ps, x, ws, bslab, g.N, g.H, g.W};
  {
    UmprProfScope prof(UMPR_K_B16_WGRAD, 2.0 * (double)g.N * g.H * g.W * 64 * 27, s);
    conv1_bf16_wgrad_kernel<<<kWgrad1Splits, 256, 0, s>>>(p);
  }
  UMPR_LAUNCH_CHECK("conv1_bf16_wgrad");
  wgrad1_reduce_kernel<<<(9 * 64 * 3 + 64 + 3) / 4, 256, 0, s>>>(ws, bslab, kWgrad1Splits, dw, db, accumulate);
  UMPR_LAUNCH_CHECK("wgrad1_reduce");
  return 0;
}

// guards of up to 20 CB8-PF tensors (start-of-plane pointers) in one launch
int umpr_pf_zero_guards_multi(void* const* bases, const UmprPF* geos, const int* channels, int n, hipStream_t s) {
  UMPR_REQUIRE(n >= 1 && n <= 20, "zero_guards_multi: %d tensors", n);
  GuardSet gs;
  gs.n = n;
  for (int i = 0; i < n; ++i) {
    gs.base[i] = static_cast<bf16_t*>(bases[i]); gs.ps[i] = geos[i].ps; gs.lead[i] = geos[i].lead; gs.ptot[i] = geos[i].ptot;
    gs.planes[i] = (channels[i] + 7) / 8;
  }
  zero_guards_multi_kernel<<<dim3(64, n), 256, 0, s>>>(gs);
  UMPR_LAUNCH_CHECK("zero_guards_multi");
  return 0;
}
