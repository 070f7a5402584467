// Layout probe for the bf16 kernels (run once on an MI355X: hipcc --offload-arch=gfx950 -O2 -o mfma_bf16_probe ...).
//  (1) v_mfma_f32_32x32x16_bf16 operand / result lane maps against a host GEMM on exact small integers,
//      with asymmetric operands (cdna_hip_programming.md section 3);
//  (2) ds_read_b64_tr_b16: which (row, column) every lane receives, and a K-major ("transposed") GEMM built from it the
//      way the weight-gradient kernel does (both operands stored [k][m] / [k][n] in LDS);
//  (3) global_load_lds_dwordx4: LDS destination = M0 base + lane * 16 with per-lane source addresses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// (1) A [32][16] row-major, B [16][32] row-major (bf16), D [32][32] fp32
__global__ void k_mfma(const __bf16* A, const __bf16* B, float* D) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// (2a) raw dump: LDS image M[64 rows][64 cols] of 16-bit codes row*256+col; every lane of the wave issues ONE tr read at
// address &M[arow[l]][acol[l]] and writes its four 16-bit results.
__global__ void k_tr_dump(const int* arow, const int* acol, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short M[64 * 64];
  for (int e = threadIdx.x; e < 64 * 64; e += 64) M[e] = (unsigned short)((e / 64) * 256 + (e % 64));
  __syncthreads();
  const int l = threadIdx.x;
  typedef short s4 __attribute__((ext_vector_type(4)));
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(&M[arow[l] * 64 + acol[l]]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

// (2b) K-major GEMM: G [16 k][32 m] and X [16 k][32 n] (bf16, row = k) in LDS, D[m][n] = sum_k G[k][m] X[k][n] with both
// fragments fetched by tr reads, addressing as planned for the weight-gradient kernel:
//   16-lane group g = l >> 4 handles columns 16*(g&1) .. +15 and k-half h = g >> 1 (k = 8h .. 8h+7: two reads, 4 k each);
//   inside a group lane 4q+p supplies row (k0 + q), columns c0 + 4p .. 4p+3.
__global__ void k_tr_gemm(const __bf16* G, const __bf16* X, float* D) {
  __shared__ __attribute__((aligned(16))) __bf16 Gs[16 * 32];
  __shared__ __attribute__((aligned(16))) __bf16 Xs[16 * 32];
  const int l = threadIdx.x;
  for (int e = l; e < 16 * 32; e += 64) { Gs[e] = G[e]; Xs[e] = X[e]; }
  __syncthreads();
  const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const int c0 = 16 * (g & 1), h = g >> 1;
  bf16x8 a, b;
  for (int half4 = 0; half4 < 2; ++half4) {
    const int krow = 8 * h + 4 * half4 + q;
    bf16x4 ta = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(&Gs[krow * 32 + c0 + 4 * p]));
    bf16x4 tb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(&Xs[krow * 32 + c0 + 4 * p]));
    for (int j = 0; j < 4; ++j) { a[4 * half4 + j] = ta[j]; b[4 * half4 + j] = tb[j]; }
  }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  const int r = l & 31, hh = l >> 5;
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * hh) * 32 + r] = acc[i];
}

// (3) LDS-DMA: lane l copies 16 B from src + perm[l] * 16 bytes; destination = wave-uniform base + l * 16
__global__ void k_dma(const unsigned* src, const int* perm, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned L[2 * 256];
  const int l = threadIdx.x;
  for (int e = l; e < 512; e += 64) L[e] = 0xdeadbeefu;
  __syncthreads();
  const unsigned* s = src + perm[l] * 4;
  const unsigned dst = (unsigned)(size_t)(lds_ptr_t)(&L[256]);   // second KiB
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = l; e < 512; e += 64) out[e] = L[e];
}

static __bf16 tobf(float f) { return (__bf16)f; }

int main() {
  // ---------------- (1)
  std::vector<__bf16> A(32 * 16), B(16 * 32);
  std::vector<float> Af(32 * 16), Bf(16 * 32), Dref(32 * 32, 0.f), D(32 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) { Af[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3); A[i * 16 + k] = tobf(Af[i * 16 + k]); }
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) { Bf[k * 32 + j] = (float)((k * 11 + j * 2 + (j > 9)) % 9 - 4); B[k * 32 + j] = tobf(Bf[k * 32 + j]); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) Dref[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
  __bf16 *dA, *dB; float* dD;
  CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, D.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  k_mfma<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 1024; ++i) bad += D[i] != Dref[i];
  printf("[1] mfma_f32_32x32x16_bf16 lane maps (A[r][8h+j], B[8h+j][r], D row=(i&3)+8(i>>2)+4h col=r): %s (%d wrong)\n", bad ? "FAIL" : "PASS", bad);

  // ---------------- (2a)
  std::vector<int> arow(64), acol(64);
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    arow[l] = 8 * (g >> 1) + q + 16 * 0;          // rows: k-half by group pair
    acol[l] = 16 * (g & 1) + 4 * p;
  }
  int *dr, *dc; unsigned short* dout;
  CK(hipMalloc(&dr, 256)); CK(hipMalloc(&dc, 256)); CK(hipMalloc(&dout, 512));
  CK(hipMemcpy(dr, arow.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, acol.data(), 256, hipMemcpyHostToDevice));
  k_tr_dump<<<1, 64>>>(dr, dc, dout);
  std::vector<unsigned short> tr(256);
  CK(hipMemcpy(tr.data(), dout, 512, hipMemcpyDeviceToHost));
  int bad2 = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15;
    for (int j = 0; j < 4; ++j) {
      const int row = 8 * (g >> 1) + j, col = 16 * (g & 1) + i;     // expected: lane i of the group gets column i, row q=j
      if (tr[l * 4 + j] != row * 256 + col) ++bad2;
    }
  }
  printf("[2a] ds_read_b64_tr_b16: lane (group g, i) element j == M[row0+j][col0+i]: %s (%d wrong)\n", bad2 ? "FAIL" : "PASS", bad2);
  if (bad2) for (int l = 0; l < 64; ++l) printf("   lane %2d addr(r%2d,c%2d): (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, arow[l], acol[l],
      tr[l*4]>>8, tr[l*4]&255, tr[l*4+1]>>8, tr[l*4+1]&255, tr[l*4+2]>>8, tr[l*4+2]&255, tr[l*4+3]>>8, tr[l*4+3]&255);

  // ---------------- (2b)
  std::vector<__bf16> G(16 * 32), X(16 * 32);
  std::vector<float> Gf(16 * 32), Xf(16 * 32), D2ref(32 * 32, 0.f), D2(32 * 32);
  for (int k = 0; k < 16; ++k) for (int m = 0; m < 32; ++m) { Gf[k * 32 + m] = (float)((k * 7 + m * 3) % 11 - 5); G[k * 32 + m] = tobf(Gf[k * 32 + m]); }
  for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) { Xf[k * 32 + n] = (float)((k * 5 + n * 13 + (n > 20)) % 13 - 6); X[k * 32 + n] = tobf(Xf[k * 32 + n]); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) D2ref[m * 32 + n] += Gf[k * 32 + m] * Xf[k * 32 + n];
  CK(hipMemcpy(dA, G.data(), G.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, X.data(), X.size() * 2, hipMemcpyHostToDevice));
  k_tr_gemm<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D2.data(), dD, D2.size() * 4, hipMemcpyDeviceToHost));
  int bad3 = 0;
  for (int i = 0; i < 1024; ++i) bad3 += D2[i] != D2ref[i];
  printf("[2b] K-major GEMM through tr reads (wgrad addressing): %s (%d wrong)\n", bad3 ? "FAIL" : "PASS", bad3);

  // ---------------- (3)
  std::vector<unsigned> src(64 * 4), out(512);
  std::vector<int> perm(64);
  for (int i = 0; i < 256; ++i) src[i] = 1000u + i;
  for (int l = 0; l < 64; ++l) perm[l] = (l * 37 + 5) % 64;
  unsigned *dsrc, *dout2; int* dperm;
  CK(hipMalloc(&dsrc, 1024)); CK(hipMalloc(&dout2, 2048)); CK(hipMalloc(&dperm, 256));
  CK(hipMemcpy(dsrc, src.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dperm, perm.data(), 256, hipMemcpyHostToDevice));
  k_dma<<<1, 64>>>(dsrc, dperm, dout2);
  CK(hipMemcpy(out.data(), dout2, 2048, hipMemcpyDeviceToHost));
  int bad4 = 0;
  for (int e = 0; e < 256; ++e) bad4 += out[e] != 0xdeadbeefu;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) bad4 += out[256 + l * 4 + j] != 1000u + perm[l] * 4 + j;
  printf("[3] global_load_lds_dwordx4 (dst = M0 + lane*16, per-lane source): %s (%d wrong)\n", bad4 ? "FAIL" : "PASS", bad4);
  return (bad || bad2 || bad3 || bad4) ? 1 : 0;
}
