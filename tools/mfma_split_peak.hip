// Micro-benchmark: fp32 GEMM inner loop emulated on the bf16 matrix pipe.  Each fp32 operand is split exactly into
// 3 bf16 pieces (hi + mid + lo = x); NP of the 9 cross products are issued per (A tile, B tile) pair with
// v_mfma_f32_32x32x16_bf16.  All fragments are re-read from LDS every k-step (ds_read_b128, fragment-order image).
// Prints the fp32-equivalent TFLOP/s (2*M*N*K per tile product, counted once).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NP>
__global__ __launch_bounds__(256) void k(float* out, const uint4* in, int iters) {
  // image: 2 phases x 12 fragments (A: 3 pieces x 2 tiles, B: 3 pieces x 2 tiles) x 64 lanes x 16 B = 24 KB
  __shared__ uint4 S[2 * 12 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 2 * 12 * 64; e += 256) S[e] = in[e];
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint4* sp = S + ((it & 1) ^ (iters & 1)) * 12 * 64 + lane;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        uint4 ua = sp[(t * 3 + p) * 64], ub = sp[(6 + t * 3 + p) * 64];
        a[t][p] = *reinterpret_cast<bf16x8*>(&ua);
        b[t][p] = *reinterpret_cast<bf16x8*>(&ub);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
          for (int pb = 0; pb < 3; ++pb) {
            const bool use = NP == 9 ? true : NP == 8 ? (pa + pb < 4) : NP == 6 ? (pa + pb < 3) : (pa + pb < 2);
            if (use) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i * 2 + j], 0, 0, 0);
          }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NP>
void run(float* out, const uint4* in, int iters) {
  for (int blocks : {256, 512}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      k<NP><<<blocks, 256>>>(out, in, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 4 * (2.0 * 32 * 32 * 16);  // fp32-equivalent
    printf("pieces=%d blocks=%4d (%d waves/SIMD): %.3f ms  %.1f fp32-equivalent TFLOP/s  (%.0f bf16 MFMA TFLOP/s)\n", NP, blocks,
           blocks / 256, ms, flop / ms / 1e9, flop * NP / ms / 1e9);
  }
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint4* in; float* out;
  (void)hipMalloc(&in, 2 * 12 * 64 * 16); (void)hipMalloc(&out, 512 * 256 * 4);
  unsigned short* h = (unsigned short*)malloc(2 * 12 * 64 * 16);
  for (int i = 0; i < 2 * 12 * 64 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff)) | ((rand() & 1) << 15);  // |x| in [~0.008, 0.03)
  (void)hipMemcpy(in, h, 2 * 12 * 64 * 16, hipMemcpyHostToDevice);
  run<9>(out, in, iters); run<8>(out, in, iters); run<6>(out, in, iters); run<3>(out, in, iters);
  return 0;
}
