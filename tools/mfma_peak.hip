// Micro-benchmark: achievable v_mfma_f32_32x32x2_f32 rate on this device (registers only / with LDS operand reads),
// and the clock the chip holds meanwhile (s_memtime vs s_memrealtime).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREAD>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, unsigned long long* clk) {
  __shared__ float S[64 * 130];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 64 * 130; e += 256) S[e] = in[e % 4096];
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a0 = in[lane], a1 = in[64 + lane], b0 = in[128 + lane], b1 = in[192 + lane];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const float* sp0 = S + (lane & 31) + (lane >> 5) * 130;
  for (int it = 0; it < iters; ++it) {
    // the address depends on `it` (and on a value the compiler cannot see), so the reads stay inside the loop
    const float* sp = sp0 + ((it & 1) ^ (iters & 1)) * 32 * 130;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (LDSREAD == 1) {
        na0 = sp[(2 * kk) * 130]; na1 = sp[(2 * kk) * 130 + 32]; nb0 = sp[(2 * kk) * 130 + 64]; nb1 = sp[(2 * kk) * 130 + 96];
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      if (LDSREAD == 2) { __builtin_amdgcn_sched_barrier(0); na0 = sp[(2 * kk) * 130]; na1 = sp[(2 * kk) * 130 + 32]; __builtin_amdgcn_sched_barrier(0); }
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      if (LDSREAD == 2) { __builtin_amdgcn_sched_barrier(0); nb0 = sp[(2 * kk) * 130 + 64]; nb1 = sp[(2 * kk) * 130 + 96]; __builtin_amdgcn_sched_barrier(0); }
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
      if (LDSREAD) __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float *in, *out; unsigned long long* clk;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 16);
  float* h = (float*)malloc(4096 * 4);
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, 4096 * 4, hipMemcpyHostToDevice);
  for (int lds = 0; lds < 3; ++lds)
    for (int blocks : {256, 512, 1024}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (lds == 2) k<2><<<blocks, 256>>>(out, in, iters, clk);
        else if (lds) k<1><<<blocks, 256>>>(out, in, iters, clk); else k<0><<<blocks, 256>>>(out, in, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      double flop = (double)blocks * 4 * iters * 16 * 4 * 4096.0;
      printf("lds=%d blocks=%4d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  clock %.0f MHz\n", lds, blocks, blocks / 256, ms,
             flop / ms / 1e9, (double)hc[0] / (double)hc[1] * 100.0);
    }
  return 0;
}
