// Stand-alone reproducer for the run-to-run noise of text_ops.hip's first merge_bwd_dx_kernel ("A kernel that was not
// reproducible").  The kernel under test (a copy of that kernel) runs on one stream with FIXED inputs, again and again, next to a
// co-runner on a second stream that keeps every CU busy; each result is compared bit for bit with the result of a run on the quiet
// device.  Co-runners: none / a VALU loop / a bf16 MFMA loop / a streaming copy (memory pressure) / transposed LDS reads.
// (tools/hazard_repro.py: the same with the LIBRARY's kernels as neighbours - only the bf16 weight gradient triggers it.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/hazard_repro tools/hazard_repro.hip && tools/hazard_repro [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

constexpr int MD = 128, MK = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- the kernel under test: d_repr[m][k] = sum_n dpre[m][n] W[n][k], dpre = d_out (1 - out^2); grid (cdiv(B, 4), 2)
__global__ __launch_bounds__(256) void dx_pipelined(const float* __restrict__ out, const float* __restrict__ d_out,
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

// ---- co-runners
__global__ __launch_bounds__(256) void busy_valu(float* sink, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.9999f + 1e-4f; }
  sink[blockIdx.x * 256 + threadIdx.x] = a + b;
}
__global__ __launch_bounds__(256) void busy_mfma(float* sink, int iters) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x + e)); b[e] = (__bf16)(0.02f * e); }
  for (int i = 0; i < iters; ++i) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[1], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}
// transposed LDS reads (ds_read_b64_tr_b16, new on gfx950; the bf16 weight-gradient kernel is the only library kernel using them)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256) void busy_tr(float* sink, int iters) {
  __shared__ __attribute__((aligned(1024))) __bf16 S[32 * 1024];
  for (int e = threadIdx.x; e < 32 * 1024; e += 256) S[e] = (__bf16)(0.001f * (e & 1023));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __bf16* base = S + wave * 8192 + (lane & 15) * 8 + (lane >> 4) * 512;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(lds_ptr_t)(base + (i & 7) * 64));
    acc += (float)a[0] + (float)a[1] + (float)a[2] + (float)a[3];
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
// a copy of bf16_conv.hip's wgrad_reduce_wide_kernel: 1024-thread workgroups, float4 partial sums exchanged through 16 KB of LDS
__global__ __launch_bounds__(1024) void reduce_wide(const float* __restrict__ slab, int splits, long per, float* __restrict__ dw) {
  __shared__ float4 red[16][64];
  const long nv = per / 4;
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + c;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < nv) {
    const float4* s4 = reinterpret_cast<const float4*>(slab);
    for (int q = g; q < splits; q += 16) { const float4 u = s4[(long)q * nv + i]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
  }
  red[g][c] = a;
  __syncthreads();
  if (g == 0 && i < nv) {
    float4 r = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 u = red[k][c]; r.x += u.x; r.y += u.y; r.z += u.z; r.w += u.w; }
    reinterpret_cast<float4*>(dw)[i] = r;
  }
}
__global__ void tiny(float* sink) { if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1.f; }
__global__ __launch_bounds__(256) void small_store(float* sink, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) sink[i] = 0.5f * i;
}
__global__ __launch_bounds__(256) void busy_copy(const float4* __restrict__ src, float4* __restrict__ dst, long n, int passes) {
  for (int p = 0; p < passes; ++p)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[(i + p) % n];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 400;
  const int B = 4;
  float *out, *d_out, *Wu, *Wi, *res, *sink;
  float4 *cs, *cd;
  const long NC = 64L << 20;   // 1 GiB of float4 for the copy co-runner
  CK(hipMalloc(&out, B * MD * 4)); CK(hipMalloc(&d_out, B * MD * 4)); CK(hipMalloc(&Wu, MD * MK * 4)); CK(hipMalloc(&Wi, MD * MK * 4));
  CK(hipMalloc(&res, 2 * B * MK * 4)); CK(hipMalloc(&sink, 4096 * 256 * 4)); CK(hipMalloc(&cs, NC * 16)); CK(hipMalloc(&cd, NC * 16));
  std::vector<float> h(MD * MK);
  srand(7);
  auto fill = [&](float* d, int n, float scale) { for (int i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f); return hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); };
  CK(fill(out, B * MD, 1.6f)); CK(fill(d_out, B * MD, 0.2f)); CK(fill(Wu, MD * MK, 0.25f)); CK(fill(Wi, MD * MK, 0.25f));
  CK(hipMemset(cs, 0, NC * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  std::vector<float> ref(2 * B * MK), got(2 * B * MK);
  dx_pipelined<<<dim3(1, 2), 256, 0, sa>>>(out, d_out, Wu, Wi, B, res, res + B * MK);
  CK(hipStreamSynchronize(sa));
  CK(hipMemcpy(ref.data(), res, ref.size() * 4, hipMemcpyDeviceToHost));
  const char* names[8] = {"no co-runner", "VALU loop on every CU", "bf16 MFMA loop on every CU", "streaming copy (HBM pressure)",
                          "transposed LDS reads on every CU", "40 one-wave kernels per launch", "40 short 1024-workgroup kernels per launch",
                          "1024-thread reduce kernel (LDS exchange)"};
  int total_bad = 0;
  for (int mode = 0; mode < 8; ++mode) {
    int bad_launches = 0; long bad_elems = 0; int first = -1;
    for (int it = 0; it < launches; ++it) {
      if (it % 20 == 0) {   // keep the co-runner's queue full
        if (mode == 1) busy_valu<<<2048, 256, 0, sb>>>(sink, 400000);
        if (mode == 2) busy_mfma<<<2048, 256, 0, sb>>>(sink, 60000);
        if (mode == 3) busy_copy<<<2048, 256, 0, sb>>>(cs, cd, NC, 2);
        if (mode == 4) busy_tr<<<2048, 256, 0, sb>>>(sink, 200000);
      }
      if (mode == 5) for (int q = 0; q < 40; ++q) tiny<<<1, 64, 0, sb>>>(sink);
      if (mode == 6) for (int q = 0; q < 40; ++q) small_store<<<1024, 256, 0, sb>>>(sink, 1024 * 256);
      if (mode == 7) for (int q = 0; q < 3; ++q)      // 512 x 512 x 9 weights, 32 partial slabs (reads the copy buffers: any values)
        reduce_wide<<<(unsigned)((9L * 512 * 512 / 4 + 63) / 64), 1024, 0, sb>>>(reinterpret_cast<const float*>(cs), 32, 9L * 512 * 512,
                                                                                reinterpret_cast<float*>(cd));
      CK(hipMemsetAsync(res, 0xFF, 2 * B * MK * 4, sa));
      dx_pipelined<<<dim3(1, 2), 256, 0, sa>>>(out, d_out, Wu, Wi, B, res, res + B * MK);
      CK(hipMemcpyAsync(got.data(), res, got.size() * 4, hipMemcpyDeviceToHost, sa));
      CK(hipStreamSynchronize(sa));
      long nb = 0;
      for (size_t i = 0; i < got.size(); ++i) if (memcmp(&got[i], &ref[i], 4)) { if (!nb && first < 0) first = (int)i; ++nb; }
      if (nb) { ++bad_launches; bad_elems += nb; }
    }
    CK(hipDeviceSynchronize());
    printf("%-32s: %d of %d launches differ from the quiet-device result (%ld elements; first flat index %d = column %d)\n", names[mode],
           bad_launches, launches, bad_elems, first, first < 0 ? -1 : first % MK);
    total_bad += bad_launches;
  }
  printf(total_bad ? "NOT reproducible\n" : "reproducible\n");
  return 0;
}
