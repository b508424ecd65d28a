// How fast does ONE accumulator chain of v_mfma_f32_32x32x2_f32 run (every MFMA depends on the previous one: k_down_thin's
// 8 C MFMAs per 32-pixel row), alone and with an epilogue's VALU / store instructions issued between the MFMAs?  And two chains
// per wave, and two such waves per SIMD?  (k_down_thin_ws spent 42 us of compute-wave time on 23 us of matrix-core work.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip
// one line per variant: waves per SIMD, chains per wave, VALU per MFMA, stores per MFMA -> cycles per MFMA per SIMD (at the
// clock the run sustained, from s_memrealtime-free wall time and the MFMA count), TFLOP/s
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CHAINS, int NV, int NS>
__global__ __launch_bounds__(512) void k_chain(float* __restrict__ out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float a = seed + lane, b = seed * 0.5f;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = seed + q;
  float* o = out + ((long)blockIdx.x * blockDim.x + threadIdx.x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);       // independent VALU work
#pragma unroll
        for (int q = 0; q < NS; ++q) o[(long)(kk & 15) * 1048576] = v[q];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[c][e];
#pragma unroll
  for (int q = 0; q < 8; ++q) s += v[q];
  if (s == 123.456f) out[0] = s;
}

template <int CHAINS, int NV, int NS>
static void run(float* buf, int waves_per_simd) {
  const int threads = 256 * waves_per_simd;                // 4 SIMDs x waves_per_simd waves, one workgroup per CU
  const int iters = 400 / CHAINS;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_chain<CHAINS, NV, NS>), dim3(256), dim3(threads), 0, 0, buf, iters, 1.0f);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_chain<CHAINS, NV, NS>), dim3(256), dim3(threads), 0, 0, buf, iters, 1.0f);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)iters * 24 * CHAINS * waves_per_simd;
  const double ns_per = ms * 1e6 / mfma_per_simd;
  const double tf = mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD %d  chains/wave %d  VALU/MFMA %d  stores/MFMA %d   %6.1f ns per MFMA per SIMD (%5.1f cycles at 2.1 GHz)  %6.1f TFLOP/s\n",
         waves_per_simd, CHAINS, NV, NS, ns_per, ns_per * 2.1, tf);
}

int main() {
  float* buf;
  CHECK(hipMalloc(&buf, (size_t)64 << 20));
  run<1, 0, 0>(buf, 1);
  run<1, 4, 0>(buf, 1);
  run<1, 8, 0>(buf, 1);
  run<1, 12, 0>(buf, 1);
  run<1, 6, 1>(buf, 1);
  run<2, 0, 0>(buf, 1);
  run<2, 6, 0>(buf, 1);
  run<2, 6, 1>(buf, 1);
  run<1, 0, 0>(buf, 2);
  run<1, 6, 0>(buf, 2);
  run<1, 6, 1>(buf, 2);
  run<2, 6, 1>(buf, 2);
  run<1, 6, 1>(buf, 4);
  return 0;
}
