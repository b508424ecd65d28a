// Sustained fp32 MFMA issue-rate micro-benchmark (calibrates the roofline denominator on the box).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 4; ++e) s += acc[k][e];
  if (s == 12345.f) out[threadIdx.x] = s;
}

// Same issue pattern, but with operands that differ per lane and per step (8 pseudo-random register pairs,
// values ~U(-1,1) like activations/weights): the datapath toggles like in a real kernel, which matters
// for the power-managed clock.
template <int NACC>
__global__ __launch_bounds__(512) void k32r(float* out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  float ar[8], br[8];
  unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u; ar[u] = (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f;
    h = h * 1664525u + 1013904223u; br[u] = (float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[u], br[(u + k) & 7], acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <typename F> static double run(F launch, double flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flops * 5 / (ms * 1e-3) / 1e12;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int iters = 2000;
  for (int threads : {256, 512}) {
    const int waves = threads / 64;
    const double n32 = 256.0 * waves * iters * 8;
    printf("32x32x2 f32, %d waves/CU, 1 acc : %.1f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, n32 * 1 * 4096));
    printf("32x32x2 f32, %d waves/CU, 2 acc : %.1f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, n32 * 2 * 4096));
    printf("16x16x4 f32, %d waves/CU, 1 acc : %.1f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, n32 * 1 * 2048));
    printf("16x16x4 f32, %d waves/CU, 2 acc : %.1f TFLOP/s\n", waves, run([&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, n32 * 2 * 2048));
  }
  // long runs (~50 ms each) so that the clock governor settles: constant vs random operands
  const int long_iters = 60000;
  const double nl = 256.0 * 8 * long_iters * 8 * 2 * 4096;
  printf("32x32x2 f32, 8 waves/CU, 2 acc, CONSTANT operands, long run : %.1f TFLOP/s\n",
         run([&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(512), 0, 0, out, long_iters, 1.f, 2.f); }, nl));
  printf("32x32x2 f32, 8 waves/CU, 2 acc, RANDOM operands,   long run : %.1f TFLOP/s\n",
         run([&] { hipLaunchKernelGGL(k32r<2>, dim3(256), dim3(512), 0, 0, out, long_iters, 12345u); }, nl));
  printf("32x32x2 f32, 8 waves/CU, 2 acc, ZERO operands,     long run : %.1f TFLOP/s\n",
         run([&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(512), 0, 0, out, long_iters, 0.f, 0.f); }, nl));
  return 0;
}
