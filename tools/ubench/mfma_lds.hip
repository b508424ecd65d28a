// Does an LDS-fed fp32 MFMA loop reach the MFMA issue rate?  Variants:
//   regs   : operands held in 8 different VGPRs (no LDS traffic in the loop)
//   lds128 : 2 x ds_read_b128 per 4 MFMAs (32x32x2) -- the conv kernels' operand pattern
//   lds128p: same, operands prefetched one group ahead
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 16384; e += blockDim.x) lds[e] = in[e & 1023];
  __syncthreads();
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  const float* ap = lds + lane * 4;            // conflict-free: consecutive lanes, 16 bytes each
  const float* bp = lds + 8192 + lane * 4;
  if (MODE == 0) {
    f32x4 a = *(const f32x4*)ap, b = *(const f32x4*)bp;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[j], 0, 0, 0);
    }
  } else if (MODE == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        f32x4 a = *(const f32x4*)(ap + g * 256), b = *(const f32x4*)(bp + g * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[j], 0, 0, 0);
      }
    }
  } else {
    f32x4 A[2], B[2];
    A[0] = *(const f32x4*)ap; B[0] = *(const f32x4*)bp;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int cur = g & 1;
        A[cur ^ 1] = *(const f32x4*)(ap + ((g + 1) & 7) * 256); B[cur ^ 1] = *(const f32x4*)(bp + ((g + 1) & 7) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][j], B[cur][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  if (s == 12345.f) out[tid] = s;
}

template <int MODE> static void bench(const char* name, float* out, float* in, int threads) {
  const int iters = 1000;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, out, in, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * (threads / 64) * iters * 32 * 4096 * 5;
  printf("%-8s %d waves/CU : %.1f TFLOP/s\n", name, threads / 64, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *out, *in; hipMalloc(&out, 4096); hipMalloc(&in, 4096);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 7919) % 1000) / 1000.f - 0.5f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    bench<0>("regs", out, in, threads);
    bench<1>("lds128", out, in, threads);
    bench<2>("lds128p", out, in, threads);
  }
  return 0;
}
