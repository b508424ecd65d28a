// What can HBM deliver for the traffic MIX of the thin convolution kernels?  (conv1 forward at B = 1024: 50 MB read as 256-byte
// image rows, 138 MB written as 128-byte NHWC pixels -- 73 % writes; the kernels sit at 3.6-3.8 TB/s, the guide's float4 copy at
// 6.3.)  Streaming kernels with NOTHING but the accesses, one line of output per (pattern, read MB, write MB):
//   w16   every lane 16-byte accesses, a wave moves 1 KB contiguous per instruction
//   w4    every lane 4-byte accesses, a wave moves 256 B contiguous per instruction (k_down_thin's loads; its stores are 2 x 128 B)
//   w16p  16-byte stores in the TRANSPOSED-product pattern: lane (i, h) writes 16 B at pixel i * 128 + 32 g + 16 h, g = 0..3
//         (four instructions complete 32 lines: what an MFMA epilogue without LDS staging would issue)
//   nt    w16 with non-temporal stores (and loads)
// persistent grid of G workgroups per CU, each workgroup walks 16 KB output blocks (= one unit of k_down_thin) + the matching
// share of the input.   hipcc --offload-arch=gfx950 -O3 -o hbm_mix hbm_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one "unit" = WQ float4 written + RQ float4 read per thread of a 256-thread workgroup
template <int PAT, int RQ, int WQ>
__global__ __launch_bounds__(256) void k_mix(const float* __restrict__ src, float* __restrict__ dst, int n_units) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  f32x4 keep = {0.f, 0.f, 0.f, 0.f};
  for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
    const float* s = src + (long)u * (RQ * 1024);
    float* d = dst + (long)u * (WQ * 1024);
    f32x4 r[RQ > 0 ? RQ : 1];
    if (PAT == 1) {                                       // dword accesses
#pragma unroll
      for (int q = 0; q < RQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[q][e] = s[(q * 4 + e) * 256 + tid];
    } else {
#pragma unroll
      for (int q = 0; q < RQ; ++q) {
        const f32x4* p = reinterpret_cast<const f32x4*>(s) + q * 256 + tid;
        r[q] = PAT == 3 ? __builtin_nontemporal_load(p) : *p;
      }
    }
    f32x4 v = keep;
#pragma unroll
    for (int q = 0; q < RQ; ++q) v += r[q];
    if (PAT == 1) {
#pragma unroll
      for (int q = 0; q < WQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) d[(q * 4 + e) * 256 + tid] = v[e] + (float)q;
    } else if (PAT == 2) {                                // transposed-product epilogue: wave owns 4 KB = 32 pixels x 128 B per 4 stores
      const int i = lane & 31, h = lane >> 5;
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        const int blk = q >> 2, g = q & 3;                // WQ / 4 blocks of 4 KB per wave
        float* p = d + (wv * (WQ / 4) + blk) * 1024 + i * 32 + 8 * g + 4 * h;
        *reinterpret_cast<f32x4*>(p) = v + (float)q;
      }
    } else {
#pragma unroll
      for (int q = 0; q < WQ; ++q) {
        f32x4* p = reinterpret_cast<f32x4*>(d) + q * 256 + tid;
        if (PAT == 3) __builtin_nontemporal_store(v + (float)q, p); else *p = v + (float)q;
      }
    }
    keep = v * 0.5f;
  }
  if (keep[0] == 123.456f) dst[0] = keep[1];
}

template <int PAT, int RQ, int WQ>
static void run(const char* name, float* src, float* dst, int n_units, int wg_per_cu) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * wg_per_cu < n_units ? 256 * wg_per_cu : n_units;
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_mix<PAT, RQ, WQ>), dim3(grid), dim3(256), 0, 0, src, dst, n_units);
  const int reps = 20;
  CHECK(hipEventRecord(e0, 0));
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_mix<PAT, RQ, WQ>), dim3(grid), dim3(256), 0, 0, src, dst, n_units);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  const double rmb = (double)n_units * RQ * 4096 / 1e6, wmb = (double)n_units * WQ * 4096 / 1e6;
  printf("%-5s read %6.1f MB  write %6.1f MB  %d wg/CU  %7.1f us  %5.2f TB/s\n", name, rmb, wmb, wg_per_cu, us, (rmb + wmb) / us);
}

int main() {
  const int n_units = 8192;                               // B = 1024 images x 8 units
  float *src, *dst;
  CHECK(hipMalloc(&src, (size_t)n_units * 8 * 4096 + 4096));
  CHECK(hipMalloc(&dst, (size_t)n_units * 8 * 4096 + 4096));
  CHECK(hipMemset(src, 0, (size_t)n_units * 8 * 4096));
  for (int g : {2, 4, 8}) {
    // conv1 forward's mix: 6 KB in (2 x 4 KB here: 67 MB), 16 KB out per unit
    run<0, 2, 4>("w16", src, dst, n_units, g);
    run<1, 2, 4>("w4", src, dst, n_units, g);
    run<2, 2, 4>("w16p", src, dst, n_units, g);
    run<3, 2, 4>("nt", src, dst, n_units, g);
  }
  for (int g : {4, 8}) {
    run<0, 4, 4>("w16", src, dst, n_units, g);            // copy
    run<1, 4, 4>("w4", src, dst, n_units, g);
    run<0, 0, 4>("w16", src, dst, n_units, g);            // write only
    run<1, 0, 4>("w4", src, dst, n_units, g);
    run<2, 0, 4>("w16p", src, dst, n_units, g);
    run<0, 4, 1>("w16", src, dst, n_units, g);            // read mostly (the weight-gradient kernels' mix)
    run<1, 4, 1>("w4", src, dst, n_units, g);
    run<0, 6, 4>("w16", src, dst, n_units, g);            // convT3 forward + likelihood: 184 MB in, 100 MB out
  }
  // larger units (more in flight per workgroup)
  run<0, 4, 8>("w16", src, dst, n_units / 2, 4);
  run<0, 2, 8>("w16", src, dst, n_units / 2, 4);
  return 0;
}
