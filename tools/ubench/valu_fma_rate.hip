// Issue rate of the fp32 FMA forms the thin conv kernels (conv_thin.hip) use, per wave and per SIMD:
//   0: v_fma_f32      vdst, v, s, vacc          (k_up_thin: activation in a VGPR, wave-uniform weight in an SGPR)
//   1: v_pk_fma_f32   vdst, v[x:x+1], s[w:w+1], vacc  op_sel_hi:[0,1,1]   (k_up_thin_pk: activation broadcast, weight pair in SGPRs)
//   2: v_pk_fma_f32   vdst, v[a:a+1], v[b:b+1], vacc                        (all-VGPR packed form)
// 8 independent accumulators per form, `waves` waves per SIMD (block = 256 * waves threads, one block per CU).
// Prints cycles per instruction per wave (s_memtime) and the chip-level TFLOP/s from the wall clock.
//   hipcc --offload-arch=gfx950 -O2 -w -o tools/ubench/valu_fma_rate tools/ubench/valu_fma_rate.hip && tools/ubench/valu_fma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM>
__global__ void k(float* out, long long* cyc, int iters, float w0, float w1) {
  f32x2 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{(float)threadIdx.x, 1.f};
  f32x2 x = {1.0f + threadIdx.x * 1e-6f, 0.5f};
  f32x2 wv = {w0, w1};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (FORM == 0) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(x[0]), "s"(w0));
        } else if (FORM == 1) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(wv));
        } else {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(wv));
        }
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int FORM>
static void run(int waves, const char* name) {
  const int iters = 4096, blocks = 256, threads = 256 * waves;
  float* out; long long* cyc;
  hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 64, 1.0001f, 0.9999f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.9999f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double ninst = (double)iters * 32;
  const double flop_per_inst_lane = FORM == 0 ? 2.0 : 4.0;
  const double tflops = ninst * flop_per_inst_lane * (double)blocks * threads / (ms * 1e-3) / 1e12;
  printf("%-44s waves/SIMD %d: %6.2f counter ticks / instruction / wave, %7.1f TFLOP/s chip (%.3f ms)\n", name, waves, c / ninst, tflops, ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int waves = 1; waves <= 4; waves *= 2) {
    run<0>(waves, "v_fma_f32 v, v, s");
    run<1>(waves, "v_pk_fma_f32 v, v(bcast), s[pair]");
    run<2>(waves, "v_pk_fma_f32 v, v[pair], v[pair]");
  }
  return 0;
}
