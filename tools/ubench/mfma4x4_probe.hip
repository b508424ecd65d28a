// Lane / register layout of v_mfma_f32_4x4x1_16b_f32 as fc_chain.hip assumes it, checked on the device:
//   A: lane l supplies (block l/4, row l%4); B: lane l supplies (block l/4, column l%4);
//   D[v] on lane l = (block l/4, row v, column l%4).
// One launch per (a-lane, b-lane) pair with unit inputs; prints "LAYOUT OK" or the first deviations.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mfma4x4_probe tools/ubench/mfma4x4_probe.hip && tools/ubench/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(int la, int lb, float* out) {
  const int lane = threadIdx.x;
  const float a = lane == la ? 1.f : 0.f, b = lane == lb ? 2.f : 0.f;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[lane * 4 + v] = c[v];
}

int main() {
  float* d; hipMalloc(&d, 256 * sizeof(float));
  float h[256];
  int bad = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, la, lb, d);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
          const float want = (la / 4 == lb / 4 && l == (lb / 4) * 4 + lb % 4 && v == la % 4) ? 2.f : 0.f;
          if (h[l * 4 + v] != want && bad++ < 12)
            printf("a-lane %d b-lane %d: D[v=%d] lane %d = %g, expected %g\n", la, lb, v, l, h[l * 4 + v], want);
        }
    }
  printf(bad ? "LAYOUT MISMATCH (%d deviations)\n" : "LAYOUT OK\n", bad);
  return bad ? 1 : 0;
}
