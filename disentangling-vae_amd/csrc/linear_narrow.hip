// The last layer of the FactorVAE discriminator (reference discriminator.py:56 lin6: 1000 -> 2), forward and input gradient:
// with a 2-wide side these are streaming passes over one [M, 1000] activation tensor (8 MB at M = 2048), not GEMMs.  The generic
// MFMA paths of linear.hip took 14.5 us (forward) and 9.8 us (input gradient) per launch at M = 2048; as bandwidth-shaped VALU
// kernels with 16-byte accesses they take 4.1 and 4.9 us (profiles/r04_v9_narrow_ab.txt).
//   forward,  narrow OUTPUT:        y[M,N<=8] = act(x[M,K] w[N,K]^T + b)            k_narrow_out_fwd   (wave per row)
//   dgrad,    narrow contraction:   dx[M,K]   = (dy[M,N<=8] w[N,K]) * act'(x_act)    k_narrow_out_dgrad (elementwise)
// Measured and NOT kept (same file of profiles, r04_v10_narrow_ab.txt): VALU forms of the two narrow weight gradients (lin6:
// 18.9 vs 15.5 us; lin1, 10 -> 1000: 84 vs 30 us) and of lin1's input gradient (16.8 vs 13.5 us) -- they stay on the MFMA
// kernels of linear.hip (split contraction + fixed-order reduction).
#include "common.h"

namespace dvae {

constexpr int NARROW_MAX = 8;        // narrow width covered

__device__ __forceinline__ float narrow_act(float v, int act) {
  if (act == DVAE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DVAE_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
  return v;
}

// ---- y[m][n] = act(sum_k x[m][k] w[n][k] + b[n]), n < N <= 8: a wave per row, lanes stride the contraction in 16-byte chunks
__global__ __launch_bounds__(256) void k_narrow_out_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int M, int K, int N,
                                                        int act) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc[NARROW_MAX];
#pragma unroll
  for (int n = 0; n < NARROW_MAX; ++n) acc[n] = 0.f;
  const float* xr = x + (long)m * K;
  for (int c = lane * 4; c < K; c += 256) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
    for (int n = 0; n < NARROW_MAX; ++n)
      if (n < N) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)n * K + c);
        acc[n] = fmaf(xv[3], wv[3], fmaf(xv[2], wv[2], fmaf(xv[1], wv[1], fmaf(xv[0], wv[0], acc[n]))));
      }
  }
#pragma unroll
  for (int n = 0; n < NARROW_MAX; ++n)
    if (n < N) acc[n] = wave_sum(acc[n]);
  if (lane == 0) {
#pragma unroll
    for (int n = 0; n < NARROW_MAX; ++n)
      if (n < N) y[(long)m * N + n] = narrow_act(acc[n] + (b ? b[n] : 0.f), act);
  }
}

// ---- dx[m][k] = (sum_{n<N} dy[m][n] w[n][k]) * act'(x_act[m][k]): one 16-byte chunk per thread
__global__ __launch_bounds__(256) void k_narrow_out_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                          const float* __restrict__ x_act, int act, float* __restrict__ dx,
                                                          int M, int K, int N) {
  const int kq = K >> 2;
  const long total = (long)M * kq;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / kq), c = (int)(idx - (long)m * kq) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NARROW_MAX; ++n)
      if (n < N) {
        const float g = dy[(long)m * N + n];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long)n * K + c);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaf(g, wv[q], v[q]);
      }
    if (x_act) {
      const f32x4 mv = *reinterpret_cast<const f32x4*>(x_act + (long)m * K + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (act == DVAE_ACT_RELU) v[q] = mv[q] > 0.f ? v[q] : 0.f;
        else if (act == DVAE_ACT_LEAKY02) v[q] = mv[q] > 0.f ? v[q] : 0.2f * v[q];
      }
    }
    *reinterpret_cast<f32x4*>(dx + (long)m * K + c) = v;
  }
}

static inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

// true if the launch was taken
bool try_narrow_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, hipStream_t s) {
  static const bool off = env_off("DVAE_NARROW");     // A/B switch, debug builds only
  if (off || N > NARROW_MAX || K < 256 || K % 4 || !aligned16(x, w)) return false;    // (y: scalar stores)
  hipLaunchKernelGGL(k_narrow_out_fwd, dim3((M + 3) / 4), dim3(256), 0, s, x, w, b, y, M, K, N, act);
  return true;
}

// narrow contraction only (lin6).  The narrow-INPUT input gradient (lin1: dz = g[M,1000] w[1000,10]) stays on the MFMA path of
// linear.hip: a wave-per-row VALU kernel with the weights transposed in LDS measured 16.8 us against 13.5 at M = 2048
// (profiles/r04_v9_narrow_ab.txt: ten 64-lane reductions per row).
bool try_narrow_dgrad(const float* dy, const float* w, const float* x_act, int act, float* dx, int M, int K, int N,
                      hipStream_t s) {
  static const bool off = env_off("DVAE_NARROW");     // A/B switch, debug builds only
  if (off || N > NARROW_MAX || K < 256 || K % 4 || !aligned16(w, x_act, dx)) return false;
  long blocks = ((long)M * (K / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_narrow_out_dgrad, dim3((int)blocks), dim3(256), 0, s, dy, w, x_act, x_act ? act : 0, dx, M, K, N);
  return true;
}

}  // namespace dvae
