// Three chained fully-connected layers in ONE launch: the encoder's lin1 -> lin2 -> mu_logvar_gen (encoders.py:81-86), the
// decoder's lin1 -> lin2 -> lin3 (decoders.py:71-73), and their input-gradient chains under training.py:157.
//
// Why: each of these layers is a few microseconds of work (<= 512 wide, batch rows independent), and six of them sit back
// to back on the critical path of the forward pass and again of the backward pass; launched one by one (k_fc32) each pays
// the ~6 us dependent-launch latency, and in the backward pass each one queues again behind the chip-filling
// weight-gradient kernels of the side stream.  Here a workgroup takes 16 batch rows through all three layers:
//   * activations of the tile live in LDS (two ping-pong images), every layer's output is also written to HBM (the backward
//     pass needs it: ReLU masks, weight gradients);
//   * weights are streamed from L2 straight into the MFMA B operand: forward: one 16-byte load per lane = 4 consecutive
//     contraction indices of one output column (w[n][k] is contiguous in k); dgrad: 4-byte loads (the contraction runs over
//     the rows of w), one group of loads in flight ahead of the MFMAs that consume it;
//   * v_mfma_f32_16x16x4_f32, 16 rows x 16 outputs per accumulator, 4 accumulators (64 outputs) per wave share one A read;
//     the contraction index inside a group of 16 is permuted (lane quarter kq takes k = 16g + 4kq + j in MFMA j) so that
//     both operands are 16-byte accesses -- exact fp32 either way (an MFMA is an fmaf chain, the order within a dot
//     product changes with the tiling like in every other kernel here).
// One workgroup per 16 rows: B = 1024 -> 64 workgroups of ~10 us; B = 128 -> 8.
#include "common.h"

namespace dvae {

typedef float f32x4m __attribute__((ext_vector_type(4)));

#define MLP_ROWS 16
#define MLP_MAXW 512                    // widest layer input / output
#define MLP_LD (MLP_MAXW + 4)           // LDS row stride (floats): 516 = odd multiple of 4 -> conflict-free 16-byte row reads
#define MLP_MIDW 256                    // widest FIRST-stage output (it lives in the second, smaller LDS image)
#define MLP_LDB (MLP_MIDW + 4)

struct Mlp3Args {
  const float* w[3]; const float* b[3]; const float* mask[3]; float* out[3];
  int K[3], N[3], act[3];               // layer l: in width K[l], out width N[l]
};

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == DVAE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DVAE_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
  return v;
}

// x[M,K0] -> three layers.  FWD: out_l = act_l(in_l w_l^T + b_l), w_l[N_l][K_l].
//                           DGRAD: out_l = (in_l w_l) * act'(mask_l), w_l[K_l][N_l] (= the forward layer's weight [out][in] with
//                           K_l = forward out width, N_l = forward in width), mask_l[M,N_l] = forward activation entering that layer.
template <bool FWD>
__global__ __launch_bounds__(256) void k_mlp3(const float* __restrict__ x, const Mlp3Args p, int M) {
  __shared__ __attribute__((aligned(16))) float bufA[MLP_ROWS * MLP_LD];
  __shared__ __attribute__((aligned(16))) float bufB[MLP_ROWS * MLP_LDB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * MLP_ROWS;

  // ---- stage the input tile (zero beyond M rows / K0 columns, up to the next multiple of 16 columns)
  {
    const int K0 = p.K[0];
    const int Kp = (K0 + 15) & ~15;
    const bool vec = (K0 % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
    for (int c = tid; c < MLP_ROWS * (Kp / 4); c += 256) {
      const int row = c / (Kp / 4), k4 = (c % (Kp / 4)) * 4;
      f32x4m v = {0.f, 0.f, 0.f, 0.f};
      if (m0 + row < M) {
        const float* src = x + (long)(m0 + row) * K0 + k4;
        if (vec && k4 + 3 < K0) v = *reinterpret_cast<const f32x4m*>(src);
        else {
#pragma unroll
          for (int u = 0; u < 4; ++u) if (k4 + u < K0) v[u] = src[u];
        }
      }
      *reinterpret_cast<f32x4m*>(bufA + row * MLP_LD + k4) = v;
    }
  }
  __syncthreads();

#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float* in = (l & 1) ? bufB : bufA;
    float* on = (l & 1) ? bufA : bufB;
    const int ld_in = (l & 1) ? MLP_LDB : MLP_LD, ld_on = (l & 1) ? MLP_LD : MLP_LDB;
    const int K = p.K[l], N = p.N[l];
    const int Kp = (K + 15) & ~15;
    const int ntiles = (N + 15) >> 4;
    const float* __restrict__ w = p.w[l];
    const bool wvec = FWD && (K % 4 == 0) && ((((uintptr_t)w) & 15) == 0);
    // this wave's output tiles: wv, wv+4, wv+8, ... four at a time
    for (int t0 = wv; t0 < ntiles; t0 += 16) {
      f32x4m acc[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[s] = f32x4m{0.f, 0.f, 0.f, 0.f};
      int ncol[4];
      bool tv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { ncol[s] = (t0 + 4 * s) * 16 + r; tv[s] = (t0 + 4 * s) < ntiles; }
      // B operand of group g (contraction indices 16g + 4kq + j, j = 0..3) for the four tiles
      auto loadB = [&](int g, f32x4m (&B)[4]) {
        const int k = 16 * g + 4 * kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          f32x4m v = {0.f, 0.f, 0.f, 0.f};
          if (tv[s] && ncol[s] < N) {
            if (FWD) {                                   // w[n][k .. k+3]
              const float* src = w + (long)ncol[s] * K + k;
              if (wvec && k + 3 < K) v = *reinterpret_cast<const f32x4m*>(src);
              else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (k + u < K) v[u] = src[u];
              }
            } else {                                     // w[k + j][n]: rows of w are the contraction index
#pragma unroll
              for (int u = 0; u < 4; ++u) if (k + u < K) v[u] = w[(long)(k + u) * N + ncol[s]];
            }
          }
          B[s] = v;
        }
      };
      f32x4m Bc[4], Bn[4];
      const int ngroups = Kp >> 4;
      loadB(0, Bc);
      for (int g = 0; g < ngroups; ++g) {
        if (g + 1 < ngroups) loadB(g + 1, Bn);           // next group's weights in flight during this group's MFMAs
        const f32x4m A = *reinterpret_cast<const f32x4m*>(in + r * ld_in + 16 * g + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], Bc[s][j], acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) Bc[s] = Bn[s];
      }
      // epilogue: D[row = 4 kq + v][col = r] of each tile
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (!tv[s]) continue;
        const int n = ncol[s];
        const float bias = (FWD && p.b[l] && n < N) ? p.b[l][n] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 4 * kq + v;
          float val = acc[s][v];
          const bool ok = (m0 + row < M) && (n < N);
          if (FWD) {
            val = act_apply(val + bias, p.act[l]);
          } else if (p.mask[l] && ok) {
            const float mv = p.mask[l][(long)(m0 + row) * N + n];
            if (p.act[l] == DVAE_ACT_RELU) val = mv > 0.f ? val : 0.f;
            else if (p.act[l] == DVAE_ACT_LEAKY02) val = mv > 0.f ? val : 0.2f * val;
          }
          if (!ok) val = 0.f;
          if (l < 2) on[row * ld_on + n] = val;          // next layer's input (zero in the padding: n < 16 * ntiles <= ld)
          if (ok && p.out[l]) p.out[l][(long)(m0 + row) * N + n] = val;
        }
      }
    }
    if (l < 2) {
      // columns between N and the next multiple of 16 were written as zeros by the tiles above (n >= N -> val = 0)
      __syncthreads();
    }
  }
}

int launch_mlp3(bool fwd, const float* x, const Mlp3Args& a, int M, hipStream_t s) {
  const int grid = (M + MLP_ROWS - 1) / MLP_ROWS;
  if (fwd) hipLaunchKernelGGL(k_mlp3<true>, dim3(grid), dim3(256), 0, s, x, a, M);
  else hipLaunchKernelGGL(k_mlp3<false>, dim3(grid), dim3(256), 0, s, x, a, M);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae

using namespace dvae;

extern "C" {

int dvae_mlp3_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                  const float* b3, float* y1, float* y2, float* y3, int M, int K0, int N1, int N2, int N3, int act1, int act2,
                  int act3, void* stream) {
  DVAE_CHECK_ARG(x && w1 && w2 && w3 && y3 && M > 0 && K0 > 0 && N1 > 0 && N2 > 0 && N3 > 0);
  DVAE_CHECK_ARG(K0 <= MLP_MAXW && N1 <= MLP_MIDW && N2 <= MLP_MAXW && N3 <= MLP_MAXW);
  Mlp3Args a;
  memset(&a, 0, sizeof(a));
  a.w[0] = w1; a.w[1] = w2; a.w[2] = w3; a.b[0] = b1; a.b[1] = b2; a.b[2] = b3;
  a.out[0] = y1; a.out[1] = y2; a.out[2] = y3;
  a.K[0] = K0; a.N[0] = N1; a.K[1] = N1; a.N[1] = N2; a.K[2] = N2; a.N[2] = N3;
  a.act[0] = act1; a.act[1] = act2; a.act[2] = act3;
  return launch_mlp3(true, x, a, M, (hipStream_t)stream);
}

int dvae_mlp3_dgrad(const float* dy, const float* w3, const float* w2, const float* w1, const float* act2, const float* act1,
                    const float* act0, float* g2, float* g1, float* dx, int M, int K0, int N1, int N2, int N3, int mask_act,
                    void* stream) {
  DVAE_CHECK_ARG(dy && w1 && w2 && w3 && dx && M > 0 && K0 > 0 && N1 > 0 && N2 > 0 && N3 > 0);
  DVAE_CHECK_ARG(K0 <= MLP_MAXW && N1 <= MLP_MAXW && N2 <= MLP_MIDW && N3 <= MLP_MAXW);
  DVAE_CHECK_ARG(mask_act == DVAE_ACT_NONE || mask_act == DVAE_ACT_RELU || mask_act == DVAE_ACT_LEAKY02);
  // stage 0: dy[M,N3] w3[N3,N2] -> g2[M,N2] (mask act2); stage 1: w2[N2,N1] -> g1 (mask act1); stage 2: w1[N1,K0] -> dx (mask act0)
  Mlp3Args a;
  memset(&a, 0, sizeof(a));
  a.w[0] = w3; a.w[1] = w2; a.w[2] = w1;
  a.mask[0] = act2; a.mask[1] = act1; a.mask[2] = act0;
  a.out[0] = g2; a.out[1] = g1; a.out[2] = dx;
  a.K[0] = N3; a.N[0] = N2; a.K[1] = N2; a.N[1] = N1; a.K[2] = N1; a.N[2] = K0;
  a.act[0] = act2 ? mask_act : DVAE_ACT_NONE; a.act[1] = act1 ? mask_act : DVAE_ACT_NONE; a.act[2] = act0 ? mask_act : DVAE_ACT_NONE;
  return launch_mlp3(false, dy, a, M, (hipStream_t)stream);
}

}  // extern "C"
