// fp32 MFMA GEMM for the fully-connected layers (encoders.py:63-67, decoders.py:53-55,
// discriminator.py:51-56) and their backward passes.  One kernel, generic operand strides:
//   C[i][j] = sum_k A(i,k) * B(k,j),  A(i,k) = a[i*sAi + k*sAk],  B(k,j) = b[k*sBk + j*sBj]
//   fwd   : y  = x w^T       A = x  (k fast)   B = w  (k fast)   + bias[j], act
//   dgrad : dx = dy w        A = dy (k fast)   B = w  (j fast)   * act'(x_act)
//   wgrad : dw = dy^T x      A = dy (i fast)   B = x  (j fast)   + db[i] = sum_k A(i,k)
// 64x64 output tile per 256-thread workgroup (2x2 waves of one 32x32 MFMA accumulator each),
// K staged through LDS in slices of 32 with register prefetch of the next slice.
#include <stdlib.h>
#include "common.h"

namespace dvae {

#define GT 64
#define GK 32
#define GLD (GT + 1)

template <bool A_KFAST, bool B_JFAST>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ a, long sAi, long sAk,
                                              const float* __restrict__ b, long sBk, long sBj,
                                              float* __restrict__ c, long ldc, int M, int N, int K,
                                              const float* __restrict__ bias, int act,
                                              const float* __restrict__ mask, int mask_act,
                                              float* __restrict__ rowsum, int klen) {
  __shared__ float As[GK][GLD];
  __shared__ float Bs[GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wi = wv >> 1, wj = wv & 1;
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  // split-K: slice blockIdx.z covers k in [kbeg, kend) and writes a raw partial tile to c + z*M*N
  const int kbeg = blockIdx.z * klen;
  const int kend = (kbeg + klen < K) ? kbeg + klen : K;
  if (gridDim.z > 1) {
    c += (long)blockIdx.z * M * N;
    if (rowsum) rowsum += (long)blockIdx.z * M;
  }

  // per-thread staging coordinates: 2048 elements per operand tile / 256 threads = 8 each
  float pa[8], pb[8];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = tid + r * 256;
      int ai, ak, bk, bj;
      if (A_KFAST) { ak = e & (GK - 1); ai = e >> 5; } else { ai = e & (GT - 1); ak = e >> 6; }
      if (B_JFAST) { bj = e & (GT - 1); bk = e >> 6; } else { bk = e & (GK - 1); bj = e >> 5; }
      const int gi = m0 + ai, gka = k0 + ak, gkb = k0 + bk, gj = n0 + bj;
      pa[r] = (gi < M && gka < kend) ? a[gi * sAi + gka * sAk] : 0.f;
      pb[r] = (gkb < kend && gj < N) ? b[gkb * sBk + gj * sBj] : 0.f;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = tid + r * 256;
      int ai, ak, bk, bj;
      if (A_KFAST) { ak = e & (GK - 1); ai = e >> 5; } else { ai = e & (GT - 1); ak = e >> 6; }
      if (B_JFAST) { bj = e & (GT - 1); bk = e >> 6; } else { bk = e & (GK - 1); bj = e >> 5; }
      As[ak][ai] = pa[r];
      Bs[bk][bj] = pb[r];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float rs = 0.f;
  const bool do_rowsum = rowsum != nullptr && blockIdx.x == 0;

  load_tiles(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (k0 + GK < kend) load_tiles(k0 + GK);
#pragma unroll
    for (int s = 0; s < GK / 2; ++s) {
      const float av = As[2 * s + h][wi * 32 + i];
      const float bv = Bs[2 * s + h][wj * 32 + i];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (do_rowsum && tid < GT) {
#pragma unroll
      for (int k = 0; k < GK; ++k) rs += As[k][tid];
    }
  }
  if (do_rowsum && tid < GT && m0 + tid < M) rowsum[m0 + tid] = rs;

  const int col = n0 + wj * 32 + i;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row < M && col < N) {
      float v = acc[e];
      if (bias) v += bias[col];
      if (act == DVAE_ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (act == DVAE_ACT_LEAKY02) v = v > 0.f ? v : 0.2f * v;
      const long o = (long)row * ldc + col;
      if (mask) {
        const float mv = mask[o];
        if (mask_act == DVAE_ACT_RELU) v = mv > 0.f ? v : 0.f;
        else if (mask_act == DVAE_ACT_LEAKY02) v = mv > 0.f ? v : 0.2f * v;
      }
      c[o] = v;
    }
  }
}

// ---- small-problem GEMM: 32x32 output tile per workgroup, contraction split over its 4 waves ---
// The VAE's FC layers are tiny (M = batch, K/N <= 512): a 64x64-tile kernel leaves the chip empty
// and a split-K pair of launches is latency-bound.  Here a workgroup owns ONE 32x32 tile, each wave
// takes a quarter of the contraction, loads its operands straight into the MFMA register layout
// (all loads of a 128-deep round in flight at once: one memory latency per round, no LDS staging),
// runs up to 64 MFMAs on 4 independent accumulator chains, and the 4 partial tiles are summed
// through LDS in a fixed order.  Operands contiguous along k are read with 16-byte loads (VA / VB).
template <bool VA, bool VB>
__global__ __launch_bounds__(256) void k_gemm32(const float* __restrict__ a, long sAi, long sAk,
                                                const float* __restrict__ b, long sBk, long sBj,
                                                float* __restrict__ c, long ldc, int M, int N, int Kc,
                                                const float* __restrict__ bias, int act,
                                                const float* __restrict__ mask, int mask_act,
                                                float* __restrict__ rowsum) {
  __shared__ float red[3][16][64];
  __shared__ float rsum[4][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  int per = (Kc + 3) / 4;                  // contraction elements per wave, multiple of 8
  per = (per + 7) & ~7;
  const int k0 = wv * per;
  const int k1 = (k0 + per < Kc) ? k0 + per : Kc;
  const int gi = m0 + i, gj = n0 + i;
  const bool vi = gi < M, vj = gj < N;
  const float* arow = a + (vi ? gi : 0) * sAi;
  const float* bcol = b + (vj ? gj : 0) * sBj;
  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  float rs = 0.f;

  for (int kb = k0; kb < k1; kb += 128) {
    const int rem = k1 - kb;
    const int S = rem >= 128 ? 64 : (rem + 1) / 2;        // this round: lane half h takes k = kb + h*S + s, s < S
    const int kh = kb + h * S;
    float av[64], bv[64];
    if (VA) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int k = kh + 4 * t;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vi && 4 * t < S && k + 3 < k1) v = *reinterpret_cast<const f32x4*>(arow + k);
        else if (vi && 4 * t < S) {
#pragma unroll
          for (int u = 0; u < 4; ++u) if (k + u < k1 && 4 * t + u < S) v[u] = arow[k + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) av[4 * t + u] = v[u];
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 64; ++s2) {
        const int k = kh + s2;
        av[s2] = (vi && s2 < S && k < k1) ? arow[k * sAk] : 0.f;
      }
    }
    if (VB) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int k = kh + 4 * t;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vj && 4 * t < S && k + 3 < k1) v = *reinterpret_cast<const f32x4*>(bcol + k);
        else if (vj && 4 * t < S) {
#pragma unroll
          for (int u = 0; u < 4; ++u) if (k + u < k1 && 4 * t + u < S) v[u] = bcol[k + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[4 * t + u] = v[u];
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 64; ++s2) {
        const int k = kh + s2;
        bv[s2] = (vj && s2 < S && k < k1) ? bcol[k * sBk] : 0.f;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 64; ++s2) {
      if (s2 < S) acc[s2 & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc[s2 & 3], 0, 0, 0);
      rs += av[s2];
    }
  }
  const f32x16 accs = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (wv > 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wv - 1][e][lane] = accs[e];
  }
  if (rowsum) {                                  // row sums of A (bias gradient of the wgrad form)
    rs += __shfl_xor(rs, 32, 64);
    if (h == 0) rsum[wv][i] = rs;
  }
  __syncthreads();
  if (wv == 0) {
    const int col = n0 + i;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < M && col < N) {
        float v = ((accs[e] + red[0][e][lane]) + (red[1][e][lane] + red[2][e][lane]));
        if (bias) v += bias[col];
        if (act == DVAE_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == DVAE_ACT_LEAKY02) v = v > 0.f ? v : 0.2f * v;
        const long o = (long)row * ldc + col;
        if (mask) {
          const float mv = mask[o];
          if (mask_act == DVAE_ACT_RELU) v = mv > 0.f ? v : 0.f;
          else if (mask_act == DVAE_ACT_LEAKY02) v = mv > 0.f ? v : 0.2f * v;
        }
        c[o] = v;
      }
    }
    if (rowsum && blockIdx.x == 0 && h == 0 && m0 + i < M)
      rowsum[m0 + i] = (rsum[0][i] + rsum[1][i]) + (rsum[2][i] + rsum[3][i]);
  }
}

// small problems (everything in the VAE) go to k_gemm32; large ones (discriminator) to k_gemm
// measured (profiles/r01_run12): k_gemm32 wins only for the forward form with 16-byte loads on both
// operands (11.2 vs 13.3 us at 1024x512x256); the lane-contiguous dgrad / wgrad forms are slower than the
// LDS-staged split-K kernel (16.4 vs 14.0, 22.9 vs 14.6 us) -> forward only unless DVAE_GEMM_SMALL=all
static inline bool use_small(int M, int N, int Kc, bool fwd_vec) {
  static const char* mode = getenv("DVAE_GEMM_SMALL");
  const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  if (mode && mode[0] == 'n') return false;
  if (!(tiles64 < 192 && Kc <= 4096)) return false;
  return fwd_vec || (mode && mode[0] == 'a');
}
static inline dim3 grid32(int M, int N) { return dim3((N + 31) / 32, (M + 31) / 32); }

static inline dim3 gemm_grid(int M, int N, int S = 1) { return dim3((N + GT - 1) / GT, (M + GT - 1) / GT, S); }

// c[i] = sum_z ws[z*n + i]; db[i] = sum_z wsb[z*m + i]
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, int S, long n, float* __restrict__ c,
                                                       const float* __restrict__ wsb, int m, float* __restrict__ db) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < S; ++z) v += ws[z * n + i];
    c[i] = v;
  }
  if (db && blockIdx.x == 0) {
    for (int i = threadIdx.x; i < m; i += 256) {
      float v = 0.f;
      for (int z = 0; z < S; ++z) v += wsb[(long)z * m + i];
      db[i] = v;
    }
  }
}

// y = act(sum_z partial_z + bias) * act'(mask): epilogue of the split-K schedule for fwd / dgrad
__global__ __launch_bounds__(256) void k_splitk_epilogue(const float* __restrict__ ws, int S, int M, int N,
                                                         float* __restrict__ c, const float* __restrict__ bias, int act,
                                                         const float* __restrict__ mask, int mask_act) {
  const long n = (long)M * N;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < S; ++z) v += ws[z * n + i];
    if (bias) v += bias[i % N];
    if (act == DVAE_ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (act == DVAE_ACT_LEAKY02) v = v > 0.f ? v : 0.2f * v;
    if (mask) {
      const float mv = mask[i];
      if (mask_act == DVAE_ACT_RELU) v = mv > 0.f ? v : 0.f;
      else if (mask_act == DVAE_ACT_LEAKY02) v = mv > 0.f ? v : 0.2f * v;
    }
    c[i] = v;
  }
}

// number of contraction slices: fill the chip when the output has few 64x64 tiles
static int pick_split(int tiles, int Kc, size_t out_elems, float* ws, size_t ws_floats) {
  int S = 1;
  if (!ws) return 1;
  while (S < 16 && tiles * S < 256 && Kc / (S * 2) >= 64) S *= 2;
  while (S > 1 && (size_t)S * (out_elems + 4096) > ws_floats) S /= 2;
  return S;
}

int launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, float* ws,
                      size_t ws_floats, hipStream_t s) {
  if (use_small(M, N, K, K % 4 == 0)) {
    // A = x (k contiguous), B(k,j) = w[j*K + k] (k contiguous)
    if (K % 4 == 0)
      hipLaunchKernelGGL((k_gemm32<true, true>), grid32(M, N), dim3(256), 0, s, x, (long)K, 1L, w, 1L, (long)K, y,
                         (long)N, M, N, K, b, act, (const float*)nullptr, 0, (float*)nullptr);
    else
      hipLaunchKernelGGL((k_gemm32<false, false>), grid32(M, N), dim3(256), 0, s, x, (long)K, 1L, w, 1L, (long)K, y,
                         (long)N, M, N, K, b, act, (const float*)nullptr, 0, (float*)nullptr);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  const int tiles = ((M + GT - 1) / GT) * ((N + GT - 1) / GT);
  const int S = pick_split(tiles, K, (size_t)M * N, ws, ws_floats);
  if (S == 1) {
    hipLaunchKernelGGL((k_gemm<true, false>), gemm_grid(M, N), dim3(256), 0, s, x, (long)K, 1L, w, 1L, (long)K, y,
                       (long)N, M, N, K, b, act, (const float*)nullptr, 0, (float*)nullptr, K);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  const int klen = ((K + S - 1) / S + GK - 1) / GK * GK;
  hipLaunchKernelGGL((k_gemm<true, false>), gemm_grid(M, N, S), dim3(256), 0, s, x, (long)K, 1L, w, 1L, (long)K, ws,
                     (long)N, M, N, K, (const float*)nullptr, 0, (const float*)nullptr, 0, (float*)nullptr, klen);
  DVAE_CHECK_LAUNCH();
  long n = (long)M * N;
  int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(k_splitk_epilogue, dim3(grid), dim3(256), 0, s, ws, S, M, N, y, b, act, (const float*)nullptr, 0);
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_linear_dgrad(const float* dy, const float* w, const float* x_act, int act, float* dx, int M, int K, int N,
                        float* ws, size_t ws_floats, hipStream_t s) {
  // dx[M,K] = dy[M,N] w[N,K]: contraction length N
  if (use_small(M, K, N, false)) {
    // A = dy (contraction index n contiguous), B(k=n, j) = w[n*K + j] (j contiguous -> lanes)
    if (N % 4 == 0)
      hipLaunchKernelGGL((k_gemm32<true, false>), grid32(M, K), dim3(256), 0, s, dy, (long)N, 1L, w, (long)K, 1L, dx,
                         (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, (float*)nullptr);
    else
      hipLaunchKernelGGL((k_gemm32<false, false>), grid32(M, K), dim3(256), 0, s, dy, (long)N, 1L, w, (long)K, 1L, dx,
                         (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, (float*)nullptr);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  const int tiles = ((M + GT - 1) / GT) * ((K + GT - 1) / GT);
  const int S = pick_split(tiles, N, (size_t)M * K, ws, ws_floats);
  if (S == 1) {
    hipLaunchKernelGGL((k_gemm<true, true>), gemm_grid(M, K), dim3(256), 0, s, dy, (long)N, 1L, w, (long)K, 1L, dx,
                       (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, (float*)nullptr, N);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  const int klen = ((N + S - 1) / S + GK - 1) / GK * GK;
  hipLaunchKernelGGL((k_gemm<true, true>), gemm_grid(M, K, S), dim3(256), 0, s, dy, (long)N, 1L, w, (long)K, 1L, ws,
                     (long)K, M, K, N, (const float*)nullptr, 0, (const float*)nullptr, 0, (float*)nullptr, klen);
  DVAE_CHECK_LAUNCH();
  long n = (long)M * K;
  int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(k_splitk_epilogue, dim3(grid), dim3(256), 0, s, ws, S, M, K, dx, (const float*)nullptr, 0, x_act,
                     x_act ? act : 0);
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_linear_wgrad(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, float* ws,
                        size_t ws_floats, hipStream_t s) {
  // dw[N,K] = dy^T[N,M] x[M,K]: contraction length M (the batch); db[n] = sum_m dy[m][n] = row sums of A.
  // Few output tiles + a long contraction: split the batch over gridDim.z and reduce (fixed order).
  if (use_small(N, K, M, false)) {
    // A(i=n, k=m) = dy[m*N + n], B(k=m, j) = x[m*K + j]: both lane-contiguous, contraction over the batch
    hipLaunchKernelGGL((k_gemm32<false, false>), grid32(N, K), dim3(256), 0, s, dy, 1L, (long)N, x, (long)K, 1L, dw,
                       (long)K, N, K, M, (const float*)nullptr, 0, (const float*)nullptr, 0, db);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  const int tiles = ((N + GT - 1) / GT) * ((K + GT - 1) / GT);
  int S = 1;
  if (ws) {
    while (S < 16 && tiles * S < 256 && M / (S * 2) >= 64) S *= 2;
    while (S > 1 && (size_t)S * ((size_t)N * K + N) > ws_floats) S /= 2;
  }
  if (S == 1) {
    hipLaunchKernelGGL((k_gemm<false, true>), gemm_grid(N, K), dim3(256), 0, s, dy, 1L, (long)N, x, (long)K, 1L, dw,
                       (long)K, N, K, M, (const float*)nullptr, 0, (const float*)nullptr, 0, db, M);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  const int klen = ((M + S - 1) / S + GK - 1) / GK * GK;
  float* wsb = ws + (size_t)S * N * K;
  hipLaunchKernelGGL((k_gemm<false, true>), gemm_grid(N, K, S), dim3(256), 0, s, dy, 1L, (long)N, x, (long)K, 1L, ws,
                     (long)K, N, K, M, (const float*)nullptr, 0, (const float*)nullptr, 0, db ? wsb : (float*)nullptr, klen);
  DVAE_CHECK_LAUNCH();
  const long n = (long)N * K;
  int grid = (int)((n + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(k_splitk_reduce, dim3(grid), dim3(256), 0, s, ws, S, n, dw, wsb, N, db);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
