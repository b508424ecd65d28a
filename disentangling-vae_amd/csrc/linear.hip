// fp32 MFMA GEMM for the fully-connected layers (encoders.py:63-67, decoders.py:53-55,
// discriminator.py:51-56) and their backward passes.  One kernel, generic operand strides:
//   C[i][j] = sum_k A(i,k) * B(k,j),  A(i,k) = a[i*sAi + k*sAk],  B(k,j) = b[k*sBk + j*sBj]
//   fwd   : y  = x w^T       A = x  (k fast)   B = w  (k fast)   + bias[j], act
//   dgrad : dx = dy w        A = dy (k fast)   B = w  (j fast)   * act'(x_act)
//   wgrad : dw = dy^T x      A = dy (i fast)   B = x  (j fast)   + db[i] = sum_k A(i,k)
// 64x64 output tile per 256-thread workgroup (2x2 waves of one 32x32 MFMA accumulator each),
// K staged through LDS in slices of 32 with register prefetch of the next slice.
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
