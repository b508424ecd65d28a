// Grouped weight gradients of the VAE's fully-connected layers (encoders.py:63-67, decoders.py:53-55 under
// training.py:157 loss.backward): up to DVAE_FCW_MAX independent problems
//     dw_p[n][k] = sum_m dy_p[m][n] * x_p[m][k],     db_p[n] = sum_m dy_p[m][n]
// in ONE launch.  Each of the six problems of a training step has between 8 and 128 output tiles of 32x32 and a
// contraction as long as the batch: launched one by one (round 1: k_fcw32, 5-6 launches, 176 us of kernel time per
// step at B=1024 with the matrix cores 5 % busy) every launch leaves most of the 256 CUs idle and queues behind the
// previous one; together they are ~400 short-lived workgroups that fill the chip once.
//
// Work decomposition per workgroup (256 threads): one 32x32 tile of one problem, the batch streamed through LDS in
// slabs of KP rows ([kappa][32] images of both operands, skewed by one spare row after every KP/8 rows so that the two
// lane halves of a wave read different bank halves), the contraction of a slab split over the 4 waves x 2 lane
// halves, 4 independent accumulator chains of v_mfma_f32_32x32x2_f32, the next slab's global loads in flight during the
// MFMA phase, fixed-order LDS reduction of the 4 wave partials (deterministic).  Operands whose row length is a multiple of
// 4 floats (and 16-byte aligned) are fetched with 16-byte loads; others (decoder lin1: x = z[B,10]) element-wise.
#include "common.h"

namespace dvae {

struct FcwProb {
  const float* x; const float* dy; float* dw; float* db;
  int M, N, K, tile0, tk, flags;          // tile0: first workgroup of this problem; tk: tiles along K; flags: 1 = dy rows 16-byte, 2 = x rows 16-byte
};
struct FcwTable { FcwProb p[DVAE_FCW_MAX]; int n; };

template <int KP>
__global__ __launch_bounds__(256) void k_fcw_grouped(const FcwTable t) {
  extern __shared__ __attribute__((aligned(16))) float fcg_lds[];
  constexpr int S = KP / 8;                // MFMA steps per wave and lane half per slab
  constexpr int NB = KP / 32;              // 4-column chunks per thread and operand per slab
  constexpr int ROWS = KP + 8;             // skewed rows: one spare row after every S rows
  float* As = fcg_lds;                     // [ROWS][32]  A(i = n, kappa = m)
  float* Bs = fcg_lds + ROWS * 32;         // [ROWS][32]  B(kappa = m, j = k)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int bid = blockIdx.x;

  // problem of this workgroup: wave-uniform selects over the by-value table (no dynamic indexing of kernel arguments)
  const float* x = t.p[0].x; const float* dy = t.p[0].dy; float* dw = t.p[0].dw; float* db = t.p[0].db;
  int M = t.p[0].M, N = t.p[0].N, K = t.p[0].K, tile0 = 0, tk = t.p[0].tk, flags = t.p[0].flags;
#pragma unroll
  for (int q = 1; q < DVAE_FCW_MAX; ++q) {
    if (q < t.n && bid >= t.p[q].tile0) {
      x = t.p[q].x; dy = t.p[q].dy; dw = t.p[q].dw; db = t.p[q].db;
      M = t.p[q].M; N = t.p[q].N; K = t.p[q].K; tile0 = t.p[q].tile0; tk = t.p[q].tk; flags = t.p[q].flags;
    }
  }
  const int tile = bid - tile0;
  const int n0 = (tile / tk) * 32, k0 = (tile % tk) * 32;
  const int j4 = (tid & 7) * 4;
  const bool vecA = flags & 1, vecB = flags & 2;

  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  float rs = 0.f;
  // loads are unconditional (clamped rows / columns); what lies beyond M, N or K is zeroed when the slab is written to
  // LDS, so that the prefetched registers are not touched (no s_waitcnt vmcnt) during the MFMA phase of the previous slab
  f32x4 ra[NB], rb[NB];
  int mload = 0;
  auto load = [&](int m0) {
    mload = m0;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int m = m0 + (tid >> 3) + 32 * p;
      const long mm = m < M ? m : M - 1;
      const float* pa = dy + mm * N;
      const float* pb = x + mm * K;
      if (vecA) {
        ra[p] = *reinterpret_cast<const f32x4*>(pa + (n0 + j4 < N ? n0 + j4 : 0));
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[p][u] = pa[n0 + j4 + u < N ? n0 + j4 + u : 0];
      }
      if (vecB) {
        rb[p] = *reinterpret_cast<const f32x4*>(pb + (k0 + j4 < K ? k0 + j4 : 0));
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) rb[p][u] = pb[k0 + j4 + u < K ? k0 + j4 + u : 0];
      }
    }
  };
  const int kap0 = (2 * wv + h) * S;
  const float* ap = As + (kap0 + 2 * wv + h) * 32 + i;
  const float* bp = Bs + (kap0 + 2 * wv + h) * 32 + i;
  load(0);
  for (int m0 = 0; m0 < M; m0 += KP) {
    if (m0) __syncthreads();                          // the previous slab's operand reads are done
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int kap = (tid >> 3) + 32 * p;
      const bool in = mload + kap < M;
      f32x4 va, vb;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[u] = (in && n0 + j4 + u < N) ? ra[p][u] : 0.f;
        vb[u] = (in && k0 + j4 + u < K) ? rb[p][u] : 0.f;
      }
      *reinterpret_cast<f32x4*>(As + (kap + kap / S) * 32 + j4) = va;
      *reinterpret_cast<f32x4*>(Bs + (kap + kap / S) * 32 + j4) = vb;
    }
    __syncthreads();
    if (m0 + KP < M) load(m0 + KP);
#pragma unroll
    for (int tt = 0; tt < S; tt += 4) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = ap[(tt + u) * 32]; bv[u] = bp[(tt + u) * 32]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[u], 0, 0, 0);
        rs += av[u];
      }
    }
  }
  const f32x16 accs = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  float* red = fcg_lds;                               // [4 waves][16 regs][64 lanes] + [4][32] row sums
  float* rsum = fcg_lds + 4 * 16 * 64;
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wv * 16 + e) * 64 + lane] = accs[e];
  rs += __shfl_xor(rs, 32, 64);
  if (h == 0) rsum[wv * 32 + i] = rs;
  __syncthreads();
  const int col = k0 + i;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = 4 * wv + u;
    const int row = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    const float v = (red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane]) +
                    (red[(2 * 16 + e) * 64 + lane] + red[(3 * 16 + e) * 64 + lane]);
    if (row < N && col < K) dw[(long)row * K + col] = v;
  }
  if (db && k0 == 0 && wv == 0 && h == 0 && n0 + i < N)
    db[n0 + i] = (rsum[i] + rsum[32 + i]) + (rsum[64 + i] + rsum[96 + i]);
}

template <int KP>
static void launch_fcw_grouped_t(const FcwTable& t, int tiles, hipStream_t s) {
  const size_t lds = sizeof(float) * 2 * (KP + 8) * 32;      // >= the reduction image (4224 floats) for KP >= 64
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_fcw_grouped<KP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((k_fcw_grouped<KP>), dim3(tiles), dim3(256), lds, s, t);
}

int launch_linear_wgrad_grouped(const dvae_linear_wgrad_desc* d, int n, hipStream_t s) {
  FcwTable t;
  memset(&t, 0, sizeof(t));
  t.n = n;
  int tiles = 0, maxM = 0;
  for (int q = 0; q < n; ++q) {
    FcwProb& p = t.p[q];
    p.x = d[q].x; p.dy = d[q].dy; p.dw = d[q].dw; p.db = d[q].db;
    p.M = d[q].M; p.N = d[q].N; p.K = d[q].K;
    p.tile0 = tiles;
    p.tk = (p.K + 31) / 32;
    p.flags = ((p.N % 4 == 0 && ((uintptr_t)p.dy & 15) == 0) ? 1 : 0) | ((p.K % 4 == 0 && ((uintptr_t)p.x & 15) == 0) ? 2 : 0);
    tiles += p.tk * ((p.N + 31) / 32);
    if (p.M > maxM) maxM = p.M;
  }
  if (maxM <= 64) launch_fcw_grouped_t<64>(t, tiles, s);
  else if (maxM <= 128) launch_fcw_grouped_t<128>(t, tiles, s);
  else launch_fcw_grouped_t<256>(t, tiles, s);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
