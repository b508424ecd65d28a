// fp32 MFMA GEMM for the fully-connected layers (encoders.py:63-67, decoders.py:53-55,
// discriminator.py:51-56) and their backward passes.  One kernel, generic operand strides:
//   C[i][j] = sum_k A(i,k) * B(k,j),  A(i,k) = a[i*sAi + k*sAk],  B(k,j) = b[k*sBk + j*sBj]
//   fwd   : y  = x w^T       A = x  (k fast)   B = w  (k fast)   + bias[j], act
//   dgrad : dx = dy w        A = dy (k fast)   B = w  (j fast)   * act'(x_act)
//   wgrad : dw = dy^T x      A = dy (i fast)   B = x  (j fast)   + db[i] = sum_k A(i,k)
// 64x64 output tile per 256-thread workgroup (2x2 waves of one 32x32 MFMA accumulator each),
// K staged through LDS in slices of 32 with register prefetch of the next slice.
#include <stdint.h>
#include <type_traits>
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

// ---- FC GEMM with the WHOLE contraction resident in LDS (Kc <= 512: every FC layer of the VAE) ----
// A 32x32 output tile per 256-thread workgroup.  Both operand tiles (32 x Kc each) are fetched with
// coalesced 16-byte loads that are ALL in flight at once (one memory latency per launch instead of
// one per 32-deep slice), staged in LDS, and each of the 4 waves multiplies a quarter of the
// contraction.  The contraction index is permuted so that a lane reads its MFMA operands with
// 16-byte LDS loads: step t of lane-half h uses kappa = wave*KP/4 + h*S + t (S = KP/8); the sum is
// order-independent and A and B use the same map.  The 4 partial tiles are summed through LDS in a
// fixed order, every wave finishing 4 of the 16 accumulator rows (bias / activation / mask fused).
//   A(i,k) = a[i*lda + k];  B_JFAST ? B(k,j) = b[k*ldb + j] (dgrad)  :  B(k,j) = b[j*ldb + k] (forward)
// KP = Kc rounded up to a power of two in [32,512] (the tail is zero-filled in LDS).
// SC (scalar staging for rows that are not 16-byte aligned, i.e. the 10-wide latent side of decoder lin1):
//   0 = 16-byte loads for both operands; 1 = 4-byte loads for A and B (forward, Kc = 10);
//   2 = 4-byte loads for B only (dgrad whose OUTPUT is 10 wide: B rows are 10 floats).
template <int KP, bool B_JFAST, int SC>
__global__ __launch_bounds__(256) void k_fc32(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                              float* __restrict__ c, long ldc, int M, int N, int Kc,
                                              const float* __restrict__ bias, int act,
                                              const float* __restrict__ mask, int mask_act) {
  extern __shared__ __attribute__((aligned(16))) float fc_lds[];
  constexpr int SA = KP + 4;               // row stride of a k-contiguous tile (16-byte aligned, conflict-free b128 reads)
  constexpr int KQ = KP / 4;               // 16-byte chunks per row
  constexpr int S = KP / 8;                // MFMA steps per wave (per lane half)
  constexpr int NA = KQ / 8;               // 16-byte loads per thread for a [32][KP] tile
  constexpr int NBJ = KP / 32;             // 16-byte loads per thread for a [KP][32] tile
  float* As = fc_lds;
  float* Bs = fc_lds + 32 * SA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;

  // ---- stage: all global loads first, then the LDS stores
  f32x4 ra[NA];
  f32x4 rb[B_JFAST ? NBJ : NA];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int idx = tid + 256 * p;
    const int row = idx / KQ, c4 = (idx % KQ) * 4;
    const int gi = m0 + row;
    const float* src = a + (long)(gi < M ? gi : M - 1) * lda;
    if (SC == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = src[c4 + u < Kc ? c4 + u : 0];
        ra[p][u] = (gi < M && c4 + u < Kc) ? v : 0.f;
      }
    } else {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (c4 < Kc ? c4 : 0));
      ra[p] = (gi < M && c4 < Kc) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (!B_JFAST) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + 256 * p;
      const int row = idx / KQ, c4 = (idx % KQ) * 4;
      const int gj = n0 + row;
      const float* src = b + (long)(gj < N ? gj : N - 1) * ldb;
      if (SC == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float v = src[c4 + u < Kc ? c4 + u : 0];
          rb[p][u] = (gj < N && c4 + u < Kc) ? v : 0.f;
        }
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (c4 < Kc ? c4 : 0));
        rb[p] = (gj < N && c4 < Kc) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < NBJ; ++p) {
      const int idx = tid + 256 * p;
      const int kap = idx >> 3, j4 = (idx & 7) * 4;
      const int gj = n0 + j4;
      const float* src = b + (long)(kap < Kc ? kap : 0) * ldb;
      if (SC == 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float v = src[gj + u < N ? gj + u : 0];
          rb[p][u] = (kap < Kc && gj + u < N) ? v : 0.f;
        }
      } else {
        const bool ok = kap < Kc && gj < N;                    // N % 4 == 0: a chunk is entirely in or out
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (gj < N ? gj : 0));
        rb[p] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int idx = tid + 256 * p;
    *reinterpret_cast<f32x4*>(As + (idx / KQ) * SA + (idx % KQ) * 4) = ra[p];
  }
  if (!B_JFAST) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int idx = tid + 256 * p;
      *reinterpret_cast<f32x4*>(Bs + (idx / KQ) * SA + (idx % KQ) * 4) = rb[p];
    }
  } else {
    // [kappa][32] rows with one spare row after every S rows: the two lane halves of a wave (kappa
    // apart by S, a multiple of 4) then read rows of different parity = different bank halves
#pragma unroll
    for (int p = 0; p < NBJ; ++p) {
      const int idx = tid + 256 * p;
      const int kap = idx >> 3;
      *reinterpret_cast<f32x4*>(Bs + (kap + kap / S) * 32 + (idx & 7) * 4) = rb[p];
    }
  }
  __syncthreads();

  // ---- multiply: wave wv, lane half h: kappa = (2 wv + h) S + t
  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  const int kap0 = (2 * wv + h) * S;
  const float* ap = As + i * SA + kap0;
  const float* bp = B_JFAST ? Bs + (kap0 + 2 * wv + h) * 32 + i : Bs + i * SA + kap0;
#pragma unroll
  for (int q = 0; q < S / 4; ++q) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 4 * q);
    f32x4 bv;
    if (B_JFAST) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bv[u] = bp[(4 * q + u) * 32];
    } else {
      bv = *reinterpret_cast<const f32x4*>(bp + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[u], 0, 0, 0);
  }
  const f32x16 accs = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();                                   // operand tiles are dead: reuse the space
  float* red = fc_lds;                               // [4 waves][16 regs][64 lanes]
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wv * 16 + e) * 64 + lane] = accs[e];
  __syncthreads();
  const int col = n0 + i;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = 4 * wv + u;
    const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    float v = (red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane]) +
              (red[(2 * 16 + e) * 64 + lane] + red[(3 * 16 + e) * 64 + lane]);
    if (row < M && col < N) {
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

template <int KP, bool BJ, int SC>
static void launch_fc32_t(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
                          const float* bias, int act, const float* mask, int mask_act, hipStream_t s) {
  size_t lds = (size_t)(32 * (KP + 4) + (BJ ? (KP + 8) * 32 : 32 * (KP + 4))) * sizeof(float);
  if (lds < 4 * 16 * 64 * sizeof(float)) lds = 4 * 16 * 64 * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_fc32<KP, BJ, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((k_fc32<KP, BJ, SC>), dim3((N + 31) / 32, (M + 31) / 32), dim3(256), lds, s, a, lda, b, ldb, c, ldc, M, N,
                     Kc, bias, act, mask, mask_act);
}

// true if the launch was taken by k_fc32
template <bool BJ>
static bool try_fc32(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
                     const float* bias, int act, const float* mask, int mask_act, hipStream_t s) {
  static const bool off = env_off("DVAE_GEMM_FC");   // A/B switch, debug builds only
  if (off || Kc > 512) return false;
  if ((long)((M + 63) / 64) * ((N + 63) / 64) >= 512) return false;      // big outputs: the 64x64-tile kernel
  const bool al = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
  if (!BJ && Kc <= 32 && (Kc % 4 || lda % 4 || ldb % 4 || !al)) {          // forward with a short unaligned contraction
    launch_fc32_t<32, false, 1>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s);
    return true;
  }
  if (Kc % 4 || lda % 4 || !al) return false;
  const bool scb = BJ && (N % 4 || ldb % 4);                                // dgrad into a narrow, unaligned output
  if (!BJ && ldb % 4) return false;
#define DVAE_FC_CASE(KP)                                                                                   \
  do {                                                                                                     \
    if (scb) launch_fc32_t<KP, BJ, BJ ? 2 : 0>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s); \
    else launch_fc32_t<KP, BJ, 0>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s);          \
  } while (0)
  if (Kc <= 32) DVAE_FC_CASE(32);
  else if (Kc <= 64) DVAE_FC_CASE(64);
  else if (Kc <= 128) DVAE_FC_CASE(128);
  else if (Kc <= 256) DVAE_FC_CASE(256);
  else DVAE_FC_CASE(512);
#undef DVAE_FC_CASE
  return true;
}

// ---- FC weight gradient with the contraction (the batch) streamed through LDS in slabs of KP rows ----
//   dw[n][k] = sum_m dy[m][n] x[m][k],   db[n] = sum_m dy[m][n]
// Same tile / wave split / contraction permutation as k_fc32, but BOTH operands are contraction-slow
// ([m][32 columns], 16-byte loads along the columns, skewed [kappa][32] LDS image read with conflict-free
// ds_read_b32) and the accumulators persist over M/KP slabs; the next slab's global loads are issued
// before the MFMA phase of the current one.  One launch, no partial tiles, no reduce kernel: the VAE's FC
// weight gradients are latency-bound side-stream work and every launch there delays the big conv weight
// gradients queued behind it.
template <int KP>
__global__ __launch_bounds__(256) void k_fcw32(const float* __restrict__ dy, const float* __restrict__ x,
                                               float* __restrict__ dw, float* __restrict__ db, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float fc_lds[];
  constexpr int S = KP / 8;                // MFMA steps per wave and lane half per slab
  constexpr int NB = KP / 32;              // 16-byte loads per thread and operand per slab
  constexpr int ROWS = KP + 8;             // skewed rows: one spare row after every S rows
  float* As = fc_lds;                      // [ROWS][32]  A(i = n, kappa = m)
  float* Bs = fc_lds + ROWS * 32;          // [ROWS][32]  B(kappa = m, j = k)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int j4 = (tid & 7) * 4;
  const bool okA = n0 + j4 < N, okB = k0 + j4 < K;             // N % 4 == K % 4 == 0
  const float* pa = dy + (okA ? n0 + j4 : 0);
  const float* pb = x + (okB ? k0 + j4 : 0);

  f32x16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
  float rs = 0.f;
  // unconditional loads (clamped rows); rows beyond M are zeroed when the slab is written to LDS, so that the
  // prefetched registers are not touched (no s_waitcnt vmcnt) during the MFMA phase of the previous slab
  f32x4 ra[NB], rb[NB];
  int mload = 0;
  auto load = [&](int m0) {
    mload = m0;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int m = m0 + (tid >> 3) + 32 * p;
      const long mm = m < M ? m : M - 1;
      ra[p] = *reinterpret_cast<const f32x4*>(pa + mm * N);
      rb[p] = *reinterpret_cast<const f32x4*>(pb + mm * K);
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
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(As + (kap + kap / S) * 32 + j4) = (in && okA) ? ra[p] : zero;
      *reinterpret_cast<f32x4*>(Bs + (kap + kap / S) * 32 + j4) = (in && okB) ? rb[p] : zero;
    }
    __syncthreads();
    if (m0 + KP < M) load(m0 + KP);
#pragma unroll
    for (int t = 0; t < S; t += 4) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = ap[(t + u) * 32]; bv[u] = bp[(t + u) * 32]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[u], 0, 0, 0);
        rs += av[u];
      }
    }
  }
  const f32x16 accs = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  float* red = fc_lds;                                // [4 waves][16 regs][64 lanes] + [4][32] row sums
  float* rsum = fc_lds + 4 * 16 * 64;
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
  if (db && blockIdx.x == 0 && wv == 0 && h == 0 && n0 + i < N)
    db[n0 + i] = (rsum[i] + rsum[32 + i]) + (rsum[64 + i] + rsum[96 + i]);
}

static bool try_fcw32(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, hipStream_t s) {
  static const bool off = env_off("DVAE_GEMM_FC");   // A/B switch, debug builds only
  if (off || N % 4 || K % 4 || M > 4096) return false;
  if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return false;
  if ((long)((N + 63) / 64) * ((K + 63) / 64) >= 512) return false;     // big outputs: 64x64 tiles + split contraction
  const dim3 grid((K + 31) / 32, (N + 31) / 32);
  if (M <= 64) {
    constexpr int KP = 64;
    const size_t lds = sizeof(float) * 2 * (KP + 8) * 32;               // operand tiles; the reduction image (4224 floats) fits
    hipLaunchKernelGGL((k_fcw32<KP>), grid, dim3(256), lds, s, dy, x, dw, db, M, N, K);
  } else {
    constexpr int KP = 256;
    const size_t lds = sizeof(float) * 2 * (KP + 8) * 32;
    static DeviceOnce attr;
    if (attr.first()) {
      (void)hipFuncSetAttribute((const void*)k_fcw32<KP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((k_fcw32<KP>), grid, dim3(256), lds, s, dy, x, dw, db, M, N, K);
  }
  return true;
}

// (the large discriminator shapes are taken by the LDS-DMA kernels of gemm_dma.hip before this point)
// small problems (everything in the VAE) go to k_gemm32; the remaining large ones (unaligned rows) to k_gemm
// measured (profiles/r01_run12): k_gemm32 wins only for the forward form with 16-byte loads on both
// operands (11.2 vs 13.3 us at 1024x512x256); the lane-contiguous dgrad / wgrad forms are slower than the
// LDS-staged split-K kernel (16.4 vs 14.0, 22.9 vs 14.6 us) -> forward only unless DVAE_GEMM_SMALL=all
static inline bool use_small(int M, int N, int Kc, bool fwd_vec) {
#ifdef DVAE_DEBUG_SWITCHES
  static const char* mode = getenv("DVAE_GEMM_SMALL");   // A/B switch, debug builds only
#else
  constexpr const char* mode = nullptr;
#endif
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
  // A = x (k contiguous), B(k,j) = w[j*K + k] (k contiguous)
  if (try_narrow_fwd(x, w, b, y, M, K, N, act, s) ||
      try_fc32<false>(x, (long)K, w, (long)K, y, (long)N, M, N, K, b, act, (const float*)nullptr, 0, s) ||
      try_gdma(false, x, (long)K, w, (long)K, y, (long)N, M, N, K, b, act, (const float*)nullptr, 0, s)) {
    DVAE_CHECK_LAUNCH();
    return 0;
  }
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
  if (try_narrow_dgrad(dy, w, x_act, act, dx, M, K, N, s) ||
      try_fc32<true>(dy, (long)N, w, (long)K, dx, (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, s) ||
      try_gdma(true, dy, (long)N, w, (long)K, dx, (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, s)) {
    DVAE_CHECK_LAUNCH();
    return 0;
  }
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
  if (try_gdma_wgrad(x, dy, dw, db, M, K, N, s) || try_fcw32(x, dy, dw, db, M, K, N, s)) {
    DVAE_CHECK_LAUNCH();
    return 0;
  }
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
