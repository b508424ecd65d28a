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

// ---- large FC GEMM (the 1000-wide discriminator layers): TM x 64 output tile, contraction in slabs of 64 ----
// 4 waves; wave w owns rows (w&1)*TM/2 .. +TM/2 and columns (w>>1)*32 .. +32 of the tile, i.e. TM/64 accumulators of
// 32x32 that share one B fragment.  Operand tiles are double-buffered in LDS ([row][64+4] images, 16-byte operand
// reads through the permuted contraction index kappa = h*32 + t of k_fc32; the j-fast B of the dgrad form as a
// [kappa][64] image whose upper half is skewed by 32 floats so that the two lane halves use disjoint banks).
//   A(i,k) = a[i*lda + k];  B_JFAST ? B(k,j) = b[k*ldb + j] (dgrad)  :  B(k,j) = b[j*ldb + k] (forward)
// Memory pipeline interleaved into the MFMA stream.  PMC on a phase-separated version of this kernel
// (profiles/r01_run41_gemm_pmc.txt): 43 % MFMA-busy; the one wave per SIMD spent half of its cycles outside the
// MFMA phase (address arithmetic, global-load issue, predication, LDS stores, barrier).  A wave is in-order, but
// while an MFMA occupies the pipe (64 cycles) it can issue other work for free.  Slab j lives in register set
// j&1 and LDS image j&1; iteration s
// multiplies slab s while it ISSUES the global loads of slab s+2 between its first MFMAs and WRITES slab s+1
// (loaded one iteration earlier: two-slab prefetch distance, longer than the memory latency) to LDS between its
// later MFMAs -- the order is pinned with sched_group_barrier.  Rows / columns outside the matrices are read from
// clamped addresses (they only feed outputs that are never stored); only A's contraction tail is zeroed.
template <int TM, bool B_JFAST>
__global__ __launch_bounds__(256) void k_gemm_big(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                 float* __restrict__ c, long ldc, int M, int N, int Kc,
                                                 const float* __restrict__ bias, int act,
                                                 const float* __restrict__ mask, int mask_act) {
  extern __shared__ __attribute__((aligned(16))) float fc_lds[];
  constexpr int KS = 64, SA = KS + 4;
  constexpr int A_FLOATS = TM * SA;
  constexpr int B_FLOATS = B_JFAST ? KS * 64 + 32 : 64 * SA;
  constexpr int NLA = TM / 16, NLB = 4, NACC = TM / 64;
  float* As = fc_lds;
  float* Bs = fc_lds + 2 * A_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  int tm, tn;
  {
    const int tiles_n = gridDim.x, T = gridDim.x * gridDim.y;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = L & 7, slot = L >> 3, per = T >> 3, rem = T & 7;
    const int t = xcd * per + (xcd < rem ? xcd : rem) + slot;
    tm = t / tiles_n; tn = t - tm * tiles_n;
  }
  const int m0 = tm * TM, n0 = tn * 64;
  const int r0 = (wv & 1) * (TM / 2), c0 = (wv >> 1) * 32;
  const int trow = tid >> 4, c4 = (tid & 15) * 4;          // this thread's row (+16p) and 16-byte column in a slab

  // invariant source pointers (slab offset added per load)
  const float* pa[NLA];
#pragma unroll
  for (int p = 0; p < NLA; ++p) { const int gi = m0 + trow + 16 * p; pa[p] = a + (long)(gi < M ? gi : M - 1) * lda + c4; }
  const float* pb[NLB];
#pragma unroll
  for (int p = 0; p < NLB; ++p) {
    if (!B_JFAST) { const int gj = n0 + trow + 16 * p; pb[p] = b + (long)(gj < N ? gj : N - 1) * ldb + c4; }
    else { const int gj = n0 + c4; pb[p] = b + (long)(trow + 16 * p) * ldb + (gj < N ? gj : 0); }
  }
  f32x4 ra[2][NLA], rb[2][NLB];
  bool okk[2];
  const int klast = ((Kc + KS - 1) / KS - 1) * KS;

  // one 16-byte global load / LDS store of the slab pipeline (index l: first the A chunks, then the B chunks)
  int kslab[2], kofs[2];
  auto slab_begin = [&](auto PAR, int k0) {
    constexpr int P = decltype(PAR)::value;
    k0 = k0 < klast ? k0 : klast;                          // past the end: re-load the last slab (never used)
    okk[P] = k0 + c4 < Kc;
    kslab[P] = k0;
    kofs[P] = okk[P] ? k0 : -c4;                           // contraction tail: stay inside the row (value zeroed at the store)
  };
  auto load_one = [&](auto PAR, int l) {
    constexpr int P = decltype(PAR)::value;
    if (l < NLA) { ra[P][l] = *reinterpret_cast<const f32x4*>(pa[l] + kofs[P]); return; }
    const int p = l - NLA;
    if (!B_JFAST) rb[P][p] = *reinterpret_cast<const f32x4*>(pb[p] + kofs[P]);
    else { const int kap = kslab[P] + trow + 16 * p; rb[P][p] = *reinterpret_cast<const f32x4*>(pb[p] + (long)(kap < Kc ? kslab[P] : -(trow + 16 * p)) * ldb); }
  };
  auto store_one = [&](auto PAR, int l) {
    constexpr int P = decltype(PAR)::value;
    float* Ab = As + P * A_FLOATS;
    float* Bb = Bs + P * B_FLOATS;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (l < NLA) { *reinterpret_cast<f32x4*>(Ab + (trow + 16 * l) * SA + c4) = okk[P] ? ra[P][l] : zero; return; }
    const int p = l - NLA;
    if (!B_JFAST) *reinterpret_cast<f32x4*>(Bb + (trow + 16 * p) * SA + c4) = rb[P][p];
    else { const int kap = trow + 16 * p; *reinterpret_cast<f32x4*>(Bb + kap * 64 + (kap >= 32 ? 32 : 0) + c4) = rb[P][p]; }
  };
  constexpr int NLD = NLA + NLB;           // 16-byte chunks per thread and slab
  auto load = [&](auto PAR, int k0) {
    slab_begin(PAR, k0);
#pragma unroll
    for (int l = 0; l < NLD; ++l) load_one(PAR, l);
  };
  auto store = [&](auto PAR) {
#pragma unroll
    for (int l = 0; l < NLD; ++l) store_one(PAR, l);
  };

  f32x16 acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

#ifdef DVAE_DEBUG_SWITCHES
  const int abl = act >> 8;                // timing ablations (tools/gemm_one.py): 1 no global loads, 2 no LDS stores, 4 no barriers
#else
  constexpr int abl = 0;
#endif
  act &= 0xff;
  // One slab = 8 steps of 4*NACC MFMAs.  Every step also carries its share of the memory pipeline IN PROGRAM ORDER --
  // steps 0-3 issue the global loads of slab s+2, steps 4-7 the LDS stores of slab s+1 -- and ends with a scheduling
  // fence, so that those instructions are issued in the shadow of this step's MFMAs instead of in phases of their own.
  auto body = [&](auto PAR, int sl) {
    constexpr int P = decltype(PAR)::value;
    using Q = std::integral_constant<int, 1 - P>;
    slab_begin(PAR, (sl + 2) * KS);
    const float* Ab = As + P * A_FLOATS + (r0 + i) * SA + h * 32;
    const float* Bb = Bs + P * B_FLOATS + (B_JFAST ? h * (32 * 64 + 32) + c0 + i : (c0 + i) * SA + h * 32);
    f32x4 av[2][NACC], bv[2];
    auto rd = [&](int q, int slot) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) av[slot][t] = *reinterpret_cast<const f32x4*>(Ab + t * 32 * SA + 4 * q);
      if (B_JFAST) {
#pragma unroll
        for (int u = 0; u < 4; ++u) bv[slot][u] = Bb[(4 * q + u) * 64];
      } else {
        bv[slot] = *reinterpret_cast<const f32x4*>(Bb + 4 * q);
      }
    };
    constexpr int PER = (NLD + 3) / 4;     // chunks per step
    rd(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q + 1 < 8) rd(q + 1, (q + 1) & 1);
      if (q < 4) {
        if (!(abl & 1)) {
#pragma unroll
          for (int l = q * PER; l < (q + 1) * PER && l < NLD; ++l) load_one(PAR, l);
        }
      } else if (!(abl & 2)) {
#pragma unroll
        for (int l = (q - 4) * PER; l < (q - 3) * PER && l < NLD; ++l) store_one(Q{}, l);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NACC; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][t][u], bv[q & 1][u], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(abl & 4)) __syncthreads();
  };

  const int nslab = (Kc + KS - 1) / KS;
  load(std::integral_constant<int, 0>{}, 0);
  store(std::integral_constant<int, 0>{});
  load(std::integral_constant<int, 1>{}, KS);
  __syncthreads();
  for (int sl = 0; sl < nslab; sl += 2) {
    body(std::integral_constant<int, 0>{}, sl);
    if (sl + 1 < nslab) body(std::integral_constant<int, 1>{}, sl + 1);
  }

  const int col = n0 + c0 + i;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + r0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < M && col < N) {
        float v = acc[t][e];
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

template <int TM, bool BJ>
static void launch_gemm_big_t(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
                             const float* bias, int act, const float* mask, int mask_act, hipStream_t s) {
  const size_t lds = sizeof(float) * 2 * (TM * 68 + (BJ ? 64 * 64 + 32 : 64 * 68));
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_gemm_big<TM, BJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  static const int abl = env_int("DVAE_GEMM_ABLATE", 0);   // timing ablation, debug builds only (results invalid)
  hipLaunchKernelGGL((k_gemm_big<TM, BJ>), dim3((N + 63) / 64, (M + TM - 1) / TM), dim3(256), lds, s, a, lda, b, ldb, c, ldc,
                     M, N, Kc, bias, act | (abl << 8), mask, mask_act);
}

// true if the launch was taken: long contractions and wide outputs with 16-byte-aligned rows
template <bool BJ>
static bool try_gemm_big(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
                         const float* bias, int act, const float* mask, int mask_act, hipStream_t s) {
  static const bool off = env_off("DVAE_GEMM_BIG");   // A/B switch, debug builds only
  if (off || Kc < 256 || N < 128 || Kc % 4 || lda % 4 || ldb % 4 || (BJ && N % 4)) return false;
  if ((((uintptr_t)a | (uintptr_t)b) & 15) != 0) return false;
  // TM = 64 -> 2 workgroups per CU (70 KB of LDS each): measured 76.7 vs 69.6 TFLOP/s for TM = 128 at 2048x1000x1000
  static const int force_tm = env_int("DVAE_GEMM_TM", 0);        // A/B switch, debug builds only
  if (force_tm == 128) launch_gemm_big_t<128, BJ>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s);
  else launch_gemm_big_t<64, BJ>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s);
  return true;
}

// small problems (everything in the VAE) go to k_gemm32; large ones (discriminator) to k_gemm
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
  if (try_fc32<false>(x, (long)K, w, (long)K, y, (long)N, M, N, K, b, act, (const float*)nullptr, 0, s) ||
      try_gemm_big<false>(x, (long)K, w, (long)K, y, (long)N, M, N, K, b, act, (const float*)nullptr, 0, s)) {
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
  if (try_fc32<true>(dy, (long)N, w, (long)K, dx, (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, s) ||
      try_gemm_big<true>(dy, (long)N, w, (long)K, dx, (long)K, M, K, N, (const float*)nullptr, 0, x_act, x_act ? act : 0, s)) {
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
  if (try_fcw32(x, dy, dw, db, M, K, N, s)) {
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
