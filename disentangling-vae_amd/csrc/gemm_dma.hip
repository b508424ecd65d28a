// fp32 MFMA GEMMs for the 1000-wide FactorVAE discriminator layers (reference discriminator.py:41-70 forward, its autograd
// backward under losses.py:303-306) -- the "large" shapes M x 1000 x 1000 of dvae_linear_fwd / _dgrad / _wgrad.
//
// Structure (the scheme of the 32-channel conv kernels, conv_down_dma.hip, applied to a plain GEMM):
//   * 512-thread workgroups = 4 MFMA waves (one per SIMD) + 4 loader waves.  Operand tiles travel L2 -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: 1 KB per wave instruction, no registers, no ds_write) in a ring of NS = D + 1 stages of
//     KS contraction steps, D slabs in flight.  The loaders execute ~35 instructions per slab; the MFMA waves nothing but
//     MFMAs and LDS operand reads (a transfer issued from inside the MFMA stream stalls it for ~90 cycles: measured,
//     profiles/r04_v2_gdma_abl.txt: 50.8 -> 41.8 us at 2048 x 1000 x 1000 with the transfers ablated).  Completion is
//     tracked with counted s_waitcnt vmcnt(N) in the loaders + one bare s_barrier per slab for everybody;
//   * the LDS side of a transfer is lane-linear, so the bank swizzle of the k-contiguous tiles and every boundary
//     (contraction tail, rows / columns outside the matrices) is applied on the SOURCE side: a lane fetches the global
//     16-byte chunk that belongs at its LDS position, or 16 bytes of zeros;
//   * k-contiguous operands (x, dy, and w in the forward form) are read with ONE ds_read_b128 per four MFMAs through a
//     permuted contraction index (lane half h of round r takes k = 8 r + 4 h + u for MFMA u: the sum is order-free and
//     both operands use the same map); the contraction-slow w of the input-gradient form lands as natural [k][64] rows
//     and is read with conflict-free ds_read_b32;
//   * the barrier at the end of slab s certifies slab s + 2, so the first operand reads of slab s + 1 are issued BEFORE
//     it and no ds_read latency is exposed behind a barrier;
//   * the product is computed transposed (MFMA A operand = weights, B operand = batch rows): an accumulator lane holds 4
//     consecutive output columns of one row, the epilogue is 16-byte bias / mask loads and 16-byte stores;
//   * accumulators persistent over the whole contraction, no split-K, no partial tiles in HBM, no reduction launch:
//     results are deterministic (fixed summation order per output).
// Tiles: TM x 64 (TM = 128: wave = 64 x 32 = two accumulators sharing one weight fragment; TM = 64: wave = 32 x 32 on two
// accumulator chains).  The weight gradient (k_gdma_wg) uses 64 x 64 tiles over the whole batch with the BATCH split over
// the four MFMA waves (each owns the whole tile = 4 accumulators, so one ds_read_b64 per operand feeds 4 MFMAs) and a
// fixed-order sum of the four partial tiles through LDS at the end.  Workgroup -> tile map: the 32 workgroups that share
// an XCD (blockIdx % 8) own a compact 4 x 8 block of tiles, i.e. ~4 MB of operands = its L2.
#include <type_traits>
#include "common.h"

// dma16 names m0 in its clobber list on purpose (the instruction takes its LDS base from m0)
#pragma clang diagnostic ignored "-Winline-asm"

namespace dvae {

__device__ __attribute__((aligned(16))) float k_gdma_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// one LDS-DMA transfer: lane l of the wave moves 16 bytes from its own global address to LDS byte lds_addr + 16 l
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void gdma_barrier() { asm volatile("s_barrier" ::: "memory"); }

// workgroup -> (tile row, tile column): linear tile order = (column block of 8, row, column within the block); the
// workgroups of XCD x (blockIdx % 8 == x) take the x-th eighth of that order
__device__ __forceinline__ void gdma_tile(int tiles_m, int tiles_n, int* tm, int* tn) {
  const int T = tiles_m * tiles_n;
  const int L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3, per = T >> 3, rem = T & 7;
  const int t = xcd * per + (xcd < rem ? xcd : rem) + slot;
  const int full = (tiles_n >> 3) * tiles_m * 8;            // tiles in complete 8-column blocks
  if (t < full) {
    const int nb = t / (tiles_m * 8), r = t - nb * tiles_m * 8;
    *tm = r >> 3; *tn = nb * 8 + (r & 7);
  } else {
    const int w = tiles_n & 7, r = t - full;                  // the last, narrower column block
    *tm = r / w; *tn = (tiles_n & ~7) + r % w;
  }
}

template <int TM_, int TN_, int KS_, int D_, int NWK_>
struct GdmaGeo {
  static constexpr int TM = TM_, TN = TN_, KS = KS_, D = D_, NS = D_ + 1, NWK = NWK_;
  static constexpr int CPR = KS / 4;                     // 16-byte chunks per row of a k-contiguous tile
  static constexpr int RPB = 64 / CPR;                   // its rows per 1 KB transfer
  static constexpr int SWD = 16 / CPR;                   // rows that share a swizzle value
  static constexpr int NBA = TM * KS / 256;              // transfers per slab: A tile
  static constexpr int NBB = TN * KS / 256;              //                     B tile (either orientation)
  static constexpr int P = (NBA + NBB) / 4;              // per loader wave
  static constexpr int STAGE_FLOATS = (TM + TN) * KS;
  static constexpr int NR = KS / 8;                      // rounds (4 MFMAs per accumulator) per slab
  // the four MFMA waves: NWN column blocks of 32 x NWM row blocks x NWK shares of every slab's rounds
  static constexpr int NWN = TN / 32, NWM = 4 / (NWN * NWK);
  static constexpr int NACC = TM / NWM / 32;             // 32-row accumulators per wave
  static constexpr int NRW = NR / NWK;                   // rounds per wave and slab
  static constexpr int JCH = TN / 4, JRPB = 256 / TN;    // contraction-slow B tile: chunks per row, rows per transfer
  static_assert((NBA + NBB) % 4 == 0 && NRW % 2 == 0 && NRW * NWK == NR && (D - 2) * P < 64 && D >= 2, "geometry");
  static_assert(NWN * NWM * NWK == 4 && NACC * NWM * 32 == TM && (TN == 32 || TN == 64), "wave layout");
  static_assert(NWK == 1 || NS * STAGE_FLOATS >= NWK * 16 * 64, "the ring holds the partial tiles of the final sum");
  __host__ __device__ static constexpr int swz(int r) { return (r / SWD) & (CPR - 1); }
};

// C[M,N] = A[M,Kc] * B  (+ bias, activation, mask); A(i,k) = a[i*lda + k];
//   B_JFAST = false: B(k,j) = b[j*ldb + k]  (forward:  y  = act(x w^T + bias))
//   B_JFAST = true : B(k,j) = b[k*ldb + j]  (dgrad:    dx = (dy w) * act'(mask))
// Requirements (checked by the launcher): Kc % 4 == lda % 4 == ldb % 4 == 0, N % 4 == 0 for B_JFAST, 16-byte aligned bases.
// Tile TM x TN; NWK > 1: the MFMA waves share one 32 x 32 output block and split every slab's rounds among themselves
// (small batches: 32 x 32 tiles give every CU a workgroup at M = 256), partial blocks summed through LDS in wave order.
// ABL: timing ablations, compile-time so that the instruction schedule of the surviving parts is the shipped one
// (DVAE_GDMA_ABLATE, debug builds only): 1 no transfers, 2 no LDS operand reads (results invalid); 8 loader waves NOT at
// s_setprio 2 (results valid).  Measured and dropped (profiles/r04_v4_variants.txt, r04_v5_variants.txt): ring depth D = 3 / 5
// (equal), KS = 64 with D = 2 at TM = 128 (+12 %), 8 loader waves (equal), s_setprio 1 on the MFMA waves (equal), plain
// row-major tile order (+1 %).
template <int TM, int TN, int KS, int D, int NWK, bool B_JFAST, int ABL = 0>
__global__ __launch_bounds__(512) void k_gdma(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                              float* __restrict__ c, long ldc, int M, int N, int Kc,
                                              const float* __restrict__ bias, int act,
                                              const float* __restrict__ mask, int mask_act, int tiles_m, int tiles_n,
                                              int vec_ok) {
  using G = GdmaGeo<TM, TN, KS, D, NWK>;
  constexpr int abl = ABL;
  extern __shared__ __attribute__((aligned(16))) float gd_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm, tn;
  gdma_tile(tiles_m, tiles_n, &tm, &tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int nslab = (Kc + KS - 1) / KS;

  if (wv >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wv - 4;
    const int klast = (nslab - 1) * KS;
    if (!(abl & 8)) __builtin_amdgcn_s_setprio(2);       // measured: 47.2 -> 45.9 us at 2048 x 1000 x 1000 (profiles/r04_v4_variants.txt)
    // this wave's transfers of a slab: source address per lane, its per-slab step, validity in the last slab
    const char* src[G::P];
    unsigned step[G::P];
    unsigned tail_ok = 0;
#pragma unroll
    for (int p = 0; p < G::P; ++p) {
      const int g = lw + 4 * p;                            // transfer (1 KB block) of the stage
      if (g < G::NBA || !B_JFAST) {
        const bool isA = g < G::NBA;
        const int blk = isA ? g : g - G::NBA;
        const int r = blk * G::RPB + lane / G::CPR, s = lane % G::CPR;
        const int q = s ^ G::swz(r);                       // logical chunk that lives at slot s of row r
        const int gr = isA ? (m0 + r < M ? m0 + r : M - 1) : (n0 + r < N ? n0 + r : N - 1);
        src[p] = reinterpret_cast<const char*>((isA ? a + (long)gr * lda : b + (long)gr * ldb) + 4 * q);
        step[p] = KS * 4u;
        if (klast + 4 * q < Kc) tail_ok |= 1u << p;
      } else {
        const int blk = g - G::NBA;
        const int kk = blk * G::JRPB + lane / G::JCH, col = n0 + 4 * (lane % G::JCH);
        const bool col_ok = col < N;
        src[p] = col_ok ? reinterpret_cast<const char*>(b + (long)kk * ldb + col) : reinterpret_cast<const char*>(k_gdma_zero16);
        step[p] = col_ok ? (unsigned)(KS * ldb * 4) : 0u;
        if (klast + kk < Kc) tail_ok |= 1u << p;
      }
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)gd_lds + lw * 1024u;
    auto issue = [&](int slab, int buf) {                  // slab < nslab
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * (G::STAGE_FLOATS * 4u));
      if (abl & 1) return;
      if (slab < nslab - 1) {
#pragma unroll
        for (int p = 0; p < G::P; ++p) {
          dma16(src[p], dst + p * 4096u);
          src[p] += step[p];
        }
      } else {                                             // last slab: chunks beyond the contraction come from the zero block
#pragma unroll
        for (int p = 0; p < G::P; ++p)
          dma16((tail_ok >> p) & 1u ? src[p] : reinterpret_cast<const char*>(k_gdma_zero16), dst + p * 4096u);
      }
    };
    // D slabs in flight; slabs 0 and 1 landed before the first barrier (slab 1 feeds the early reads of iteration 0)
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nslab) issue(s, s);
    if (nslab >= D) wait_vm<(D - 2) * G::P>();
    else wait_vm<0>();
    gdma_barrier();
    int isb = D;                                           // stage that receives slab s + D (= the stage slab s - 1 just left)
    for (int s = 0; s < nslab; ++s) {
      if (s + D < nslab) {
        issue(s + D, isb);
        wait_vm<(D - 2) * G::P>();                         // all but the newest D - 2 slabs: slab s + 2 has landed
      } else {
        wait_vm<0>();
      }
      gdma_barrier();
      isb = isb + 1 == G::NS ? 0 : isb + 1;
    }
    if (NWK > 1) gdma_barrier();                           // the barrier of the final sum
    return;
  }

  // -------------------------------------------------------------------------------------------------- MFMA waves
  const int i = lane & 31, h = lane >> 5;
  const int wj = wv % G::NWN, wi = (wv / G::NWN) % G::NWM, wk = wv / (G::NWN * G::NWM);
  // operand read offsets (floats, relative to a stage): this wave's rounds of a slab are wk NRW .. wk NRW + NRW - 1
  int offr[G::NRW];                                        // k-contiguous tiles: row i, round -> swizzled chunk of this lane half
#pragma unroll
  for (int j = 0; j < G::NRW; ++j) offr[j] = i * KS + (((2 * (wk * G::NRW + j) + h) ^ G::swz(i)) << 2);
  const int a_base = wi * (TM / G::NWM) * KS;
  const int b_base = TM * KS + (B_JFAST ? (8 * wk * G::NRW + 4 * h) * TN + wj * 32 + i : wj * 32 * KS);

  constexpr int NCH = G::NACC == 1 ? 2 : 1;                // accumulator chains per 32x32 output block
  f32x16 acc[G::NACC][NCH];
#pragma unroll
  for (int t = 0; t < G::NACC; ++t)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][ch][e] = 0.f;

  f32x4 av[2][G::NACC], bv[2];
  if (abl & 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bv[q] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
#pragma unroll
      for (int t = 0; t < G::NACC; ++t) av[q][t] = f32x4{1.f, 2.f + lane, 3.f, 4.f};
    }
  }
  auto rd = [&](const float* st, int j, int slot) {        // round j of this wave's share of a slab
    if (abl & 2) return;
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) av[slot][t] = *reinterpret_cast<const f32x4*>(st + a_base + t * 32 * KS + offr[j]);
    if (B_JFAST) {
#pragma unroll
      for (int u = 0; u < 4; ++u) bv[slot][u] = st[b_base + (8 * j + u) * TN];
    } else {
      bv[slot] = *reinterpret_cast<const f32x4*>(st + b_base + offr[j]);
    }
  };

  gdma_barrier();                                          // slabs 0 and 1 are in stages 0 and 1
  rd(gd_lds, 0, 0);
  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    const float* st = gd_lds + cur * G::STAGE_FLOATS;
    const int nxt = cur + 1 == G::NS ? 0 : cur + 1;
    const float* stn = gd_lds + nxt * G::STAGE_FLOATS;
#pragma unroll
    for (int j = 0; j < G::NRW; ++j) {
      const int sl = j & 1;
      if (j + 1 < G::NRW) rd(st, j + 1, sl ^ 1);
      else rd(stn, 0, sl ^ 1);                             // slab s + 1 was certified by the previous barrier (unused after the last slab)
      // transposed product: MFMA A operand = the weight fragment (rows of D = output columns), B operand = batch rows
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < G::NACC; ++t)
          acc[t][NCH == 2 ? (u & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[sl][u], av[sl][t][u], acc[t][NCH == 2 ? (u & 1) : 0], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, G::NACC + (B_JFAST ? 4 : 1), 0);   // the next round's DS reads first
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * G::NACC, 0);                   // then this round's MFMAs
    }
    gdma_barrier();                                        // slab s + 2 has landed; stage `cur` is released
    cur = nxt;
  }

  // ---- epilogue: lane (i, h) of accumulator t holds row m0 + wi TM/NWM + 32 t + i, columns n0 + 32 wj + 8 g + 4 h + (0..3)
  // in registers 4 g .. 4 g + 3: bias, activation, mask (act'(x_act) of the producing layer) on 16-byte accesses
  const int colb = n0 + wj * 32 + 4 * h;
  auto finish = [&](int row, int g, f32x4 v) {
    const int col = colb + 8 * g;
    if (row >= M || col >= N) return;
    const long o = (long)row * ldc + col;
    f32x4 bb = {0.f, 0.f, 0.f, 0.f}, mv = {1.f, 1.f, 1.f, 1.f};
    if (vec_ok) {
      if (bias) bb = *reinterpret_cast<const f32x4*>(bias + col);
      if (mask) mv = *reinterpret_cast<const f32x4*>(mask + o);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (col + q < N) {
          if (bias) bb[q] = bias[col + q];
          if (mask) mv[q] = mask[o + q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = v[q] + bb[q];
      if (act == DVAE_ACT_RELU) x = x > 0.f ? x : 0.f;
      else if (act == DVAE_ACT_LEAKY02) x = x > 0.f ? x : 0.2f * x;
      if (mask) {
        if (mask_act == DVAE_ACT_RELU) x = mv[q] > 0.f ? x : 0.f;
        else if (mask_act == DVAE_ACT_LEAKY02) x = mv[q] > 0.f ? x : 0.2f * x;
      }
      v[q] = x;
    }
    if (vec_ok) {
      *reinterpret_cast<f32x4*>(c + o) = v;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (col + q < N) c[o + q] = v[q];
    }
  };
  if (NWK == 1) {
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) {
      f32x16 r = acc[t][0];
      if (NCH == 2) r += acc[t][NCH - 1];
      const int row = m0 + wi * (TM / G::NWM) + t * 32 + i;
#pragma unroll
      for (int g = 0; g < 4; ++g) finish(row, g, f32x4{r[4 * g], r[4 * g + 1], r[4 * g + 2], r[4 * g + 3]});
    }
  } else {
    // the NWK waves hold partial sums of the SAME 32 x 32 block: red[wave][register][lane], summed in wave order; wave w
    // finishes registers 4 w .. 4 w + 3 (the ring is free: every transfer has landed and been consumed)
    static_assert(NWK == 1 || (NWK == 4 && G::NACC == 1), "the final sum assumes four waves on one block");
    float* red = gd_lds;
    const f32x16 r = acc[0][0] + acc[0][NCH - 1];
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wk * 16 + e) * 64 + lane] = r[e];
    gdma_barrier();
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = (4 * wk + q) * 64 + lane;
      v[q] = (red[o] + red[o + 16 * 64]) + (red[o + 2 * 16 * 64] + red[o + 3 * 16 * 64]);
    }
    finish(m0 + i, wk, v);
  }
}

// ---- weight gradient: dw[N,K] = dy^T x over the whole batch, db[N] = column sums of dy --------------------------------
// 64 x 64 output tile per workgroup, the batch in slabs of KS rows; both operand tiles are natural [m][64] rows (dy columns
// n0.., x columns k0..).  MFMA wave w owns rows 16 w .. 16 w + 15 of every slab and the WHOLE tile: lane i of an operand
// reads the column pair (2 i, 2 i + 1) of a row with one ds_read_b64, the four products (q, q') of the two pairs go to four
// accumulators (output row n0 + 2 i + q, column k0 + 2 j + q').  After the last slab the four partial tiles are summed
// through LDS in the fixed order (w0 + w1) + (w2 + w3).
template <int KS, int D>
__global__ __launch_bounds__(512) void k_gdma_wg(const float* __restrict__ dy, const float* __restrict__ x,
                                                 float* __restrict__ dw, float* __restrict__ db, int M, int N, int K,
                                                 int tiles_n, int tiles_k) {
  constexpr int NS = D + 1;
  constexpr int NB = KS / 4;                               // transfers per operand tile and slab ([KS][64] floats)
  constexpr int P = 2 * NB / 4;
  constexpr int STAGE_FLOATS = 2 * KS * 64;
  constexpr int NPAIR = KS / 8;                            // contraction pairs per MFMA wave and slab
  static_assert((D - 2) * P < 64 && D >= 2 && KS % 32 == 0 && NPAIR % 2 == 0, "geometry");
  static_assert(NS * STAGE_FLOATS >= 4 * 4 * 16 * 64 + 256, "the ring holds the four partial tiles of the final sum");
  extern __shared__ __attribute__((aligned(16))) float gd_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tnn, tk;
  gdma_tile(tiles_n, tiles_k, &tnn, &tk);
  const int n0 = tnn * 64, k0 = tk * 64;
  const int nslab = (M + KS - 1) / KS;

  if (wv >= 4) {
    // ------------------------------------------------------------------------------------------------ loader waves
    const int lw = wv - 4;
    const int mlast = (nslab - 1) * KS;
    __builtin_amdgcn_s_setprio(2);
    const char* src[P];
    unsigned step[P];
    unsigned tail_ok = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int g = lw + 4 * p;
      const bool isA = g < NB;
      const int blk = isA ? g : g - NB;
      const int mm = blk * 4 + (lane >> 4);
      const int col = (isA ? n0 : k0) + 4 * (lane & 15);
      const bool col_ok = col < (isA ? N : K);
      const float* base = isA ? dy + (long)mm * N + col : x + (long)mm * K + col;
      src[p] = col_ok ? reinterpret_cast<const char*>(base) : reinterpret_cast<const char*>(k_gdma_zero16);
      step[p] = col_ok ? (unsigned)(KS * (isA ? N : K) * 4) : 0u;
      if (mlast + mm < M) tail_ok |= 1u << p;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)gd_lds + lw * 1024u;
    auto issue = [&](int slab, int buf) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * (STAGE_FLOATS * 4u));
      if (slab < nslab - 1) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          dma16(src[p], dst + p * 4096u);
          src[p] += step[p];
        }
      } else {
#pragma unroll
        for (int p = 0; p < P; ++p)
          dma16((tail_ok >> p) & 1u ? src[p] : reinterpret_cast<const char*>(k_gdma_zero16), dst + p * 4096u);
      }
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nslab) issue(s, s);
    if (nslab >= D) wait_vm<(D - 2) * P>();
    else wait_vm<0>();
    gdma_barrier();
    int isb = D;
    for (int s = 0; s < nslab; ++s) {
      if (s + D < nslab) {
        issue(s + D, isb);
        wait_vm<(D - 2) * P>();
      } else {
        wait_vm<0>();
      }
      gdma_barrier();
      isb = isb + 1 == NS ? 0 : isb + 1;
    }
    gdma_barrier();                                        // the barrier of the final sum
    return;
  }

  // -------------------------------------------------------------------------------------------------- MFMA waves
  const int i = lane & 31, h = lane >> 5;
  const int a_off = (16 * wv + h) * 64 + 2 * i;            // + 2 t * 64 for pair t
  const int b_off = KS * 64 + a_off;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x16 acc[2][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[q >> 1][q & 1][e] = 0.f;
  float rs[2] = {0.f, 0.f};
  f32x2 av[2], bv[2];
  auto rd = [&](const float* st, int t, int slot) {
    av[slot] = *reinterpret_cast<const f32x2*>(st + a_off + t * 128);
    bv[slot] = *reinterpret_cast<const f32x2*>(st + b_off + t * 128);
  };
  gdma_barrier();
  rd(gd_lds, 0, 0);
  int cur = 0;
  for (int s = 0; s < nslab; ++s) {
    const float* st = gd_lds + cur * STAGE_FLOATS;
    const int nxt = cur + 1 == NS ? 0 : cur + 1;
    const float* stn = gd_lds + nxt * STAGE_FLOATS;
#pragma unroll
    for (int t = 0; t < NPAIR; ++t) {
      const int sl = t & 1;
      if (t + 1 < NPAIR) rd(st, t + 1, sl ^ 1);
      else rd(stn, 0, sl ^ 1);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
          acc[q][q2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sl][q], bv[sl][q2], acc[q][q2], 0, 0, 0);
        rs[q] += av[sl][q];
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    gdma_barrier();
    cur = nxt;
  }

  // ---- fixed-order sum of the four waves' partial tiles: part[w][acc][e][lane]; wave w finishes registers 4 w .. 4 w + 3
  float* part = gd_lds;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) part[((wv * 4 + q) * 16 + e) * 64 + lane] = acc[q >> 1][q & 1][e];
  float* rsp = gd_lds + 4 * 4 * 16 * 64;                   // [w][q][32]: bias-gradient partials (the two lane halves added)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float v = rs[q] + __shfl_xor(rs[q], 32, 64);
    if (h == 0) rsp[(wv * 2 + q) * 32 + i] = v;
  }
  gdma_barrier();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = 4 * wv + u;
    const int ie = (e & 3) + 8 * (e >> 2) + 4 * h;         // accumulator row of register e
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x2 v;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int o = ((q * 2 + q2) * 16 + e) * 64 + lane;
        v[q2] = (part[o] + part[o + 4 * 16 * 64]) + (part[o + 2 * 4 * 16 * 64] + part[o + 3 * 4 * 16 * 64]);
      }
      const int row = n0 + 2 * ie + q, col = k0 + 2 * i;
      if (row < N && col < K) *reinterpret_cast<f32x2*>(dw + (long)row * K + col) = v;   // K % 4 == 0: the pair is inside the row
    }
  }
  if (db && tk == 0 && wv == 0 && h == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + 2 * i + q;
      if (n < N) db[n] = (rsp[(0 * 2 + q) * 32 + i] + rsp[(1 * 2 + q) * 32 + i]) + (rsp[(2 * 2 + q) * 32 + i] + rsp[(3 * 2 + q) * 32 + i]);
    }
  }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
template <int TM, int TN, int KS, int D, int NWK, bool BJ, int ABL = 0>
static void launch_gdma_t(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
                          const float* bias, int act, const float* mask, int mask_act, hipStream_t s) {
  using G = GdmaGeo<TM, TN, KS, D, NWK>;
  const size_t lds = (size_t)G::NS * G::STAGE_FLOATS * sizeof(float);
  static DeviceOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute((const void*)k_gdma<TM, TN, KS, D, NWK, BJ, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
  const int vec_ok = N % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)c | (uintptr_t)bias | (uintptr_t)mask) & 15) == 0;
  hipLaunchKernelGGL((k_gdma<TM, TN, KS, D, NWK, BJ, ABL>), dim3(tiles_m * tiles_n), dim3(512), lds, s, a, lda, b, ldb, c, ldc,
                     M, N, Kc, bias, act, mask, mask_act, tiles_m, tiles_n, vec_ok);
}

// true if the launch was taken: long contractions and wide outputs with 16-byte-aligned rows
bool try_gdma(bool b_jfast, const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
              const float* bias, int act, const float* mask, int mask_act, hipStream_t s) {
  static const bool off = env_off("DVAE_GEMM_DMA");    // A/B switch, debug builds only
  if (off || Kc < 256 || N < 128 || M < 1 || Kc % 4 || lda % 4 || ldb % 4 || (b_jfast && N % 4)) return false;
  if ((((uintptr_t)a | (uintptr_t)b) & 15) != 0) return false;
  if ((long)Kc * ldb * 4 >= (1L << 31) || (long)M * lda * 4 >= (1L << 40)) return false;   // 32-bit per-slab steps
  // the largest tile that still gives (nearly) every CU a workgroup: 128 x 64, 64 x 64, else 32 x 32 with the
  // contraction split over the workgroup's four MFMA waves
  const long cols64 = (N + 63) / 64;
  static const int force = env_int("DVAE_GDMA_TILE", 0);       // A/B switch, debug builds only: 128 / 64 / 32
  const int tile = force ? force : ((M + 127) / 128 * cols64 >= 224 ? 128 : ((M + 63) / 64 * cols64 >= 192 ? 64 : 32));
#ifdef DVAE_DEBUG_SWITCHES
  static const int abl = env_int("DVAE_GDMA_ABLATE", 0);   // timing ablations of the forward form (results invalid)
  if (abl && !b_jfast && tile != 32) {
#define DVAE_GDMA_ABL(V)                                                                                               \
  if (abl == V) {                                                                                                      \
    if (tile == 128) launch_gdma_t<128, 64, 32, 4, 1, false, V>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s); \
    else launch_gdma_t<64, 64, 64, 3, 1, false, V>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s);    \
    return true;                                                                                                       \
  }
    DVAE_GDMA_ABL(1) DVAE_GDMA_ABL(2) DVAE_GDMA_ABL(3) DVAE_GDMA_ABL(8)
#undef DVAE_GDMA_ABL
  }
#endif
#define DVAE_GDMA_GO(TM, TN, KS, D, NWK)                                                                         \
  do {                                                                                                           \
    if (b_jfast) launch_gdma_t<TM, TN, KS, D, NWK, true>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s); \
    else launch_gdma_t<TM, TN, KS, D, NWK, false>(a, lda, b, ldb, c, ldc, M, N, Kc, bias, act, mask, mask_act, s);        \
  } while (0)
  if (tile == 128) DVAE_GDMA_GO(128, 64, 32, 4, 1);
  else if (tile == 64) DVAE_GDMA_GO(64, 64, 64, 3, 1);
  else DVAE_GDMA_GO(32, 32, 64, 3, 4);
#undef DVAE_GDMA_GO
  return true;
}

bool try_gdma_wgrad(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, hipStream_t s) {
  static const bool off = env_off("DVAE_GEMM_DMA");    // A/B switch, debug builds only
  if (off || M < 64 || N % 4 || K % 4) return false;
  if ((long)((N + 63) / 64) * ((K + 63) / 64) < 128) return false;     // few output tiles: the VAE's own FC layers (k_fcw32 / grouped)
  if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) != 0) return false;   // dw: 8-byte f32x2 stores at dw + row * K + col
  if ((long)64 * (N > K ? N : K) * 4 >= (1L << 31)) return false;
  constexpr int KS = 64, D = 3;
  const size_t lds = (size_t)(D + 1) * 2 * KS * 64 * sizeof(float);
  static DeviceOnce attr;
  if (attr.first())
    (void)hipFuncSetAttribute((const void*)k_gdma_wg<KS, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int tiles_n = (N + 63) / 64, tiles_k = (K + 63) / 64;
  hipLaunchKernelGGL((k_gdma_wg<KS, D>), dim3(tiles_n * tiles_k), dim3(512), lds, s, dy, x, dw, db, M, N, K, tiles_n, tiles_k);
  return true;
}

}  // namespace dvae
