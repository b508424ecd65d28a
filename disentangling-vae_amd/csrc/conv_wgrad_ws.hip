// Wave-specialised weight-gradient kernel of the 32 <-> 32 channel k4/s2/p1 convolutions at the two large geometries
// (32x32 <-> 16x16 and 16x16 <-> 8x8; conv2 / conv3 / convT1 / convT2 under training.py:157):
//     dw[cs][cb][kh][kw] = sum_p small[p][cs] * big[pix(p, kh, kw)][cb]          M = 32 cs, N = 32 cb per tap, K = pixels
//
// k_wgrad32 (conv_mfma.hip) keeps both activation tiles pixel-major in LDS (as they come from HBM), so every MFMA operand
// is a ds_read_b32 with a swizzled, non-affine address: 192 LDS reads and ~790 VALU instructions next to the 64 MFMAs of a
// unit and wave, 240 VGPRs (nothing else fits on the CU), matrix cores 0.58 busy.  Here
//   * the LOADER waves (4-7) write both tiles into LDS TRANSPOSED -- channel-major: sT[cs][pixel], bT[cb][row][column
//     parity][column/2] -- with row strides that are odd multiples of 16 bytes (conflict-free 16-byte reads);
//   * a COMPUTE wave (0-3, one per SIMD) owns the four taps of one kernel row kh: for 8 consecutive small pixels of a row it
//     reads ONE 16-byte A operand (its channel, 4 pixels per lane half) and FOUR aligned 16-byte quads of the big tile
//     (two per column parity); the kw = 0, 1 taps use a quad as it is, the kw = 2, 3 taps use the same data shifted by one
//     column -- a choice of registers, not a load.  5 LDS reads per 16 MFMAs, all addresses lane-base + immediates;
//   * tiles are double-buffered in LDS, one workgroup barrier per unit, the next-but-one tile is in flight in registers.
// Accumulators persist over the workgroup's units (4 taps x 16 registers per compute wave); the per-workgroup partial
// sums go to the workspace in k_wgrad32's format and k_wgrad32_reduce (conv_mfma.hip) finishes them in a fixed order.
#include "common.h"
#include "conv_mfma_common.h"
#include "wgrad_reduce.h"

namespace dvae {

#define WGW_MAX_BLOCKS WG_MAX_BLOCKS
#define WGW_STRIDE WG_STRIDE               // the reduce kernel (and its partial format) is shared with k_wgrad32

template <int HS>
struct WGeo {
  using G = Geo<HS>;
  static constexpr int CWP = (G::CW + 3) / 4 * 4;                       // column pairs per parity, padded to a quad: 20 / 12
  static constexpr int BSTR_RAW = G::BROWS * 2 * CWP;                   // floats per channel of the big tile
  static constexpr int BSTR = (BSTR_RAW / 4) % 2 ? BSTR_RAW : BSTR_RAW + 4;   // odd multiple of 4 floats: 404 / 436
  static constexpr int SSTR = 64 + 4;                                   // 64 pixels per channel of the small tile, padded
  static constexpr int BT_FLOATS = 32 * BSTR;
  static constexpr int ST_FLOATS = 32 * SSTR;
  static constexpr int BIG_SLOTS = G::BROWS * G::BPC * 8;               // 16-byte chunks of the big tile (with halo)
  static constexpr int BIG_NPF = (BIG_SLOTS + 255) / 256;
  static constexpr int BUF_FLOATS = BT_FLOATS + ST_FLOATS;
};

template <int HS>
__global__ __launch_bounds__(512) void k_wgrad32ws(const float* __restrict__ big, const float* __restrict__ small,
                                                   float* __restrict__ ws, int n_units
#ifdef DVAE_DEBUG_SWITCHES
                                                   , int abl   // timing ablations (DVAE_WGWS_ABLATE; results invalid): 1 loaders do not
                                                               // write LDS, 2 no tile loads, 8 no MFMAs, 16 no LDS operand reads, 32 no barrier
#endif
                                                   ) {
#ifndef DVAE_DEBUG_SWITCHES
  constexpr int abl = 0;
#endif
  using G = Geo<HS>;
  using W = WGeo<HS>;
  static_assert(G::IMGS == 1, "one image per unit");
  constexpr int HB = 2 * HS;
  extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 x (bT | sT)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wv < 4;
  const int stride = gridDim.x;
  const int unit0 = blockIdx.x;

  if (is_compute) {
    // ------------------------------------------------------------------ compute wave: kernel row kh, taps kw = 0..3
    const int kh = wv;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float sumS = 0.f, sumK1 = 0.f, sumK2 = 0.f;
    const int abase = i * W::SSTR + 4 * h;                            // + sy * HS + 8 gx
    const int bbase = i * W::BSTR + kh * 2 * W::CWP + 4 * h;          // + (2 sy * 2 + par) * CWP + 8 gx  (+ 4 for the second quad)
    __builtin_amdgcn_s_setprio(1);
    __syncthreads();                                                  // tile of the first unit is in buffer 0
    int buf = 0;
    for (int unit = unit0; unit < n_units; unit += stride) {
      const float* bt = smem + buf * W::BUF_FLOATS;
      const float* st = bt + W::BT_FLOATS;
      constexpr int NG = 64 / 8;                                      // groups of 8 pixels: (sy, gx)
      constexpr int GPR = HS / 8;                                     // groups per small row
      f32x4 A[2], P0a[2], P0b[2], P1a[2], P1b[2];
      if (abl & 16) {
#pragma unroll
        for (int c = 0; c < 2; ++c) { A[c] = f32x4{1.f, 2.f, 3.f, 4.f}; P0a[c] = A[c]; P0b[c] = A[c]; P1a[c] = A[c]; P1b[c] = A[c]; }
      }
      auto rd = [&](int g, int slot) {
        if (abl & 16) return;
        const int sy = g / GPR, gx = g % GPR;
        A[slot] = *reinterpret_cast<const f32x4*>(st + abase + sy * HS + 8 * gx);
        const float* bp = bt + bbase + (4 * sy) * W::CWP + 8 * gx;    // row 2 sy + kh, parity 0
        P0a[slot] = *reinterpret_cast<const f32x4*>(bp);
        P0b[slot] = *reinterpret_cast<const f32x4*>(bp + 4);
        P1a[slot] = *reinterpret_cast<const f32x4*>(bp + W::CWP);
        P1b[slot] = *reinterpret_cast<const f32x4*>(bp + W::CWP + 4);
      };
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      if (!(abl & 8)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int c = g & 1;
        if (g + 1 < NG) rd(g + 1, c ^ 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // small pixel sx = 8 gx + 4 h + j; big column pair cw = sx + (kw >> 1), parity kw & 1
          const float a = A[c][j];
          const float b0 = P0a[c][j];                                  // kw = 0
          const float b1 = P1a[c][j];                                  // kw = 1
          const float b2 = j < 3 ? P0a[c][j + 1] : P0b[c][0];          // kw = 2: parity 0, one column pair to the right
          const float b3 = j < 3 ? P1a[c][j + 1] : P1b[c][0];          // kw = 3
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
          sumS += a; sumK1 += b1; sumK2 += b2;
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);            // 5 DS reads (next group)
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);           // 16 MFMAs (this group)
      }
      }
      if (!(abl & 32)) __syncthreads();
      buf ^= 1;
    }
    // partial results of this workgroup, k_wgrad32's layout: ws[block][tap][cs][cb] + 160 bias floats
    float* wsw = ws + (long)blockIdx.x * WGW_STRIDE;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tap = kh * 4 + t;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int cs = (e & 3) + 8 * (e >> 2) + 4 * h;
        wsw[(tap * 32 + cs) * 32 + i] = acc[t][e];
      }
    }
    float* wsb = wsw + 16384;
    sumS += __shfl_xor(sumS, 32, 64);
    sumK1 += __shfl_xor(sumK1, 32, 64);
    sumK2 += __shfl_xor(sumK2, 32, 64);
    if (h == 0) {
      if (kh == 0) wsb[i] = sumS;                                     // sum of the small side per cs
      // taps (1,1) = 5, (1,2) = 6, (2,1) = 9, (2,2) = 10 together cover every big pixel exactly once (bias from the big side)
      if (kh == 1) { wsb[32 + i] = sumK1; wsb[64 + i] = sumK2; }
      if (kh == 2) { wsb[96 + i] = sumK1; wsb[128 + i] = sumK2; }
    }
  } else {
    // ------------------------------------------------------------------ loader waves
    const int lt = tid - 256;
    // big tile: slot s = (row r, padded column pc, 16-byte chunk) -> 4 channel-major LDS floats
    int b_lds[W::BIG_NPF], b_gofs[W::BIG_NPF], b_row[W::BIG_NPF];
#pragma unroll
    for (int k = 0; k < W::BIG_NPF; ++k) {
      const int s = lt + k * 256;
      b_lds[k] = -1; b_gofs[k] = 0; b_row[k] = 0;
      if (s < W::BIG_SLOTS) {
        const int chunk = s & 7;
        int q = s >> 3;
        const int pc = q % G::BPC; const int r = q / G::BPC;
        const int par = pc & 1, cw = pc >> 1, bx = pc - 1;
        b_lds[k] = (4 * chunk) * W::BSTR + (r * 2 + par) * W::CWP + cw;
        b_gofs[k] = ((r - 1) * HB + bx) * 32 + chunk * 4;
        b_row[k] = r | ((bx >= 0 && bx < HB) ? (1 << 16) : 0);
      }
    }
    // small tile: 64 pixels x 8 chunks = 512 slots, two per loader thread
    // two register sets: tiles k+2 and k+3 are in flight from HBM while tile k+1 sits in LDS (the whole chip issues its tile
    // loads in the same few hundred cycles after a barrier: one unit time of look-ahead does not cover the queueing delay)
    f32x4 pbA[W::BIG_NPF], psA[2], pbB[W::BIG_NPF], psB[2];
    auto load_unit = [&](int u, f32x4 (&pb)[W::BIG_NPF], f32x4 (&ps)[2]) {
      if (abl & 2) return;
      const long P0 = (long)u * G::U;
      const int n0 = (int)(P0 / (HS * HS));
      const int sy0 = (int)(P0 % (HS * HS)) / HS;
      const float* bbase_g = big + ((long)n0 * HB + 2 * sy0) * HB * 32;
#pragma unroll
      for (int k = 0; k < W::BIG_NPF; ++k) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int r = b_row[k] & 0xff;
        const int by = 2 * sy0 - 1 + r;
        if ((b_row[k] >> 16) && by >= 0 && by < HB) v = *reinterpret_cast<const f32x4*>(bbase_g + b_gofs[k]);
        pb[k] = v;
      }
      const float* sbase_g = small + P0 * 32;
#pragma unroll
      for (int k = 0; k < 2; ++k) ps[k] = *reinterpret_cast<const f32x4*>(sbase_g + (lt + k * 256) * 4);
    };
    auto store_unit = [&](int b, const f32x4 (&pb)[W::BIG_NPF], const f32x4 (&ps)[2]) {
      if (abl & 1) return;
      float* bt = smem + b * W::BUF_FLOATS;
      float* st = bt + W::BT_FLOATS;
#pragma unroll
      for (int k = 0; k < W::BIG_NPF; ++k) {
        if (b_lds[k] >= 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) bt[b_lds[k] + u * W::BSTR] = pb[k][u];
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int s = lt + k * 256;
        const int chunk = s & 7, p = s >> 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) st[(4 * chunk + u) * W::SSTR + p] = ps[k][u];
      }
    };
    if (unit0 < n_units) { load_unit(unit0, pbA, psA); store_unit(0, pbA, psA); }
    if (unit0 + stride < n_units) load_unit(unit0 + stride, pbA, psA);
    if (unit0 + 2 * stride < n_units) load_unit(unit0 + 2 * stride, pbB, psB);
    __syncthreads();
    // set A holds tiles k+1 (k even), set B tiles k+1 (k odd); the loop is unrolled by two so that the sets stay static
    int unit = unit0;
    while (unit < n_units) {
      if (unit + stride < n_units) store_unit(1, pbA, psA);
      if (unit + 3 * stride < n_units) load_unit(unit + 3 * stride, pbA, psA);
      if (!(abl & 32)) __syncthreads();
      unit += stride;
      if (unit >= n_units) break;
      if (unit + stride < n_units) store_unit(0, pbB, psB);
      if (unit + 3 * stride < n_units) load_unit(unit + 3 * stride, pbB, psB);
      if (!(abl & 32)) __syncthreads();
      unit += stride;
    }
  }
}

template <int HS>
static int launch_wgrad_ws_t(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N,
                             float* ws, hipStream_t s) {
  using W = WGeo<HS>;
  const int n_units = (int)(((long)N * HS * HS) / 64);
  int grid = n_units < WGW_MAX_BLOCKS ? n_units : WGW_MAX_BLOCKS;
  {
    static const int cap = env_int("DVAE_WGRAD_GRID", WGW_MAX_BLOCKS);   // debug builds: A/B of the persistent grid size
    if (cap > 0 && cap < grid) grid = cap;
  }
  const size_t lds = (size_t)2 * W::BUF_FLOATS * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) { (void)hipFuncSetAttribute((const void*)k_wgrad32ws<HS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
#ifdef DVAE_DEBUG_SWITCHES
  static const int abl = env_int("DVAE_WGWS_ABLATE", 0);
  hipLaunchKernelGGL(k_wgrad32ws<HS>, dim3(grid), dim3(512), lds, s, big, small, ws, n_units, abl);
#else
  hipLaunchKernelGGL(k_wgrad32ws<HS>, dim3(grid), dim3(512), lds, s, big, small, ws, n_units);
#endif
  DVAE_CHECK_LAUNCH();
  return launch_wgrad32_reduce(ws, dw, db, bias_from_big, grid, s);
}

// NHWC on both sides, Hs in {8, 16}; returns 1 if not applicable
int launch_wgrad_mfma32_ws(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N, int Hs,
                           float* ws, hipStream_t s) {
  if (Hs == 16) return launch_wgrad_ws_t<16>(big, small, dw, db, bias_from_big, N, ws, s);
  if (Hs == 8) return launch_wgrad_ws_t<8>(big, small, dw, db, bias_from_big, N, ws, s);
  return 1;
}

}  // namespace dvae
