// Round-3 candidate of the wave-specialised "down" kernel (big -> small: Conv2d forward, ConvTranspose2d dgrad), debug builds
// only, selected with DVAE_DOWN_D=1.  Same tiles, loaders, MFMA loop and results as k_down32ws (conv_mfma.hip); what changes is
// WHEN a unit's epilogue runs.  k_down32ws: MFMAs of unit u | barrier | chain sums, bias, ReLU, mask, 8 scattered stores per lane
// -- the matrix pipe idles during that epilogue, and the timing ablations price it (profiles/r02_run14_15_*: output stores 3 %
// alone / 8 % once the loaders are out of the way, per-unit barrier 6.5 %).  Here the four accumulator chains of unit u are summed
// right after its last MFMA and everything else -- bias, activation, mask, the 8 stores -- is issued inside the MFMA stream of unit
// u+1, one result row per kernel tap; the mask of unit u is requested at the start of unit u and first looked at one unit later.
// The barrier follows the last MFMA directly.  Measured: see profiles/ of round 3 (first run: tools/r3_first.sh).
#include <stdlib.h>
#include "common.h"
#include "conv_mfma_common.h"

namespace dvae {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int HS, bool MASK>
__global__ __launch_bounds__(512) void k_down32wsd(const float* __restrict__ big, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ mask,
                                                   float* __restrict__ out, int N, int act, int n_units) {
  using G = Geo<HS>;
  static_assert(G::IMGS == 1, "one image per unit");
  constexpr int LNPF = (G::BIG_SLOTS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                          // 16384 floats
  float* bt0 = smem + 16384;
  float* bt1 = bt0 + G::BIG_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wv < 4;
  const int i16 = lane & 15, kq = lane >> 4;
  const int p = (wv & 3) * 16 + i16;
  const int sy_l = (p / HS) % G::R, sx = p % HS;
  const int stride = gridDim.x;

  SlotDesc<LNPF> sd;
  f32x4 pfa[LNPF], pfb[LNPF];
  const int ltid = tid - 256;
  if (!is_compute) init_big_slots<HS, 256, LNPF>(sd, ltid);
  int unit = blockIdx.x;
  if (!is_compute && unit < n_units) load_big<HS, LNPF>(pfa, sd, big, unit, N);
  stage_weights<true>(w, wl, tid);
  if (!is_compute && unit < n_units) store_big<HS, LNPF>(pfa, sd, bt0);
  __syncthreads();
  if (!is_compute) {
    if (unit + stride < n_units) load_big<HS, LNPF>(pfa, sd, big, unit + stride, N);
    if (unit + 2 * stride < n_units) load_big<HS, LNPF>(pfb, sd, big, unit + 2 * stride, N);
  }
  if (is_compute) {
    const float bv0 = bias ? bias[i16] : 0.f, bv1 = bias ? bias[16 + i16] : 0.f;
    struct Res { f32x4v a[2]; float m[2][4]; long obase; };      // a finished unit: chain sums, its mask, where it goes
    // one result row of a finished unit: bias, activation, mask, store (nh = channel half, r = pixel within the 4-pixel group)
    auto finish_row = [&](const Res& P, int nh, int r) {
      float v = epilogue_act(P.a[nh][r] + (nh ? bv1 : bv0), act);
      if (MASK) v = P.m[nh][r] > 0.f ? v : 0.f;
      out[P.obase + r * 32 + nh * 16] = v;
    };
    // MFMAs of one unit from tile `bt` into C (chain-summed at the end); the previous unit P is finished in the shadow of the first
    // eight taps.  The mask of THIS unit is requested first and only copied into C at the very end: a whole unit of latency cover.
    auto unit_body = [&](Res& C, const Res& P, const float* bt, long obase, bool have_prev) {
      float mv[2][4];
      if (MASK) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int r = 0; r < 4; ++r) mv[nh][r] = mask[obase + r * 32 + nh * 16];
      }
      f32x4v acc[2][4];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[nh][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
      f32x4 A0[2], A1[2], B00[2], B01[2], B10[2], B11[2];
      auto rd = [&](int tap, int slot) {
        const int kh = tap >> 2, kw = tap & 3;
        const int r = 2 * sy_l + kh;
        const int par = kw & 1, cw = sx + (kw >> 1);
        const float* arow = bt + ((r * 2 + par) * G::CW + cw) * 32;
        const int sw = swz_big<HS>(r, cw);
        const float* brow = wl + (tap * 8) * 128 + i16 * 4;
        A0[slot] = *reinterpret_cast<const f32x4*>(arow + ((kq ^ sw) << 2));
        A1[slot] = *reinterpret_cast<const f32x4*>(arow + (((4 + kq) ^ sw) << 2));
        B00[slot] = *reinterpret_cast<const f32x4*>(brow + kq * 128);
        B01[slot] = *reinterpret_cast<const f32x4*>(brow + kq * 128 + 64);
        B10[slot] = *reinterpret_cast<const f32x4*>(brow + (4 + kq) * 128);
        B11[slot] = *reinterpret_cast<const f32x4*>(brow + (4 + kq) * 128 + 64);
      };
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int cur = t & 1;
        if (t + 1 < 16) rd(t + 1, cur ^ 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[cur][j], B00[cur][j], acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[cur][j], B01[cur][j], acc[1][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[cur][j], B10[cur][j], acc[0][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[cur][j], B11[cur][j], acc[1][j], 0, 0, 0);
        if (have_prev && t < 8) finish_row(P, t >> 2, t & 3);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);    // 6 DS reads (next tap)
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // 16 MFMAs (this tap)
        if (have_prev && t < 8) {
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // bias + activation (+ mask select) of one result
          __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // its store
        }
      }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        C.a[nh] = (acc[nh][0] + acc[nh][1]) + (acc[nh][2] + acc[nh][3]);      // same order as k_down32ws: bit-identical results
#pragma unroll
        for (int r = 0; r < 4; ++r) C.m[nh][r] = MASK ? mv[nh][r] : 0.f;
      }
      C.obase = obase;
    };
    auto obase_of = [&](int u) -> long { return ((long)u * G::U + (wv & 3) * 16 + 4 * kq) * 32 + i16; };
    __builtin_amdgcn_s_setprio(1);
    Res X, Y;
    // units alternate between X and Y; every iteration ends with exactly one barrier (pairs with the loaders' loop)
    if (unit < n_units) {
      unit_body(X, X, bt0, obase_of(unit), false);
      __syncthreads();
      unit += stride;
      for (;;) {
        if (unit >= n_units) {
#pragma unroll
          for (int t = 0; t < 8; ++t) finish_row(X, t >> 2, t & 3);
          break;
        }
        unit_body(Y, X, bt1, obase_of(unit), true);
        __syncthreads();
        unit += stride;
        if (unit >= n_units) {
#pragma unroll
          for (int t = 0; t < 8; ++t) finish_row(Y, t >> 2, t & 3);
          break;
        }
        unit_body(X, Y, bt0, obase_of(unit), true);
        __syncthreads();
        unit += stride;
      }
    }
  } else {
    // loader: registers pfa hold tile u+1, pfb tile u+2 (in flight); alternate -- identical to k_down32ws
    while (unit < n_units) {
      if (unit + stride < n_units) store_big<HS, LNPF>(pfa, sd, bt1);
      __syncthreads();
      if (unit + 3 * stride < n_units) load_big<HS, LNPF>(pfa, sd, big, unit + 3 * stride, N);
      unit += stride;
      if (unit >= n_units) break;
      if (unit + stride < n_units) store_big<HS, LNPF>(pfb, sd, bt0);
      __syncthreads();
      if (unit + 3 * stride < n_units) load_big<HS, LNPF>(pfb, sd, big, unit + 3 * stride, N);
      unit += stride;
    }
  }
}

template <int HS>
static int launch_down_wsd(const ConvArgs& a, hipStream_t s) {
  using G = Geo<HS>;
  const int n_units = (int)(((long)a.N * HS * HS + 63) / 64);
  const int grid = n_units < 256 ? n_units : 256;
  const size_t lds = (16384 + 2 * G::BIG_FLOATS) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_down32wsd<HS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_down32wsd<HS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  if (a.mask) hipLaunchKernelGGL((k_down32wsd<HS, true>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
  else hipLaunchKernelGGL((k_down32wsd<HS, false>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
  DVAE_CHECK_LAUNCH();
  return 0;
}

// 32 <-> 32 channels, NHWC on both sides, Hs == Ws in {8, 16}; returns 1 if not applicable (the caller falls back to k_down32ws)
int launch_down_mfma32_d(const ConvArgs& a, hipStream_t s) {
  if (!(a.Cb == 32 && a.Cs == 32 && a.Hs == a.Ws && (a.Hs == 8 || a.Hs == 16) && a.big_layout == DVAE_NHWC &&
        a.out_layout == DVAE_NHWC))
    return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  return a.Hs == 16 ? launch_down_wsd<16>(a, s) : launch_down_wsd<8>(a, s);
}

}  // namespace dvae
