// The 4x4 end of the conv stacks (conv_64 / convT_64 of the 64x64 geometry: encoders.py:57-58,76-77, decoders.py:55-57,74-76) as
// __device__ bodies of ONE unit = 4 images, for the launches that own whole images anyway: the FC chain kernels
// (fc_chain.hip) run them as prologue / epilogue, so that a training step no longer pays four 8-12 us launches (k_down32<4> /
// k_up32<4>, conv_mfma.hip) for 0.5 GFLOP each next to the chain.  The arithmetic -- operand order, accumulator chains, the
// order in which the four kh-slices are added -- is k_down32<4, MASK>'s / k_up32<4, MASK>'s, statement for statement: results
// are bit-identical to those launches (tests/test_gpu_fused_core.py::test_fc_chain_with_conv_ends).
#pragma once
#include "conv_mfma_common.h"

namespace dvae {

#define C4_WL_FLOATS 16384                                  // a staged 64 KB weight image
#define C4_BT_FLOATS (Geo<4>::BIG_FLOATS)                   // 12800: the 8x8 side of 4 images with halo
#define C4_RED_FLOATS 8192                                  // k_down32's cross-wave reduction buffer
#define C4_ST_FLOATS (Geo<4>::SH_FLOATS)                    // 4608: the 4x4 side of 4 images with halo

// LDS-DMA of a staged weight image: issue only (copy_weight_image waits; here the transfers fly under the chain's layers)
__device__ __forceinline__ void c4_weight_image_issue(const float* __restrict__ img, float* wl, int tid) {
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int base = (k * 8 + wv) * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + base + lane * 4),
                                     (__attribute__((address_space(3))) void*)(wl + base), 16, 0, 0);
  }
}

// float offset of (image img_l of the unit, channel c, position pos = 4 y + x) inside the small tile with halo (store_small_halo's
// layout: 16-byte channel chunks swizzled by swz_small<4>)
__device__ __forceinline__ int c4_st_index(int img_l, int c, int pos) {
  const int row = (pos >> 2) + 1, col = (pos & 3) + 1;
  return ((img_l * 6 + row) * 6 + col) * 32 + (((c >> 2) ^ swz_small<4>(row, col)) << 2) + (c & 3);
}

// "down" (Conv2d forward / ConvTranspose2d input gradient), unit `unit` of big[N][8][8][32] (NHWC; its tile is in pf, loaded by the
// caller with load_big<4>) -> out[N][512] ((c,h,w)
// order) AND rows 0..3 of `tile` (LDS, row stride `ts`; images beyond N: zero rows).  512 threads; wl = the layer's "down"
// image (complete and published by a barrier before the call), bt / red = scratch.  Two workgroup barriers inside; the caller
// adds the one that publishes `tile` (TILE = false: no tile, out only).  mask (MASK): [N][512], same order as out.
template <bool MASK, bool TILE>
__device__ __forceinline__ void c4_down_unit(const float* __restrict__ bias,
                                             const float* __restrict__ mask, float* __restrict__ out, float* tile, int ts,
                                             int N, int act, int unit, const float* wl, float* bt, float* red,
                                             const f32x4 (&pf)[Geo<4>::BIG_NPF], const SlotDesc<Geo<4>::BIG_NPF>& sd) {
  using G = Geo<4>;
  constexpr int HS = 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mt = wv & 1, kh = wv >> 1;
  const int i = lane & 31, h = lane >> 5;
  const int p = mt * 32 + i;
  const int img_l = p / (G::R * HS), sy_l = (p / HS) % G::R, sx = p % HS;
  const int r = 2 * sy_l + kh;
  const float bv = bias ? bias[i] : 0.f;
  store_big<HS>(pf, sd, bt);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int kw = 0; kw < 4; ++kw) {
    const int par = kw & 1, cw = sx + (kw >> 1);
    const float* arow = bt + (((img_l * G::BROWS + r) * 2 + par) * G::CW + cw) * 32;
    const int sw = swz_big<HS>(r, cw);
    const float* brow = wl + ((kh * 4 + kw) * 8) * 128 + i * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int chunk = 2 * q + h;
      f32x4 a = *reinterpret_cast<const f32x4*>(arow + ((chunk ^ sw) << 2));
      f32x4 b = *reinterpret_cast<const f32x4*>(brow + chunk * 128);
      MFMA4(acc, a, b)
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(((mt * 4 + (e >> 2)) * 4 + kh) * 4 + (e & 3)) * 64 + lane] = acc[e];
  __syncthreads();
  const int pl0 = mt * 32 + 8 * kh + 4 * h;                  // pixel inside the unit of this lane's first value
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = (kh == 0 ? acc[j] : kh == 1 ? acc[4 + j] : kh == 2 ? acc[8 + j] : acc[12 + j]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s != kh) v += red[(((mt * 4 + kh) * 4 + s) * 4 + j) * 64 + lane];
    const int pl = pl0 + j, il = pl >> 4, pos = pl & 15;
    const long n = (long)unit * G::IMGS + il;
    const long idx = (n * 32 + i) * 16 + pos;
    v = epilogue_act(v + bv, act);
    const bool in = n < N;
    if (MASK) v = (in && mask[idx] > 0.f) ? v : 0.f;
    if (!in) v = 0.f;
    if (in) out[idx] = v;
    if (TILE) tile[il * ts + i * 16 + pos] = v;
  }
}

// "up" (ConvTranspose2d forward / Conv2d input gradient): the small tile `st` (LDS, with halo, complete and published) of unit
// `unit` -> out[N][8][8][32] (NHWC).  wl = the layer's "up" image (complete and published).  No barrier inside.
// mask (MASK): [N][8][8][32], same order as out.
template <bool MASK>
__device__ __forceinline__ void c4_up_unit(const float* st, const float* wl, const float* __restrict__ bias,
                                           const float* __restrict__ mask, float* __restrict__ out, int N, int act, int unit) {
  using G = Geo<4>;
  constexpr int HS = 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cls = wv & 3, mt = wv >> 2;
  const int py = cls >> 1, px = cls & 1;
  const int i = lane & 31, h = lane >> 5;
  const int p = mt * 32 + i;
  const int img_l = p / (G::R * HS), m = (p / HS) % G::R, l = p % HS;
  const float bv = bias ? bias[i] : 0.f;
  long offs[16];
  float mv[16];
  // (up_offsets<4> of conv_mfma.hip)
  {
    const long P0 = (long)unit * G::U;
    const int n0 = (int)(P0 / (HS * HS));
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rowp = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int pp = mt * 32 + rowp;
      const int im = pp / (G::R * HS), mm = (pp / HS) % G::R, ll = pp % HS;
      const int by = 2 * mm + py, bx = 2 * ll + px;
      offs[e] = (((long)(n0 + im) * G::HB + by) * G::HB + bx) * 32 + i;
      if (MASK) mv[e] = (n0 + im < N) ? mask[offs[e]] : 0.f;
    }
  }
  f32x16 accs[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) accs[c][e] = 0.f;
  f32x4 Av[2], Bv[2];
  auto rd = [&](int g, int slot) {
    const int ty = g >> 3, tx = (g >> 2) & 1, q = g & 3;
    const int kh = 1 - py + 2 * ty, kw = 1 - px + 2 * tx;
    const int row = m + (py - ty) + 1, col = l + (px - tx) + 1;
    const float* arow = st + ((img_l * G::SROWS + row) * G::SCOLS + col) * 32;
    const int sw = swz_small<HS>(row, col);
    const float* brow = wl + ((kh * 4 + kw) * 8) * 128 + i * 4;
    const int chunk = 2 * q + h;
    Av[slot] = *reinterpret_cast<const f32x4*>(arow + ((chunk ^ sw) << 2));
    Bv[slot] = *reinterpret_cast<const f32x4*>(brow + chunk * 128);
  };
  rd(0, 0);
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int cur = g & 1;
    if (g + 1 < 16) rd(g + 1, cur ^ 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) accs[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[cur][j], Bv[cur][j], accs[j], 0, 0, 0);
  }
  const f32x16 acc = (accs[0] + accs[1]) + (accs[2] + accs[3]);
  const int n0 = unit * G::IMGS;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float v = epilogue_act(acc[e] + bv, act);
    if (MASK) v = mv[e] > 0.f ? v : 0.f;
    const int im = (mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) / (G::R * HS);
    if (n0 + im < N) out[offs[e]] = v;
  }
}

}  // namespace dvae
