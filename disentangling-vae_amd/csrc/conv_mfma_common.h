// Shared pieces of the gfx950 MFMA conv kernels (conv_mfma.hip and its siblings): unit geometry, LDS swizzles,
// the per-thread staging descriptors and loaders, weight staging, small helpers.  See conv_mfma.hip for the design.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace dvae {

template <int HS>
struct Geo {
  static constexpr int HB = 2 * HS;
  static constexpr int U = 64;                                        // small pixels per unit
  static constexpr int IMGS = (HS * HS >= U) ? 1 : U / (HS * HS);      // images per unit
  static constexpr int R = (HS * HS >= U) ? U / HS : HS;              // small rows per image per unit
  static constexpr int BROWS = 2 * R + 2;                             // big rows (with halo) per image
  static constexpr int CW = HS + 1;                                   // column pairs per parity
  static constexpr int BPC = 2 * HS + 2;                              // padded big columns
  static constexpr int BIG_FLOATS = IMGS * BROWS * 2 * CW * 32;
  static constexpr int BIG_SLOTS = IMGS * BROWS * BPC * 8;            // 16-byte slots to stage
  static constexpr int BIG_NPF = (BIG_SLOTS + 511) / 512;
  static constexpr int SROWS = R + 2, SCOLS = HS + 2;                 // small tile with halo
  static constexpr int SH_FLOATS = IMGS * SROWS * SCOLS * 32;
  static constexpr int SH_SLOTS = IMGS * SROWS * SCOLS * 8;
  static constexpr int SH_NPF = (SH_SLOTS + 511) / 512;
};

template <int HS> __device__ __forceinline__ int swz_big(int r, int cw);
template <> __device__ __forceinline__ int swz_big<16>(int r, int cw) { return (cw >> 1) & 7; }
template <> __device__ __forceinline__ int swz_big<8>(int r, int cw) { return ((cw >> 1) & 3) | (((r >> 1) & 1) << 2); }
template <> __device__ __forceinline__ int swz_big<4>(int r, int cw) { return ((cw >> 1) & 1) | (((r >> 1) & 3) << 1); }
template <int HS> __device__ __forceinline__ int swz_small(int row, int col);
template <> __device__ __forceinline__ int swz_small<16>(int row, int col) { return (col >> 1) & 7; }
template <> __device__ __forceinline__ int swz_small<8>(int row, int col) { return ((col >> 1) & 3) | ((row & 1) << 2); }
template <> __device__ __forceinline__ int swz_small<4>(int row, int col) { return ((col >> 1) & 1) | ((row & 3) << 1); }

// ---- staging helpers ----------------------------------------------------------------------
// A thread stages the same 16-byte slots of every unit, so the slot -> (LDS offset, global offset,
// row) decode (integer div/mod) is done ONCE per kernel; per unit only the image / row range
// checks remain.
template <int NPF>
struct SlotDesc {
  int lds[NPF];    // float offset of the (swizzled) LDS destination, -1: slot unused
  int gofs[NPF];   // float offset of the source relative to the unit base pointer
  int rimg[NPF];   // row-in-tile | (image-in-unit << 8) | (column valid << 16)
};

template <int HS, int NTHR = 512, int NPF = Geo<HS>::BIG_NPF>
__device__ __forceinline__ void init_big_slots(SlotDesc<NPF>& d, int tid) {
  using G = Geo<HS>;
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    int s = tid + k * NTHR;
    d.lds[k] = -1; d.gofs[k] = 0; d.rimg[k] = 0;
    if (s < G::BIG_SLOTS) {
      int chunk = s & 7;
      int t = s >> 3;
      int pc = t % G::BPC; t /= G::BPC;
      int r = t % G::BROWS;
      int img = t / G::BROWS;
      int par = pc & 1, cw = pc >> 1, bx = pc - 1;
      d.lds[k] = (((img * G::BROWS + r) * 2 + par) * G::CW + cw) * 32 + ((chunk ^ swz_big<HS>(r, cw)) << 2);
      d.gofs[k] = ((img * G::HB + (r - 1)) * G::HB + bx) * 32 + chunk * 4;
      d.rimg[k] = r | (img << 8) | ((bx >= 0 && bx < G::HB) ? (1 << 16) : 0);
    }
  }
}

template <int HS, int NPF = Geo<HS>::BIG_NPF>
__device__ __forceinline__ void load_big(f32x4 (&pf)[NPF], const SlotDesc<NPF>& d,
                                         const float* __restrict__ big, int unit, int N) {
  using G = Geo<HS>;
  const long P0 = (long)unit * G::U;
  const int n0 = (int)(P0 / (HS * HS));
  const int sy0 = (int)(P0 % (HS * HS)) / HS;
  const float* base = big + ((long)n0 * G::HB + 2 * sy0) * G::HB * 32;
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int r = d.rimg[k] & 0xff, img = (d.rimg[k] >> 8) & 0xff;
    const int by = 2 * sy0 - 1 + r;
    if ((d.rimg[k] >> 16) && n0 + img < N && by >= 0 && by < G::HB)
      v = *reinterpret_cast<const f32x4*>(base + d.gofs[k]);
    pf[k] = v;
  }
}

template <int HS, int NPF = Geo<HS>::BIG_NPF>
__device__ __forceinline__ void store_big(const f32x4 (&pf)[NPF], const SlotDesc<NPF>& d,
                                          float* bt) {
#pragma unroll
  for (int k = 0; k < NPF; ++k)
    if (d.lds[k] >= 0) *reinterpret_cast<f32x4*>(bt + d.lds[k]) = pf[k];
}

// small_nchw: the small tensor is [N][32][HS*HS] (the FC stack's (c,h,w) order at the 4x4 end of the
// network) instead of NHWC: a 16-byte LDS slot = 4 channels of one pixel is then gathered with four
// 4-byte loads HS*HS floats apart (the tensor is 2 KB per image).
template <int HS>
__device__ __forceinline__ void init_small_slots(SlotDesc<Geo<HS>::SH_NPF>& d, int tid, int small_nchw = 0) {
  using G = Geo<HS>;
#pragma unroll
  for (int k = 0; k < G::SH_NPF; ++k) {
    int s = tid + k * 512;
    d.lds[k] = -1; d.gofs[k] = 0; d.rimg[k] = 0;
    if (s < G::SH_SLOTS) {
      int chunk = s & 7;
      int t = s >> 3;
      int col = t % G::SCOLS; t /= G::SCOLS;
      int row = t % G::SROWS;
      int img = t / G::SROWS;
      int sx = col - 1;
      d.lds[k] = ((img * G::SROWS + row) * G::SCOLS + col) * 32 + ((chunk ^ swz_small<HS>(row, col)) << 2);
      d.gofs[k] = small_nchw ? ((img * 32 + chunk * 4) * HS + (row - 1)) * HS + sx
                             : ((img * HS + (row - 1)) * HS + sx) * 32 + chunk * 4;
      d.rimg[k] = row | (img << 8) | ((sx >= 0 && sx < HS) ? (1 << 16) : 0);
    }
  }
}

template <int HS>
__device__ __forceinline__ void load_small_halo(f32x4 (&pf)[Geo<HS>::SH_NPF], const SlotDesc<Geo<HS>::SH_NPF>& d,
                                                const float* __restrict__ small, int unit, int N, int small_nchw = 0) {
  using G = Geo<HS>;
  const long P0 = (long)unit * G::U;
  const int n0 = (int)(P0 / (HS * HS));
  const int sy0 = (int)(P0 % (HS * HS)) / HS;
  const float* base = small_nchw ? small + (long)n0 * 32 * HS * HS + sy0 * HS : small + ((long)n0 * HS + sy0) * HS * 32;
#pragma unroll
  for (int k = 0; k < G::SH_NPF; ++k) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int row = d.rimg[k] & 0xff, img = (d.rimg[k] >> 8) & 0xff;
    const int sy = sy0 - 1 + row;
    if ((d.rimg[k] >> 16) && n0 + img < N && sy >= 0 && sy < HS) {
      if (small_nchw) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = base[d.gofs[k] + u * HS * HS];
      } else {
        v = *reinterpret_cast<const f32x4*>(base + d.gofs[k]);
      }
    }
    pf[k] = v;
  }
}

template <int HS>
__device__ __forceinline__ void store_small_halo(const f32x4 (&pf)[Geo<HS>::SH_NPF], const SlotDesc<Geo<HS>::SH_NPF>& d,
                                                 float* st) {
  using G = Geo<HS>;
#pragma unroll
  for (int k = 0; k < G::SH_NPF; ++k)
    if (d.lds[k] >= 0) *reinterpret_cast<f32x4*>(st + d.lds[k]) = pf[k];
}

// weights w[cs][cb][16] -> LDS image wl[tap][kc/4][n][kc%4] where kc is the contracted channel
// and n the output channel.  KC_IS_CB: down (contract over cb, n = cs); else up (contract cs).
template <bool KC_IS_CB>
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, float* wl, int tid) {
  // all 8 float4 loads of a thread are issued before the first LDS write: ONE memory latency per
  // workgroup instead of one per loop iteration (the staging is on every launch's critical path)
  f32x4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(w + (tid + k * 512) * 4);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = (tid + k * 512) * 4;            // (cs, cb, tap..tap+3)
    const int cs = idx >> 9, cb = (idx >> 4) & 31, tap = idx & 15;
    const int kc = KC_IS_CB ? cb : cs;
    const int n = KC_IS_CB ? cs : cb;
    float* dst = wl + (((kc >> 2) * 32 + n) * 4 + (kc & 3));
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(tap + j) * 1024] = v[k][j];
  }
}

// Straight copy of a PRE-STAGED 64 KB weight image (dvae_stage_weights: the image already is in the LDS order above) by the
// first 8 waves of the workgroup: 8 LDS-DMA transfers of 1 KB per wave (global_load_lds_dwordx4: no staging registers, no
// ds_write pass); the data is in LDS once every wave has passed this function AND the following workgroup barrier.
__device__ __forceinline__ void copy_weight_image(const float* __restrict__ img, float* wl, int tid) {
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wv < 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int base = (k * 8 + wv) * 256;             // 256 floats = 64 lanes x 16 bytes; the LDS side is lane-linear
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img + base + lane * 4),
                                       (__attribute__((address_space(3))) void*)(wl + base), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // LDS-DMA completes in vmcnt order; the barrier that follows publishes it
}

__device__ __forceinline__ float epilogue_act(float v, int act) {
  if (act == DVAE_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

#define MFMA4(acc, a, b)                                                       \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a)[0], (b)[0], acc, 0, 0, 0);    \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a)[1], (b)[1], acc, 0, 0, 0);    \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a)[2], (b)[2], acc, 0, 0, 0);    \
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a)[3], (b)[3], acc, 0, 0, 0);

}  // namespace dvae
