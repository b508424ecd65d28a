// Fixed-order reductions of the per-workgroup partial sums of the conv weight-gradient kernels (k_wgrad32 / k_wgrad32ws:
// ws[block][tap][cs][cb] + 160 bias floats per block; k_wgrad_thin<C>: ws[block][nt][cs][32] + (1 + NT) x 32 bias floats), as
// __device__ bodies shared by the per-layer reduce kernels and the grouped one (wgrad_grouped.hip: every layer of a
// training step in ONE launch).
#pragma once
#include "common.h"

namespace dvae {

#define WG_MAX_BLOCKS 256
// stride between per-workgroup partial buffers: NOT a multiple of 64 KB, so that the reduce kernel's
// loads of one output across all partials spread over HBM channels instead of hammering one
#define WG_STRIDE (16384 + 320)
#define WG_REDUCE_BLOCKS (256 + 2)         // workgroups of the 32-channel reduction (the last two: bias)
#define WT_MAX_BLOCKS 512
#define WT_REDUCE_BLOCKS(C) (((16 * (C) + 31) / 32) * 64 + 2)   // NT x 1024 workspace positions, 16 per workgroup (+ 2: bias)

// bias gradient: 2 workgroups x (16 channels x 16 partial-groups)
__device__ __forceinline__ void wgrad32_bias_reduce(const float* __restrict__ ws, float* __restrict__ db,
                                                    int bias_from_big, int nblk, int blk) {
  __shared__ float red[16][16];
  const int o = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int c = blk * 16 + o;
  const float* wsb = ws + 16384;
  float pv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = gq; g < nblk; g += 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gg = g + 16 * u;
      const float* q = wsb + (long)(gg < nblk ? gg : 0) * WG_STRIDE;
      float v = bias_from_big ? (q[32 + c] + q[64 + c]) + (q[96 + c] + q[128 + c]) : q[c];
      pv[u] += gg < nblk ? v : 0.f;
    }
  }
  red[gq][o] = (pv[0] + pv[1]) + (pv[2] + pv[3]);
  __syncthreads();
  if (gq == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    db[c] = t;
  }
}

// 256 workgroups x (16 positions of 16 bytes = 64 outputs) x 16 partial-groups.  Every lane has ALL its partials in flight
// at once (16 x 16-byte loads, a wave instruction covers 4 partials x 256 contiguous bytes): the reduction is one round trip
// to L2 / HBM instead of two rounds of 4-byte loads.  Fixed summation order: tree over u per lane, then groups 0..15.
__device__ __forceinline__ void wgrad32_reduce_body(int blk_x, const float* __restrict__ ws, float* __restrict__ dw,
                                                    float* __restrict__ db, int bias_from_big, int nblk) {
  if (blk_x >= 256) {                                // the last two workgroups reduce the bias gradient
    if (db) wgrad32_bias_reduce(ws, db, bias_from_big, nblk, blk_x - 256);
    return;
  }
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const int p = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const float* src = ws + (long)gq * WG_STRIDE + (blk_x * 16 + p) * 4;
  f32x4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {                     // partial gq + 16 u  (nblk <= WG_MAX_BLOCKS = 256)
    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (gq + 16 * u < nblk) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(16 * u) * WG_STRIDE);
  }
#pragma unroll
  for (int w = 1; w < 16; w *= 2)
#pragma unroll
    for (int u = 0; u < 16; u += 2 * w) v[u] += v[u + w];
  *reinterpret_cast<f32x4*>(&red[gq][p * 4]) = v[0];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int o = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    const int idx = blk_x * 64 + o;                  // (tap, cs, cb)
    const int tap = idx >> 10, cs = (idx >> 5) & 31, cb = idx & 31;
    dw[(cs * 32 + cb) * 16 + tap] = t;
  }
}


// bias gradient of the thin layers: (16 channels x 16 partial-groups) per workgroup
template <int C>
__device__ __forceinline__ void wgrad_thin_bias_reduce(const float* __restrict__ ws, float* __restrict__ db,
                                                       int bias_from_big, int nblk, int blk) {
  constexpr int NT = (16 * C + 31) / 32;
  constexpr int STRIDE = NT * 1024 + 32 + NT * 32;
  __shared__ float redb[16][16];
  const int o = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int c = blk * 16 + o;
  const int nout = bias_from_big ? C : 32;
  // slot [0,32) = sum of the small side per cs; slots 32.. = per (cb,tap) column sums of the big side, of which
  // taps (kh,kw) in {1,2}x{1,2} cover every big pixel exactly once
  const int cc = c < nout ? c : 0;
  const int t5 = cc * 16 + 5, t6 = cc * 16 + 6, t9 = cc * 16 + 9, t10 = cc * 16 + 10;
  float pv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = gq; g < nblk; g += 64) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gg = g + 16 * u;
      const float* q = ws + (long)(gg < nblk ? gg : 0) * STRIDE + NT * 1024;
      float v;
      if (bias_from_big)
        v = (q[32 + (t5 >> 5) * 32 + (t5 & 31)] + q[32 + (t6 >> 5) * 32 + (t6 & 31)]) +
            (q[32 + (t9 >> 5) * 32 + (t9 & 31)] + q[32 + (t10 >> 5) * 32 + (t10 & 31)]);
      else
        v = q[cc];
      pv[u] += gg < nblk ? v : 0.f;
    }
  }
  redb[gq][o] = (pv[0] + pv[1]) + (pv[2] + pv[3]);
  __syncthreads();
  if (gq == 0 && c < nout) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += redb[k][o];
    db[c] = t;
  }
}

// Workgroup = 4 workspace positions of 16 bytes (16 of the NT x 1024 [nt][cs][j] slots) x 64 partial-groups: all 8 loads of a
// lane in flight at once (nblk <= WT_MAX_BLOCKS = 512), one round trip.  Fixed summation order: tree over u per lane, groups
// in eights, then the eight sums.  Slots j >= 16 C - 32 nt are the zero columns that pad the last N-tile: not written.
template <int C>
__device__ __forceinline__ void wgrad_thin_reduce_body(int blk_x, const float* __restrict__ ws, float* __restrict__ dw,
                                                       float* __restrict__ db, int bias_from_big, int nblk) {
  constexpr int NT = (16 * C + 31) / 32;
  constexpr int STRIDE = NT * 1024 + 32 + NT * 32;
  constexpr int NB = NT * 64;                        // workgroups reducing dw; two more reduce db
  if (blk_x >= NB) {
    if (db) wgrad_thin_bias_reduce<C>(ws, db, bias_from_big, nblk, blk_x - NB);
    return;
  }
  __shared__ __attribute__((aligned(16))) float red[64][16];
  __shared__ float red2[8][16];
  const int p = threadIdx.x & 3, gq = threadIdx.x >> 2;
  const float* src = ws + (long)gq * STRIDE + (blk_x * 4 + p) * 4;
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {                      // partial gq + 64 u
    v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (gq + 64 * u < nblk) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(64 * u) * STRIDE);
  }
#pragma unroll
  for (int w = 1; w < 8; w *= 2)
#pragma unroll
    for (int u = 0; u < 8; u += 2 * w) v[u] += v[u + w];
  *reinterpret_cast<f32x4*>(&red[gq][p * 4]) = v[0];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int o = threadIdx.x & 15, part = threadIdx.x >> 4;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[part * 8 + k][o];
    red2[part][o] = t;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int o = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red2[k][o];
    const int q = blk_x * 16 + o;                    // workspace slot [nt][cs][j]
    const int nt = q >> 10, cs = (q >> 5) & 31, nidx = nt * 32 + (q & 31);   // nidx = cb * 16 + tap
    if (nidx < 16 * C) dw[cs * 16 * C + nidx] = t;   // dw[cs][cb][tap]
  }
}


}  // namespace dvae
