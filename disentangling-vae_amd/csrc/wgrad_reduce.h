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
#define WG_REDUCE_BLOCKS (256 + 4)         // workgroups of the 32-channel reduction (the last four: bias, 8 channels each)
#define WT_MAX_BLOCKS 512
// thin layers: the whole partial buffer (NT x 1024 weight slots + 32 + NT x 32 bias slots), 16 slots per workgroup
#define WT_REDUCE_BLOCKS(C) ((((16 * (C) + 31) / 32) * (1024 + 32) + 32) / 16)

// bias gradient: 4 workgroups x (8 channels x 32 partial-groups); every lane has its 8 partials (x 4 slots when the
// bias comes from the big side) in flight at once
template <bool LEAN>
__device__ __forceinline__ void wgrad32_bias_reduce(const float* __restrict__ ws, float* __restrict__ db,
                                                    int bias_from_big, int nblk, int blk) {
  __shared__ float redb[32][8];
  const int o = threadIdx.x & 7, gq = threadIdx.x >> 3;
  const int c = blk * 8 + o;
  const float* wsb = ws + 16384 + c;
  float v[8];
  if (LEAN) {
    // one partial at a time, the same tree ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)): a handful of registers
    float e = 0.f, pr = 0.f, pp = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = gq + 32 * u;
      const float* q = wsb + (long)(g < nblk ? g : nblk - 1) * WG_STRIDE;
      float t;
      if (bias_from_big) t = (q[32] + q[64]) + (q[96] + q[128]);
      else t = q[0];
      if (g >= nblk) t = 0.f;
      if ((u & 1) == 0) e = t; else pr = e + t;
      if ((u & 3) == 1) pp = pr;
      if ((u & 3) == 3) { if (u == 3) q0 = pp + pr; else q1 = pp + pr; }
      __builtin_amdgcn_sched_barrier(0);
    }
    redb[gq][o] = q0 + q1;
  } else if (bias_from_big) {
    float a[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                    // partial gq + 32 u, clamped (zeroed below)
      const int g = gq + 32 * u;
      const float* q = wsb + (long)(g < nblk ? g : nblk - 1) * WG_STRIDE;
      a[u][0] = q[32]; a[u][1] = q[64]; a[u][2] = q[96]; a[u][3] = q[128];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]);
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = gq + 32 * u;
      v[u] = wsb[(long)(g < nblk ? g : nblk - 1) * WG_STRIDE];
    }
  }
  if (!LEAN) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (gq + 32 * u >= nblk) v[u] = 0.f;
    redb[gq][o] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    float t = 0.f;
    if (LEAN) {
#pragma unroll 4
      for (int k = 0; k < 32; ++k) t += redb[k][o];
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) t += redb[k][o];
    }
    db[c] = t;
  }
}

// 256 workgroups x (16 positions of 16 bytes = 64 outputs) x 16 partial-groups.  Every lane has ALL its partials in flight
// at once (16 x 16-byte loads, a wave instruction covers 4 partials x 256 contiguous bytes): the reduction is one round trip
// to L2 / HBM instead of two rounds of 4-byte loads.  Fixed summation order: tree over u per lane, then groups 0..15.
// LEAN (steps of >= WGR_LEAN_MIN_IMAGES images): the four quarters of the summation tree one after the other -- 4 loads in flight
// instead of 16, 34 VGPRs instead of 74 (the bias sums one partial at a time) -- so that a reduction workgroup FITS beside the other stream's persistent kernels
// (k_up32ws<8>: 48 free VGPRs per SIMD, k_down32dma: 64-80) instead of waiting for their workgroups to leave: inside the
// 1024-image step these launches took 22 us on average, one of them 75 (profiles/r06_final4_b1024_timeline.md), alone 6.  The
// same tree, bit-identical sums.  1024 images 1.0405 -> 1.0288 ms; at 256 / 512 images, where nothing blocks the launch and its
// own four round trips count, +2.4 / +0.8 %: not used there (profiles/r06_s2_lean.txt).  Two loads in flight (28 VGPRs: fits beside
// k_up32ws<16> as well): 1.0364 against 1.0306 ms (profiles/r06_s2_lean2.txt): not used.
#define WGR_LEAN_MIN_IMAGES 768
template <bool LEAN>
__device__ __forceinline__ void wgrad32_reduce_body(int blk_x, const float* __restrict__ ws, float* __restrict__ dw,
                                                    float* __restrict__ db, int bias_from_big, int nblk) {
  if (blk_x >= 256) {                                // the last four workgroups reduce the bias gradient
    if (db) wgrad32_bias_reduce<LEAN>(ws, db, bias_from_big, nblk, blk_x - 256);
    return;
  }
  __shared__ __attribute__((aligned(16))) float red[16][64];
  const int p = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const float* src = ws + (blk_x * 16 + p) * 4;
  if constexpr (LEAN) {
  f32x4 pair[2];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    f32x4 qa;
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      const int hh = 2 * pp + hq;
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int g = gq + 16 * (4 * hh + u);
        v[u] = *reinterpret_cast<const f32x4*>(src + (long)(g < nblk ? g : nblk - 1) * WG_STRIDE);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (gq + 16 * (4 * hh + u) >= nblk) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 q = (v[0] + v[1]) + (v[2] + v[3]);
      if (hq == 0) qa = q; else pair[pp] = qa + q;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  *reinterpret_cast<f32x4*>(&red[gq][p * 4]) = pair[0] + pair[1];
  } else {
  f32x4 v[16];
  // unconditional loads from a clamped partial index, zeroed afterwards: a branch per load would serialise them
#pragma unroll
  for (int u = 0; u < 16; ++u) {                     // partial gq + 16 u  (nblk <= WG_MAX_BLOCKS = 256)
    const int g = gq + 16 * u;
    v[u] = *reinterpret_cast<const f32x4*>(src + (long)(g < nblk ? g : nblk - 1) * WG_STRIDE);
  }
#pragma unroll
  for (int u = 0; u < 16; ++u)
    if (gq + 16 * u >= nblk) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 1; w < 16; w *= 2)
#pragma unroll
    for (int u = 0; u < 16; u += 2 * w) v[u] += v[u + w];
  *reinterpret_cast<f32x4*>(&red[gq][p * 4]) = v[0];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int o = threadIdx.x;
    float t = 0.f;
    if (LEAN) {
#pragma unroll 4
      for (int k = 0; k < 16; ++k) t += red[k][o];
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][o];
    }
    const int idx = blk_x * 64 + o;                  // (tap, cs, cb)
    const int tap = idx >> 10, cs = (idx >> 5) & 31, cb = idx & 31;
    dw[(cs * 32 + cb) * 16 + tap] = t;
  }
}


// Workgroup = 4 positions of 16 bytes (16 slots of the partial buffer: NT x 1024 weight slots [nt][cs][j], then 32 sums of the
// small side per cs, then NT x 32 column sums of the big side) x 64 partial-groups: all 8 loads of a lane in flight at once
// (nblk <= WT_MAX_BLOCKS = 512), one round trip.  Fixed summation order: tree over u per lane, groups in eights, then the
// eight sums.  Weight slots j >= 16 C - 32 nt are the zero columns that pad the last N-tile: not written.
template <int C>
__device__ __forceinline__ void wgrad_thin_reduce_body(int blk_x, const float* __restrict__ ws, float* __restrict__ dw,
                                                       float* __restrict__ db, int bias_from_big, int nblk) {
  constexpr int NT = (16 * C + 31) / 32;
  constexpr int STRIDE = NT * 1024 + 32 + NT * 32;
  constexpr int QB = NT * 1024;                      // first bias slot
  __shared__ __attribute__((aligned(16))) float red[64][16];
  __shared__ float red2[8][16];
  __shared__ float fin[16];
  const int p = threadIdx.x & 3, gq = threadIdx.x >> 2;
  const float* src = ws + (blk_x * 4 + p) * 4;
  f32x4 v[8];
  // unconditional loads from a clamped partial index, zeroed afterwards: a branch per load would serialise them
#pragma unroll
  for (int u = 0; u < 8; ++u) {                      // partial gq + 64 u
    const int g = gq + 64 * u;
    v[u] = *reinterpret_cast<const f32x4*>(src + (long)(g < nblk ? g : nblk - 1) * STRIDE);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (gq + 64 * u >= nblk) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
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
  float t = 0.f;
  const int o = threadIdx.x & 15;
  if (threadIdx.x < 16) {
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red2[k][o];
  }
  const int q0 = blk_x * 16;                         // first slot of this workgroup (uniform)
  if (q0 < QB) {
    if (threadIdx.x < 16) {
      const int q = q0 + o;                          // weight slot [nt][cs][j]
      const int nt = q >> 10, cs = (q >> 5) & 31, nidx = nt * 32 + (q & 31);   // nidx = cb * 16 + tap
      if (nidx < 16 * C) dw[cs * 16 * C + nidx] = t; // dw[cs][cb][tap]
    }
    return;
  }
  if (!db) return;
  const int k = (q0 - QB) >> 4;                      // 16-slot chunk of the bias region
  if (!bias_from_big) {
    // chunks 0, 1: sum of the small side per cs
    if (k < 2 && threadIdx.x < 16) db[k * 16 + o] = t;
    return;
  }
  // chunk 2 + cb holds the 16 (kh,kw) column sums of big-side channel cb, of which taps (kh,kw) in {1,2}x{1,2} = 5, 6, 9, 10
  // cover every big pixel exactly once
  if (k < 2 || k - 2 >= C) return;
  if (threadIdx.x < 16) fin[o] = t;
  __syncthreads();
  if (threadIdx.x == 0) db[k - 2] = (fin[5] + fin[6]) + (fin[9] + fin[10]);
}


}  // namespace dvae
