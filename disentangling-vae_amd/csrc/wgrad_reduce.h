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
#define WG_REDUCE_BLOCKS (1024 + 2)        // workgroups of the 32-channel reduction (the last two: bias)
#define WT_MAX_BLOCKS 512
#define WT_REDUCE_BLOCKS(C) ((32 * 16 * (C) + 15) / 16 + 2)

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

// 1024 workgroups x (16 outputs x 16 partial-groups), 8 loads in flight per lane, fixed order
__device__ __forceinline__ void wgrad32_reduce_body(int blk_x, const float* __restrict__ ws, float* __restrict__ dw,
                                                    float* __restrict__ db, int bias_from_big, int nblk) {
  if (blk_x >= 1024) {                               // the last two workgroups reduce the bias gradient
    if (db) wgrad32_bias_reduce(ws, db, bias_from_big, nblk, blk_x - 1024);
    return;
  }
  __shared__ float red[16][16];
  const int o = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int idx = blk_x * 16 + o;                  // (tap, cs, cb)
  float pv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) pv[u] = 0.f;
  int g = gq;
  for (; g + 112 < nblk; g += 128) {
#pragma unroll
    for (int u = 0; u < 8; ++u) pv[u] += ws[(long)(g + 16 * u) * WG_STRIDE + idx];
  }
  for (; g < nblk; g += 16) pv[0] += ws[(long)g * WG_STRIDE + idx];
  red[gq][o] = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
  __syncthreads();
  if (gq == 0) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][o];
    const int tap = idx >> 10, cs = (idx >> 5) & 31, cb = idx & 31;
    dw[(cs * 32 + cb) * 16 + tap] = v;
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

// 16 outputs x 16 partial-groups per workgroup; 8 loads in flight per lane; fixed summation order
template <int C>
__device__ __forceinline__ void wgrad_thin_reduce_body(int blk_x, const float* __restrict__ ws, float* __restrict__ dw,
                                                       float* __restrict__ db, int bias_from_big, int nblk) {
  constexpr int NT = (16 * C + 31) / 32;
  constexpr int STRIDE = NT * 1024 + 32 + NT * 32;
  constexpr int NB = (32 * 16 * C + 15) / 16;        // workgroups reducing dw; two more reduce db
  if (blk_x >= NB) {
    if (db) wgrad_thin_bias_reduce<C>(ws, db, bias_from_big, nblk, blk_x - NB);
    return;
  }
  __shared__ float red[16][16];
  const int o = threadIdx.x & 15, gq = threadIdx.x >> 4;
  // dw[cs][cb][tap] : element idx = cs * 16C + nidx, nidx = cb*16 + tap
  const int idx = blk_x * 16 + o;
  float pv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) pv[u] = 0.f;
  if (idx < 32 * 16 * C) {
    const int cs = idx / (16 * C), nidx = idx % (16 * C);
    const int off = (nidx >> 5) * 1024 + cs * 32 + (nidx & 31);
    int g = gq;
    for (; g + 112 < nblk; g += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[u] += ws[(long)(g + 16 * u) * STRIDE + off];
    }
    for (; g < nblk; g += 16) pv[0] += ws[(long)g * STRIDE + off];
  }
  red[gq][o] = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
  __syncthreads();
  if (gq == 0 && idx < 32 * 16 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][o];
    dw[idx] = t;
  }
}


}  // namespace dvae
