// Shape-generic direct kernels for the k4/s2/p1 conv family (any channel count, any H/W,
// NCHW or NHWC).  They serve the shapes the tuned gfx950 kernels do not cover (e.g. the
// 32x32 MNIST geometry, odd batch remainders) and as the on-device A/B reference for the
// MFMA kernels (DVAE_FORCE_GENERIC=1).  Reference semantics: torch.nn.Conv2d /
// ConvTranspose2d as used in disvae/models/encoders.py:54-60 and decoders.py:57-65.
#include "common.h"

namespace dvae {

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == DVAE_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DVAE_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  if (act == DVAE_ACT_LEAKY02) return v > 0.f ? v : 0.2f * v;
  return v;
}

// exact sigmoid used for the final layer (torch.sigmoid = 1/(1+exp(-x)) in fp32)
__device__ __forceinline__ float sigmoid_exact(float v) { return 1.f / (1.f + expf(-v)); }

__global__ void k_down_generic(const float* __restrict__ big, Strides sb, const float* __restrict__ w,
                               const float* __restrict__ bias, const float* __restrict__ mask,
                               float* __restrict__ out, Strides so, int out_nhwc, int N, int Cb, int Cs,
                               int Hs, int Ws, int act) {
  long total = (long)N * Cs * Hs * Ws;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    int n, cs, sy, sx;
    long r = idx;
    if (out_nhwc) { cs = r % Cs; r /= Cs; sx = r % Ws; r /= Ws; sy = r % Hs; n = r / Hs; }
    else { sx = r % Ws; r /= Ws; sy = r % Hs; r /= Hs; cs = r % Cs; n = r / Cs; }
    float acc = bias ? bias[cs] : 0.f;
    const int Hb = 2 * Hs, Wb = 2 * Ws;
    for (int cb = 0; cb < Cb; ++cb) {
      const float* wp = w + ((long)cs * Cb + cb) * 16;
      const float* bp = big + n * sb.n + cb * sb.c;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        int by = 2 * sy - 1 + kh;
        if (by < 0 || by >= Hb) continue;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          int bx = 2 * sx - 1 + kw;
          if (bx < 0 || bx >= Wb) continue;
          acc = fmaf(bp[by * sb.h + bx * sb.w], wp[kh * 4 + kw], acc);
        }
      }
    }
    long o = n * so.n + cs * so.c + sy * so.h + sx * so.w;
    if (act == DVAE_ACT_SIGMOID) acc = sigmoid_exact(acc); else acc = apply_act(acc, act);
    if (mask) acc = mask[o] > 0.f ? acc : 0.f;
    out[o] = acc;
  }
}

__global__ void k_up_generic(const float* __restrict__ small, Strides ss, const float* __restrict__ w,
                             const float* __restrict__ bias, const float* __restrict__ mask,
                             float* __restrict__ out, Strides so, int out_nhwc, int N, int Cb, int Cs,
                             int Hs, int Ws, int act) {
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  long total = (long)N * Cb * Hb * Wb;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    int n, cb, by, bx;
    long r = idx;
    if (out_nhwc) { cb = r % Cb; r /= Cb; bx = r % Wb; r /= Wb; by = r % Hb; n = r / Hb; }
    else { bx = r % Wb; r /= Wb; by = r % Hb; r /= Hb; cb = r % Cb; n = r / Cb; }
    float acc = bias ? bias[cb] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      int t = by + 1 - kh;
      if (t < 0 || (t & 1)) continue;
      int sy = t >> 1;
      if (sy >= Hs) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        int u = bx + 1 - kw;
        if (u < 0 || (u & 1)) continue;
        int sx = u >> 1;
        if (sx >= Ws) continue;
        const float* sp = small + n * ss.n + sy * ss.h + sx * ss.w;
        const float* wp = w + (long)cb * 16 + kh * 4 + kw;
        for (int cs = 0; cs < Cs; ++cs) acc = fmaf(sp[cs * ss.c], wp[(long)cs * Cb * 16], acc);
      }
    }
    long o = n * so.n + cb * so.c + by * so.h + bx * so.w;
    if (act == DVAE_ACT_SIGMOID) acc = sigmoid_exact(acc); else acc = apply_act(acc, act);
    if (mask) acc = mask[o] > 0.f ? acc : 0.f;
    out[o] = acc;
  }
}

// ---- the thin ends at ANY spatial size (round 6) -------------------------------------------------------------------------
// conv1 (Cb -> 32), convT3's input gradient (the same product with a mask) and convT3 forward (32 -> Cb) of images the tuned
// 64x64 kernels do not cover -- the 32x32 geometry of BASELINE configs[0] -- took 21 / 21 / 44 us per launch at 64 images on
// the kernels above: one thread per output ELEMENT, every operand a strided scalar load from global memory
// (profiles/r05_v33_mnist_generic_wgrad.txt).  Here the weights sit in LDS, a thread of the down kernel owns 4 consecutive output
// channels of a pixel (its 16 Cb input values loaded once, one 16-byte store) and a thread of the up kernel reads its <= 4
// contributing pixels as 16-byte chunks.  Every output is the SAME fmaf chain as in k_down_generic / k_up_generic (bias, then
// cb / kh / kw resp. kh / kw / cs ascending): results are bit-identical (DVAE_FORCE_GENERIC=1 still selects the plain kernels:
// tests/test_gpu_kernels.py::test_thin_ends_at_any_size_match_the_plain_generic_kernels).
template <int CB>
__global__ __launch_bounds__(256) void k_down_thin_px(const float* __restrict__ big, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ mask,
                                                      float* __restrict__ out, int N, int Hs, int Ws, int act) {
  __shared__ float wl[32 * CB * 16];               // w[cs][cb][kh][kw] as it is
  for (int e = threadIdx.x; e < 32 * CB * 16; e += 256) wl[e] = w[e];
  __syncthreads();
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const long total = (long)N * Hs * Ws * 8;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int q = idx & 7;
    long r = idx >> 3;
    const int sx = r % Ws; r /= Ws;
    const int sy = r % Hs;
    const int n = r / Hs;
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bias ? bias[4 * q + j] : 0.f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const float* bp = big + ((long)n * CB + cb) * Hb * Wb;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const int by = 2 * sy - 1 + kh;
        if (by < 0 || by >= Hb) continue;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const int bx = 2 * sx - 1 + kw;
          if (bx < 0 || bx >= Wb) continue;
          const float v = bp[by * Wb + bx];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, wl[((4 * q + j) * CB + cb) * 16 + kh * 4 + kw], acc[j]);
        }
      }
    }
    const long o = (((long)n * Hs + sy) * Ws + sx) * 32 + 4 * q;
    f32x4 res;
#pragma unroll
    for (int j = 0; j < 4; ++j) res[j] = act == DVAE_ACT_SIGMOID ? sigmoid_exact(acc[j]) : apply_act(acc[j], act);
    if (mask) {
      const f32x4 m = *reinterpret_cast<const f32x4*>(mask + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) res[j] = m[j] > 0.f ? res[j] : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + o) = res;
  }
}

template <int CB>
__global__ __launch_bounds__(256) void k_up_thin_px(const float* __restrict__ small, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ mask,
                                                    float* __restrict__ out, Strides so, int out_nhwc, int N, int Hs, int Ws,
                                                    int act) {
  __shared__ __attribute__((aligned(16))) float wl[CB * 16 * 32];   // [cb][tap][cs]
  for (int e = threadIdx.x; e < 32 * CB * 16; e += 256) {
    const int cs = e / (CB * 16), rem = e % (CB * 16);              // w[cs][cb][tap]: rem = cb * 16 + tap
    wl[rem * 32 + cs] = w[e];
  }
  __syncthreads();
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  const long total = (long)N * CB * Hb * Wb;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int n, cb, by, bx;
    long r = idx;
    if (out_nhwc) { cb = r % CB; r /= CB; bx = r % Wb; r /= Wb; by = r % Hb; n = r / Hb; }
    else { bx = r % Wb; r /= Wb; by = r % Hb; r /= Hb; cb = r % CB; n = r / CB; }
    float acc = bias ? bias[cb] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int t = by + 1 - kh;
      if (t < 0 || (t & 1)) continue;
      const int sy = t >> 1;
      if (sy >= Hs) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int u = bx + 1 - kw;
        if (u < 0 || (u & 1)) continue;
        const int sx = u >> 1;
        if (sx >= Ws) continue;
        const float* sp = small + (((long)n * Hs + sy) * Ws + sx) * 32;
        const float* wp = wl + (cb * 16 + kh * 4 + kw) * 32;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sp + 4 * c4);
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + 4 * c4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = fmaf(sv[j], wv[j], acc);
        }
      }
    }
    const long o = n * so.n + cb * so.c + by * so.h + bx * so.w;
    if (act == DVAE_ACT_SIGMOID) acc = sigmoid_exact(acc); else acc = apply_act(acc, act);
    if (mask) acc = mask[o] > 0.f ? acc : 0.f;
    out[o] = acc;
  }
}

// one block per (cs, cb): 16 taps reduced over all (n, sy, sx)
__global__ __launch_bounds__(256) void k_wgrad_generic(const float* __restrict__ big, Strides sb,
                                                       const float* __restrict__ small, Strides ss,
                                                       float* __restrict__ dw, int N, int Cb, int Cs,
                                                       int Hs, int Ws) {
  const int cs = blockIdx.x / Cb, cb = blockIdx.x % Cb;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  long total = (long)N * Hs * Ws;
  for (long p = threadIdx.x; p < total; p += blockDim.x) {
    int sx = p % Ws; long r = p / Ws; int sy = r % Hs; int n = r / Hs;
    float sv = small[n * ss.n + cs * ss.c + sy * ss.h + sx * ss.w];
    const float* bp = big + n * sb.n + cb * sb.c;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      int by = 2 * sy - 1 + kh;
      if (by < 0 || by >= Hb) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        int bx = 2 * sx - 1 + kw;
        if (bx < 0 || bx >= Wb) continue;
        acc[kh * 4 + kw] = fmaf(sv, bp[by * sb.h + bx * sb.w], acc[kh * 4 + kw]);
      }
    }
  }
  __shared__ float red[4][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float v = wave_sum(acc[t]);
    if (lane == 0) red[wv][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16)
    dw[((long)cs * Cb + cb) * 16 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// The same reduction with the positions split over gridDim.y chunks (round 5): one workgroup per (cs, cb) pair leaves most of
// the chip idle and walks 16 K positions per thread-block at the 32x32 MNIST geometry (152 us per launch at 64 images, plus
// 56 us of k_chansum on 32 -- or ONE -- workgroups: 60 % of that configuration's iteration).  Per-chunk partial sums
// [chunk][pair][17] (16 taps + the pair's share of the bias gradient) go to the caller's workspace; k_wgrad_generic_fin adds
// the chunks in order.  The bias sum rides on values the workgroup loads anyway: pair (cs, 0) sums the small-side channel cs
// (Conv2d), pair (0, cb) the big-side channel cb through the taps kh, kw in {1, 2}, which visit every big pixel exactly once
// and never leave the image (ConvTranspose2d).
#define WGG_SLOTS 17
__global__ __launch_bounds__(256) void k_wgrad_generic_part(const float* __restrict__ big, Strides sb,
                                                            const float* __restrict__ small, Strides ss,
                                                            float* __restrict__ part, int N, int Cb, int Cs, int Hs, int Ws,
                                                            int bias_from_big) {
  const int cs = blockIdx.x / Cb, cb = blockIdx.x % Cb;
  const int Hb = 2 * Hs, Wb = 2 * Ws;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  float bsum = 0.f;
  const long total = (long)N * Hs * Ws;
  const long chunk = (total + gridDim.y - 1) / gridDim.y;
  const long p0 = blockIdx.y * chunk, p1 = p0 + chunk < total ? p0 + chunk : total;
  for (long p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    int sx = p % Ws; long r = p / Ws; int sy = r % Hs; int n = r / Hs;
    float sv = small[n * ss.n + cs * ss.c + sy * ss.h + sx * ss.w];
    if (!bias_from_big) bsum += sv;
    const float* bp = big + n * sb.n + cb * sb.c;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      int by = 2 * sy - 1 + kh;
      if (by < 0 || by >= Hb) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        int bx = 2 * sx - 1 + kw;
        if (bx < 0 || bx >= Wb) continue;
        const float bv = bp[by * sb.h + bx * sb.w];
        acc[kh * 4 + kw] = fmaf(sv, bv, acc[kh * 4 + kw]);
        if (bias_from_big && (kh == 1 || kh == 2) && (kw == 1 || kw == 2)) bsum += bv;
      }
    }
  }
  __shared__ float red[4][WGG_SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float v = wave_sum(acc[t]);
    if (lane == 0) red[wv][t] = v;
  }
  {
    float v = wave_sum(bsum);
    if (lane == 0) red[wv][16] = v;
  }
  __syncthreads();
  if (threadIdx.x < WGG_SLOTS)
    part[((long)blockIdx.y * gridDim.x + blockIdx.x) * WGG_SLOTS + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void k_wgrad_generic_fin(const float* __restrict__ part, int chunks, int npairs, int Cb, int Cs,
                                    float* __restrict__ dw, float* __restrict__ db, int bias_from_big) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < npairs * 16) {
    const int pair = idx >> 4, t = idx & 15;
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += part[((long)c * npairs + pair) * WGG_SLOTS + t];
    dw[idx] = v;
  }
  const int nb = bias_from_big ? Cb : Cs;
  if (db && idx < nb) {
    const int pair = bias_from_big ? idx : idx * Cb;      // (cs = 0, cb = idx)  |  (cs = idx, cb = 0)
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += part[((long)c * npairs + pair) * WGG_SLOTS + 16];
    db[idx] = v;
  }
}

// db[c] = sum over n,h,w of t[n,c,h,w]; one block per channel
__global__ __launch_bounds__(256) void k_chansum(const float* __restrict__ t, Strides st, float* __restrict__ db,
                                                 int N, int H, int W) {
  const int c = blockIdx.x;
  float acc = 0.f;
  long total = (long)N * H * W;
  for (long p = threadIdx.x; p < total; p += blockDim.x) {
    int x = p % W; long r = p / W; int y = r % H; int n = r / H;
    acc += t[n * st.n + c * st.c + y * st.h + x * st.w];
  }
  __shared__ float red[4];
  float v = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) db[c] = red[0] + red[1] + red[2] + red[3];
}

static inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

int launch_down_generic(const ConvArgs& a, hipStream_t s) {
  Strides sb = make_strides(a.big_layout, a.Cb, 2 * a.Hs, 2 * a.Ws);
  Strides so = make_strides(a.out_layout, a.Cs, a.Hs, a.Ws);
  long total = (long)a.N * a.Cs * a.Hs * a.Ws;
#ifndef THIN_PX_OFF   // (variant builds: the A/B partner)
  if (!use_generic_only() && a.Cs == 32 && (a.Cb == 1 || a.Cb == 3) && a.big_layout == DVAE_NCHW && a.out_layout == DVAE_NHWC &&
      (((uintptr_t)a.out | (uintptr_t)a.mask) & 15) == 0) {
    const int grid = grid_for((long)a.N * a.Hs * a.Ws * 8, 256);
    if (a.Cb == 1) hipLaunchKernelGGL(k_down_thin_px<1>, dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.Hs, a.Ws, a.act);
    else hipLaunchKernelGGL(k_down_thin_px<3>, dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.Hs, a.Ws, a.act);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
#endif
  hipLaunchKernelGGL(k_down_generic, dim3(grid_for(total, 256)), dim3(256), 0, s, a.big, sb, a.w, a.bias,
                     a.mask, a.out, so, a.out_layout == DVAE_NHWC, a.N, a.Cb, a.Cs, a.Hs, a.Ws, a.act);
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_generic(const ConvArgs& a, hipStream_t s) {
  Strides ss = make_strides(a.small_layout, a.Cs, a.Hs, a.Ws);
  Strides so = make_strides(a.out_layout, a.Cb, 2 * a.Hs, 2 * a.Ws);
  long total = (long)a.N * a.Cb * 4 * a.Hs * a.Ws;
#ifndef THIN_PX_OFF
  if (!use_generic_only() && a.Cs == 32 && (a.Cb == 1 || a.Cb == 3) && a.small_layout == DVAE_NHWC && ((uintptr_t)a.small & 15) == 0) {
    const int grid = grid_for(total, 256);
    if (a.Cb == 1) hipLaunchKernelGGL(k_up_thin_px<1>, dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.mask, a.out, so, a.out_layout == DVAE_NHWC, a.N, a.Hs, a.Ws, a.act);
    else hipLaunchKernelGGL(k_up_thin_px<3>, dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.mask, a.out, so, a.out_layout == DVAE_NHWC, a.N, a.Hs, a.Ws, a.act);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
#endif
  hipLaunchKernelGGL(k_up_generic, dim3(grid_for(total, 256)), dim3(256), 0, s, a.small, ss, a.w, a.bias,
                     a.mask, a.out, so, a.out_layout == DVAE_NHWC, a.N, a.Cb, a.Cs, a.Hs, a.Ws, a.act);
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_wgrad_generic(const float* big, int big_layout, const float* small, int small_layout, float* dw,
                         float* db, int bias_from_big, int N, int Cb, int Cs, int Hs, int Ws, float* ws, size_t ws_floats,
                         hipStream_t s) {
  Strides sb = make_strides(big_layout, Cb, 2 * Hs, 2 * Ws);
  Strides ss = make_strides(small_layout, Cs, Hs, Ws);
  // positions split over chunks when the caller lent a workspace: >= 4 positions per thread and chunk, at most 64 chunks
  const long total = (long)N * Hs * Ws, npairs = (long)Cs * Cb;
  long chunks = total / 1024;
  if (chunks > 64) chunks = 64;
  if (ws && chunks >= 2 && npairs <= 65535 && (size_t)(chunks * npairs * WGG_SLOTS) <= ws_floats) {
    hipLaunchKernelGGL(k_wgrad_generic_part, dim3((unsigned)npairs, (unsigned)chunks), dim3(256), 0, s, big, sb, small, ss, ws, N,
                       Cb, Cs, Hs, Ws, bias_from_big);
    DVAE_CHECK_LAUNCH();
    const long nthr = npairs * 16;
    hipLaunchKernelGGL(k_wgrad_generic_fin, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, ws, (int)chunks, (int)npairs,
                       Cb, Cs, dw, db, bias_from_big);
    DVAE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(k_wgrad_generic, dim3(Cs * Cb), dim3(256), 0, s, big, sb, small, ss, dw, N, Cb, Cs, Hs, Ws);
  DVAE_CHECK_LAUNCH();
  if (db) {
    if (bias_from_big)
      hipLaunchKernelGGL(k_chansum, dim3(Cb), dim3(256), 0, s, big, sb, db, N, 2 * Hs, 2 * Ws);
    else
      hipLaunchKernelGGL(k_chansum, dim3(Cs), dim3(256), 0, s, small, ss, db, N, Hs, Ws);
    DVAE_CHECK_LAUNCH();
  }
  return 0;
}

__global__ void k_relayout(const float* __restrict__ src, Strides ssrc, float* __restrict__ dst, Strides sdst,
                           int dst_nhwc, int N, int C, int H, int W) {
  long total = (long)N * C * H * W;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    int n, c, y, x;
    long r = idx;
    if (dst_nhwc) { c = r % C; r /= C; x = r % W; r /= W; y = r % H; n = r / H; }
    else { x = r % W; r /= W; y = r % H; r /= H; c = r % C; n = r / C; }
    dst[n * sdst.n + c * sdst.c + y * sdst.h + x * sdst.w] = src[n * ssrc.n + c * ssrc.c + y * ssrc.h + x * ssrc.w];
  }
}

int launch_relayout(const float* src, int src_layout, float* dst, int N, int C, int H, int W, hipStream_t s) {
  int dst_layout = src_layout == DVAE_NHWC ? DVAE_NCHW : DVAE_NHWC;
  Strides a = make_strides(src_layout, C, H, W), b = make_strides(dst_layout, C, H, W);
  long total = (long)N * C * H * W;
  hipLaunchKernelGGL(k_relayout, dim3(grid_for(total, 256)), dim3(256), 0, s, src, a, dst, b,
                     dst_layout == DVAE_NHWC, N, C, H, W);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
