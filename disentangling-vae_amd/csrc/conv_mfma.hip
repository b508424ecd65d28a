// gfx950 MFMA kernels for the 32 <-> 32 channel k4/s2/p1 convolutions of the Burgess
// encoder/decoder (conv2, conv3, conv_64, convT_64, convT1, convT2 and all their dgrads and
// wgrads: 18 of the 23 conv-shaped launches of a training step and ~75 % of its FLOPs).
//
// Formulation (all fp32, exact: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain):
//   down  : small[p][cs] = sum_{tap,cb} big[pix(p,tap)][cb] * w[cs][cb][tap]   M = 32 pixels, N = 32 cs, K = 512
//   up    : big[p'][cb]  = sum_{tap in class,cs} small[nb(p',tap)][cs] * w[cs][cb][tap]   4 parity classes, K = 128
//   wgrad : dw[cs][cb][tap] = sum_p small[p][cs] * big[pix(p,tap)][cb]           M = 32 cs, N = 32 cb, K = pixels
// Activations are NHWC (a pixel's 32 channels = one 128-byte line), so the MFMA A operand of
// down/up is read from LDS as one ds_read_b128 per 4 MFMAs (k order inside a tap is permuted
// so that lanes 0-31 / 32-63 take channels 8q+j / 8q+4+j) and the D fragment stores whole
// 128-byte pixel lines.  A workgroup (8 waves, one per CU: 64 KB of re-laid-out weights live
// in LDS) is persistent over "units" of 64 small pixels; the next unit's activation tile is
// prefetched into registers while the current one is in the MFMA loop.
// The big-side tile is stored in LDS split by column parity ([row][par][col/2][32ch]) with the
// 16-byte channel chunks XOR-swizzled, so that the stride-2 im2col reads of a lane group fall
// on distinct banks.
#include <stdlib.h>
#include "common.h"
#include "conv_mfma_common.h"
#include "wgrad_reduce.h"

namespace dvae {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// ---- down: big -> small ------------------------------------------------------------------
#define SEL4(v, g, j) ((g) == 0 ? (v)[j] : (g) == 1 ? (v)[4 + (j)] : (g) == 2 ? (v)[8 + (j)] : (v)[12 + (j)])
template <int HS, bool MASK>
__global__ __launch_bounds__(512) void k_down32(const float* __restrict__ big, const float* __restrict__ w,
                                                const float* __restrict__ bias, const float* __restrict__ mask,
                                                float* __restrict__ out, int N, int act, int n_units,
                                                int out_nchw, int w_staged) {
  using G = Geo<HS>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                    // 16384 floats
  float* bt = smem + 16384;            // G::BIG_FLOATS
  float* red = bt + G::BIG_FLOATS;     // [2 mt][4 owner][4 src][4 j][64 lanes] = 8192 floats
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mt = wv & 1, kh = wv >> 1;
  const int i = lane & 31, h = lane >> 5;
  const int p = mt * 32 + i;
  const int img_l = p / (G::R * HS), sy_l = (p / HS) % G::R, sx = p % HS;
  const int r = 2 * sy_l + kh;

  SlotDesc<G::BIG_NPF> sd;
  init_big_slots<HS>(sd, tid);
  f32x4 pf[G::BIG_NPF];
  int unit = blockIdx.x;
  if (unit < n_units) load_big<HS>(pf, sd, big, unit, N);
  if (w_staged) copy_weight_image(w, wl, tid);
  else stage_weights<true>(w, wl, tid);
  const float bv = bias ? bias[i] : 0.f;
  const long npix = (long)N * HS * HS;

  for (; unit < n_units; unit += gridDim.x) {
    store_big<HS>(pf, sd, bt);
    __syncthreads();
    if (unit + (int)gridDim.x < n_units) load_big<HS>(pf, sd, big, unit + gridDim.x, N);

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
    // balanced reduction of the 4 kh-slices: wave (mt, kh) owns accumulator registers 4*kh .. 4*kh+3
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(((mt * 4 + (e >> 2)) * 4 + kh) * 4 + (e & 3)) * 64 + lane] = acc[e];
    __syncthreads();
    // epilogue: this wave owns D-fragment rows j + 8*kh + 4*h (j = 0..3) of its M-tile.  Mask
    // loads are issued together before any store (no load->wait->store chains).
    const long P0 = (long)unit * G::U + mt * 32 + 8 * kh + 4 * h;
    // element (pixel pix, channel i) of out / mask: NHWC, or NCHW = the (c,h,w) flatten order the FC stack
    // consumes (encoders.py:80) -- written directly, no relayout pass in between
    auto oidx = [&](long pix) -> long {
      constexpr int PP = HS * HS;
      return out_nchw ? ((pix / PP) * 32 + i) * PP + (pix % PP) : pix * 32 + i;
    };
    float vals[4], mv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = SEL4(acc, kh, j);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s != kh) v += red[(((mt * 4 + kh) * 4 + s) * 4 + j) * 64 + lane];
      vals[j] = v;
    }
    const bool full = P0 - 4 * h - 8 * kh - mt * 32 + G::U <= npix;   // wave-uniform: whole unit inside the tensor
    if (full) {
      if (MASK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mv[j] = mask[oidx(P0 + j)];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = epilogue_act(vals[j] + bv, act);
        if (MASK) v = mv[j] > 0.f ? v : 0.f;
        out[oidx(P0 + j)] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long pix = P0 + j;
        if (pix < npix) {
          float v = epilogue_act(vals[j] + bv, act);
          if (MASK) v = mask[oidx(pix)] > 0.f ? v : 0.f;
          out[oidx(pix)] = v;
        }
      }
    }
  }
}

// (HS = 16, 8 of the down direction: k_down32dma, conv_down_dma.hip.  Its predecessor k_down32ws -- 64 KB weight image in LDS,
// register-staged loader waves, D fragments stored with 4-byte stores -- measured 74.5 / 80.8 us against 72.4 / 78.8 us on
// the same box at B = 1024, profiles/r03_v7_downab.txt; git history.)

// ---- up: small -> big --------------------------------------------------------------------
// output offsets of the 16 D-fragment rows of this wave's (class, M-tile) for a given unit
template <int HS>
__device__ __forceinline__ void up_offsets(long (&offs)[16], int unit, int mt, int py, int px, int h, int i) {
  using G = Geo<HS>;
  const long P0 = (long)unit * G::U;
  const int n0 = (int)(P0 / (HS * HS));
  const int sy0 = (int)(P0 % (HS * HS)) / HS;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int rowp = (e & 3) + 8 * (e >> 2) + 4 * h;
    const int pp = mt * 32 + rowp;
    const int im = pp / (G::R * HS), mm = (pp / HS) % G::R, ll = pp % HS;
    const int by = 2 * (sy0 + mm) + py, bx = 2 * ll + px;
    offs[e] = (((long)(n0 + im) * G::HB + by) * G::HB + bx) * 32 + i;
  }
}

template <int HS>
__device__ __forceinline__ void up_store(const float (&vals)[16], float* __restrict__ out, int unit, int N, int mt,
                                         int py, int px, int h, int i) {
  using G = Geo<HS>;
  long offs[16];
  up_offsets<HS>(offs, unit, mt, py, px, h, i);
  const int n0 = (int)(((long)unit * G::U) / (HS * HS));
  if (n0 + G::IMGS <= N) {          // wave-uniform: the whole unit lies inside the tensor
#pragma unroll
    for (int e = 0; e < 16; ++e) out[offs[e]] = vals[e];
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int im = (mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) / (G::R * HS);
      if (n0 + im < N) out[offs[e]] = vals[e];
    }
  }
}

// Software pipeline per workgroup:  [LDS tile(u) <- regs] | barrier | issue loads tile(u+1), mask(u) |
// store results(u-1) | MFMA(u) | results(u) -> regs.  The global stores of unit u-1 and the loads
// of unit u+1 are a whole MFMA phase old when the next iteration waits on vmcnt.
template <int HS, bool MASK>
__global__ __launch_bounds__(512) void k_up32(const float* __restrict__ small, const float* __restrict__ w,
                                              const float* __restrict__ bias, const float* __restrict__ mask,
                                              float* __restrict__ out, int N, int act, int n_units,
                                              int small_nchw, int w_staged) {
  using G = Geo<HS>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;              // 16384 floats
  float* st = smem + 16384;      // G::SH_FLOATS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cls = wv & 3, mt = wv >> 2;
  const int py = cls >> 1, px = cls & 1;
  const int i = lane & 31, h = lane >> 5;
  const int p = mt * 32 + i;
  const int img_l = p / (G::R * HS), m = (p / HS) % G::R, l = p % HS;

  SlotDesc<G::SH_NPF> sd;
  init_small_slots<HS>(sd, tid, small_nchw);
  f32x4 pf[G::SH_NPF];
  int unit = blockIdx.x;
  if (unit < n_units) load_small_halo<HS>(pf, sd, small, unit, N, small_nchw);
  if (w_staged) copy_weight_image(w, wl, tid);
  else stage_weights<false>(w, wl, tid);
  const float bv = bias ? bias[i] : 0.f;
  float vals[16];
  int prev_unit = -1;

  for (; unit < n_units; unit += gridDim.x) {
    __syncthreads();  // previous unit's reads of st are complete
    store_small_halo<HS>(pf, sd, st);
    __syncthreads();
    if (unit + (int)gridDim.x < n_units) load_small_halo<HS>(pf, sd, small, unit + gridDim.x, N, small_nchw);
    float mv[16];
    if (MASK) {       // prefetch the ReLU mask of this unit (consumed after the MFMA phase)
      long offs[16];
      up_offsets<HS>(offs, unit, mt, py, px, h, i);
      const int n0 = (int)(((long)unit * G::U) / (HS * HS));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int im = (mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) / (G::R * HS);
        mv[e] = (G::IMGS == 1 || n0 + im < N) ? mask[offs[e]] : 0.f;
      }
    }
    if (prev_unit >= 0) up_store<HS>(vals, out, prev_unit, N, mt, py, px, h, i);

    // 4 independent accumulator chains (index = ci % 4), summed after the K loop
    f32x16 accs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) accs[c][e] = 0.f;
    // 16 groups g = (ty, tx, q) of 4 MFMAs; operands of group g+1 are read before group g issues
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
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // prologue: the reads of group 0
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int cur = g & 1;
      if (g + 1 < 16) rd(g + 1, cur ^ 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) accs[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[cur][j], Bv[cur][j], accs[j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads (next group)
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMAs (this group)
    }
    const f32x16 acc = (accs[0] + accs[1]) + (accs[2] + accs[3]);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = epilogue_act(acc[e] + bv, act);
      if (MASK) v = mv[e] > 0.f ? v : 0.f;
      vals[e] = v;
    }
    prev_unit = unit;
  }
  if (prev_unit >= 0) up_store<HS>(vals, out, prev_unit, N, mt, py, px, h, i);
}

// ---- wgrad ---------------------------------------------------------------------------------
template <int HS>
__global__ __launch_bounds__(512) void k_wgrad32(const float* __restrict__ big, const float* __restrict__ small,
                                                 float* __restrict__ ws, int N, int n_units, int small_nchw) {
  using G = Geo<HS>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bt = smem;                     // G::BIG_FLOATS
  float* sp = smem + G::BIG_FLOATS;     // 64 * 32
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kh = wv >> 1, kwb = wv & 1;  // taps (kh, 2*kwb) and (kh, 2*kwb+1)
  const int i = lane & 31, h = lane >> 5;

  SlotDesc<G::BIG_NPF> sd;
  init_big_slots<HS>(sd, tid);
  f32x4 pf[G::BIG_NPF];
  f32x4 pfs;
  f32x16 acc0, acc1, acc2, acc3;   // taps (kh, 2*kwb) / (kh, 2*kwb+1), even / odd k-steps
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; acc3[e] = 0.f; }
  float sumS = 0.f, sumB0 = 0.f, sumB1 = 0.f;

  int unit = blockIdx.x;
  const long npix = (long)N * HS * HS;
  auto load_sp = [&](int u) {
    long e0 = (long)u * G::U * 32 + tid * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (e0 < npix * 32) {
      if (small_nchw) {                  // [N][32][HS*HS]: pixel e0/32, channels e0%32 .. +3, HS*HS floats apart
        const long pg = e0 >> 5;
        const float* src = small + ((pg / (HS * HS)) * 32 + (e0 & 31)) * (HS * HS) + pg % (HS * HS);
#pragma unroll
        for (int u2 = 0; u2 < 4; ++u2) v[u2] = src[u2 * HS * HS];
      } else {
        v = *reinterpret_cast<const f32x4*>(small + e0);
      }
    }
    pfs = v;
  };
  if (unit < n_units) { load_big<HS>(pf, sd, big, unit, N); load_sp(unit); }

  for (; unit < n_units; unit += gridDim.x) {
    __syncthreads();
    store_big<HS>(pf, sd, bt);
    *reinterpret_cast<f32x4*>(sp + tid * 4) = pfs;
    __syncthreads();
    if (unit + (int)gridDim.x < n_units) { load_big<HS>(pf, sd, big, unit + gridDim.x, N); load_sp(unit + gridDim.x); }

    // 16 groups of 2 k-steps (4 pixels): 6 LDS reads + 4 MFMAs; the reads of group g+1 are issued
    // before the MFMAs of group g (register ping-pong, order pinned by sched_group_barrier)
    float av[2][2], b0v[2][2], b1v[2][2];
    auto rd = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = 2 * (2 * g + u) + h;
        const int img_l = p / (G::R * HS), sy_l = (p / HS) % G::R, sx = p % HS;
        av[slot][u] = sp[p * 32 + i];
        const int r = 2 * sy_l + kh;
        const int cw = sx + kwb;
        const float* b0p = bt + (((img_l * G::BROWS + r) * 2 + 0) * G::CW + cw) * 32;
        const int sw = swz_big<HS>(r, cw);
        const int off = (((i >> 2) ^ sw) << 2) + (i & 3);
        b0v[slot][u] = b0p[off];
        b1v[slot][u] = b0p[G::CW * 32 + off];
      }
    };
    rd(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int cur = g & 1;
      if (g + 1 < 16) rd(g + 1, cur ^ 1);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][0], b0v[cur][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][0], b1v[cur][0], acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][1], b0v[cur][1], acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][1], b1v[cur][1], acc3, 0, 0, 0);
      sumS += av[cur][0] + av[cur][1]; sumB0 += b0v[cur][0] + b0v[cur][1]; sumB1 += b1v[cur][0] + b1v[cur][1];
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // 6 DS reads (next group)
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMAs (this group)
    }
  }
  acc0 += acc2; acc1 += acc3;
  // partial results of this workgroup
  float* wsw = ws + (long)blockIdx.x * WG_STRIDE;
  const int tap0 = kh * 4 + 2 * kwb;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int cs = (e & 3) + 8 * (e >> 2) + 4 * h;
    wsw[(tap0 * 32 + cs) * 32 + i] = acc0[e];
    wsw[((tap0 + 1) * 32 + cs) * 32 + i] = acc1[e];
  }
  float* wsb = wsw + 16384;   // 160 bias floats follow the 16384 weight partials
  sumS += __shfl_xor(sumS, 32, 64);
  sumB0 += __shfl_xor(sumB0, 32, 64);
  sumB1 += __shfl_xor(sumB1, 32, 64);
  if (h == 0) {
    if (wv == 0) wsb[i] = sumS;
    // taps (1,1)=5 -> wave 2 second tap, (1,2)=6 -> wave 3 first, (2,1)=9 -> wave 4 second, (2,2)=10 -> wave 5 first
    if (wv == 2) wsb[32 + i] = sumB1;
    if (wv == 3) wsb[64 + i] = sumB0;
    if (wv == 4) wsb[96 + i] = sumB1;
    if (wv == 5) wsb[128 + i] = sumB0;
  }
}

template <bool LEAN>
__global__ __launch_bounds__(256) void k_wgrad32_reduce(const float* __restrict__ ws, float* __restrict__ dw,
                                                        float* __restrict__ db, int bias_from_big, int nblk) {
  wgrad32_reduce_body<LEAN>(blockIdx.x, ws, dw, db, bias_from_big, nblk);
}

size_t wgrad32_ws_floats() { return (size_t)WG_MAX_BLOCKS * WG_STRIDE; }

// fixed-order reduction of `nblk` per-workgroup partials (also used by conv_wgrad_ws.hip, which writes the same format)
// (N = images of the step: from WGR_LEAN_MIN_IMAGES the low-register form, wgrad_reduce.h)
int launch_wgrad32_reduce(const float* ws, float* dw, float* db, int bias_from_big, int nblk, hipStream_t s, int N) {
  if (N >= WGR_LEAN_MIN_IMAGES)
    hipLaunchKernelGGL(k_wgrad32_reduce<true>, dim3(WG_REDUCE_BLOCKS), dim3(256), 0, s, ws, dw, db, bias_from_big, nblk);
  else
    hipLaunchKernelGGL(k_wgrad32_reduce<false>, dim3(WG_REDUCE_BLOCKS), dim3(256), 0, s, ws, dw, db, bias_from_big, nblk);
  DVAE_CHECK_LAUNCH();
  return 0;
}

// ---- launchers -----------------------------------------------------------------------------
static int units_for(int N, int HS) { return (int)(((long)N * HS * HS + 63) / 64); }

template <int HS>
static int launch_down_t(const ConvArgs& a, hipStream_t s) {
  using G = Geo<HS>;
  const int n_units = units_for(a.N, HS);
  const int grid = n_units < 256 ? n_units : 256;
  const size_t lds = (16384 + G::BIG_FLOATS + 8192) * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_down32<HS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_down32<HS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int out_nchw = a.out_layout == DVAE_NCHW;
  if (a.mask) hipLaunchKernelGGL((k_down32<HS, true>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, out_nchw, a.w_staged);
  else hipLaunchKernelGGL((k_down32<HS, false>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, out_nchw, a.w_staged);
  DVAE_CHECK_LAUNCH();
  return 0;
}

template <int HS>
static int launch_up_t(const ConvArgs& a, hipStream_t s) {
  using G = Geo<HS>;
  const int n_units = units_for(a.N, HS);
  const int grid = n_units < 256 ? n_units : 256;
  const size_t lds = (16384 + G::SH_FLOATS) * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_up32<HS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_up32<HS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int small_nchw = a.small_layout == DVAE_NCHW;
  if (a.mask) hipLaunchKernelGGL((k_up32<HS, true>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, small_nchw, a.w_staged);
  else hipLaunchKernelGGL((k_up32<HS, false>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, small_nchw, a.w_staged);
  DVAE_CHECK_LAUNCH();
  return 0;
}

template <int HS>
static int launch_wgrad_t(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N,
                          float* ws, hipStream_t s, int small_nchw) {
  using G = Geo<HS>;
  const int n_units = units_for(N, HS);
  // (capping the persistent grid to leave CUs to the dgrad stream was measured: 128 -> +10 % step time)
  int grid = n_units < WG_MAX_BLOCKS ? n_units : WG_MAX_BLOCKS;
  {
    static const int cap = env_int("DVAE_WGRAD_GRID", WG_MAX_BLOCKS);   // debug builds: A/B of the persistent grid size
    if (cap > 0 && cap < grid) grid = cap;
  }
  const size_t lds = (G::BIG_FLOATS + 64 * 32) * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) { (void)hipFuncSetAttribute((const void*)k_wgrad32<HS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
  hipLaunchKernelGGL(k_wgrad32<HS>, dim3(grid), dim3(512), lds, s, big, small, ws, N, n_units, small_nchw);
  DVAE_CHECK_LAUNCH();
  return launch_wgrad32_reduce(ws, dw, db, bias_from_big, grid, s, N);
}

static bool mfma32_applicable(int Cb, int Cs, int Hs, int Ws, int l0, int l1, int l2) {
  return Cb == 32 && Cs == 32 && Hs == Ws && (Hs == 4 || Hs == 8 || Hs == 16) && l0 == DVAE_NHWC &&
         l1 == DVAE_NHWC && l2 == DVAE_NHWC;
}

int launch_down_mfma32(const ConvArgs& a, hipStream_t s) {
  // the 4x4 end of the conv stack may write NCHW (= the FC stack's flatten order; its mask is then NCHW too)
  const int out_l = (a.Hs == 4 && a.out_layout == DVAE_NCHW) ? DVAE_NHWC : a.out_layout;
  if (!mfma32_applicable(a.Cb, a.Cs, a.Hs, a.Ws, a.big_layout, out_l, DVAE_NHWC)) return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  // HS = 16, 8: k_down32dma (conv_down_dma.hip: weights in registers, tiles by LDS-DMA); HS = 4: K-split 32x32x2 kernel
  if (a.Hs == 16 || a.Hs == 8) return launch_down_mfma32_dma(a, s);
  return launch_down_t<4>(a, s);
}

int launch_up_mfma32(const ConvArgs& a, hipStream_t s) {
  // the 4x4 end of the conv stack may be read NCHW (= the FC stack's (c,h,w) order)
  const int small_l = (a.Hs == 4 && a.small_layout == DVAE_NCHW) ? DVAE_NHWC : a.small_layout;
  if (!mfma32_applicable(a.Cb, a.Cs, a.Hs, a.Ws, small_l, a.out_layout, DVAE_NHWC)) return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  switch (a.Hs) {
    case 16: return launch_up_t<16>(a, s);
    case 8: return launch_up_t<8>(a, s);
    default: return launch_up_t<4>(a, s);
  }
}

int launch_wgrad_mfma32(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N,
                        int Hs, float* ws, hipStream_t s, int small_nchw) {
  static const bool no_ws = env_off("DVAE_WGRAD_WS");     // debug builds: DVAE_WGRAD_WS=0 -> k_wgrad32 for every geometry (A/B)
  if (!no_ws && !small_nchw && (Hs == 16 || Hs == 8))
    return launch_wgrad_mfma32_ws(big, small, dw, db, bias_from_big, N, Hs, ws, s);
  switch (Hs) {
    case 16: return launch_wgrad_t<16>(big, small, dw, db, bias_from_big, N, ws, s, small_nchw);
    case 8: return launch_wgrad_t<8>(big, small, dw, db, bias_from_big, N, ws, s, small_nchw);
    case 4: return launch_wgrad_t<4>(big, small, dw, db, bias_from_big, N, ws, s, small_nchw);
    default: return 1;
  }
}

}  // namespace dvae
