// Round-2 draft of the "up" MFMA kernel (small -> big: ConvT forward, Conv dgrad), selected with DVAE_UP_R2=1.
// Identical to k_up32 (conv_mfma.hip) except for the address arithmetic of the mask loads and result stores, which
// is hoisted out of the per-unit loop: tools/isa_report.py counts ~690 VALU instructions next to the 64 MFMAs of a
// unit in k_up32<16,true>, most of them the sixteen 64-bit offset chains that up_offsets() rebuilds twice per unit.
// NOT YET RUN ON HARDWARE (written after the round-1 GPU budget was spent): never selected by default; the parity
// tests of tests/test_gpu_kernels.py cover it once DVAE_UP_R2=1 is set for them.
#include <stdlib.h>
#include "common.h"
#include "conv_mfma_common.h"

namespace dvae {

// Software pipeline per workgroup:  [LDS tile(u) <- regs] | barrier | issue loads tile(u+1), mask(u) |
// store results(u-1) | MFMA(u) | results(u) -> regs.  The global stores of unit u-1 and the loads
// of unit u+1 are a whole MFMA phase old when the next iteration waits on vmcnt.
template <int HS, bool MASK>
__global__ __launch_bounds__(512) void k_up32r2(const float* __restrict__ small, const float* __restrict__ w,
                                              const float* __restrict__ bias, const float* __restrict__ mask,
                                              float* __restrict__ out, int N, int act, int n_units,
                                              int small_nchw) {
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
  stage_weights<false>(w, wl, tid);
  const float bv = bias ? bias[i] : 0.f;
  float vals[16];
  int prev_unit = -1;
  // Output / mask address of D-fragment row e of this wave's (class, M-tile):
  //   (((n0 + im) * HB + 2 (sy0 + mm) + py) * HB + 2 ll + px) * 32 + i
  // Only n0 and sy0 depend on the unit, and they are wave-uniform: the lane part lo[e] is computed ONCE, the
  // unit part is a scalar base per iteration (k_up32 recomputes sixteen 64-bit multiply-add chains twice per unit).
  int lo[16];
  unsigned im_bits = 0;                     // 2 bits per row: image-in-unit (IMGS <= 4)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int pp = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    const int im = pp / (G::R * HS), mm = (pp / HS) % G::R, ll = pp % HS;
    lo[e] = ((im * G::HB + 2 * mm + py) * G::HB + 2 * ll + px) * 32 + i;
    im_bits |= (unsigned)im << (2 * e);
  }
  auto unit_base = [&](int u) -> long {     // wave-uniform
    const long P0 = (long)u * G::U;
    const long n0 = P0 / (HS * HS);
    const int sy0 = (int)(P0 % (HS * HS)) / HS;
    return (n0 * G::HB + 2 * sy0) * G::HB * 32;
  };
  auto store_unit = [&](int u) {
    float* ob = out + unit_base(u);
    const int n0 = (int)(((long)u * G::U) / (HS * HS));
    if (n0 + G::IMGS <= N) {                // wave-uniform: the whole unit lies inside the tensor
#pragma unroll
      for (int e = 0; e < 16; ++e) ob[lo[e]] = vals[e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (n0 + (int)((im_bits >> (2 * e)) & 3) < N) ob[lo[e]] = vals[e];
    }
  };

  for (; unit < n_units; unit += gridDim.x) {
    __syncthreads();  // previous unit's reads of st are complete
    store_small_halo<HS>(pf, sd, st);
    __syncthreads();
    if (unit + (int)gridDim.x < n_units) load_small_halo<HS>(pf, sd, small, unit + gridDim.x, N, small_nchw);
    float mv[16];
    if (MASK) {       // prefetch the ReLU mask of this unit (consumed after the MFMA phase)
      const float* mb = mask + unit_base(unit);
      const int n0 = (int)(((long)unit * G::U) / (HS * HS));
#pragma unroll
      for (int e = 0; e < 16; ++e)
        mv[e] = (G::IMGS == 1 || n0 + (int)((im_bits >> (2 * e)) & 3) < N) ? mb[lo[e]] : 0.f;
    }
    if (prev_unit >= 0) store_unit(prev_unit);

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
  if (prev_unit >= 0) store_unit(prev_unit);
}


static int units_for_r2(int N, int HS) { return (int)(((long)N * HS * HS + 63) / 64); }

template <int HS>
static int launch_up_r2_t(const ConvArgs& a, hipStream_t s) {
  using G = Geo<HS>;
  const int n_units = units_for_r2(a.N, HS);
  const int grid = n_units < 256 ? n_units : 256;
  const size_t lds = (16384 + G::SH_FLOATS) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_up32r2<HS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_up32r2<HS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int small_nchw = a.small_layout == DVAE_NCHW;
  if (a.mask) hipLaunchKernelGGL((k_up32r2<HS, true>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, small_nchw);
  else hipLaunchKernelGGL((k_up32r2<HS, false>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, small_nchw);
  DVAE_CHECK_LAUNCH();
  return 0;
}

// same applicability as launch_up_mfma32; returns 1 if not applicable
int launch_up_mfma32_r2(const ConvArgs& a, hipStream_t s) {
  const int small_l = (a.Hs == 4 && a.small_layout == DVAE_NCHW) ? DVAE_NHWC : a.small_layout;
  if (!(a.Cb == 32 && a.Cs == 32 && a.Hs == a.Ws && (a.Hs == 4 || a.Hs == 8 || a.Hs == 16) && small_l == DVAE_NHWC &&
        a.out_layout == DVAE_NHWC))
    return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  switch (a.Hs) {
    case 16: return launch_up_r2_t<16>(a, s);
    case 8: return launch_up_r2_t<8>(a, s);
    default: return launch_up_r2_t<4>(a, s);
  }
}

}  // namespace dvae
