// Wave-specialised "down" MFMA kernel, second generation (big -> small: Conv2d forward = encoders.py:73-77,
// ConvTranspose2d dgrad = decoders.py:77-80 under training.py:157), 32 <-> 32 channels, NHWC, small side 16x16 or 8x8.
//
// k_down32ws (conv_mfma.hip) reads SIX 16-byte LDS operands per 16 MFMAs in its compute waves (2 activation, 4 weight
// fragments); the timing ablations of its sibling k_up32ws (profiles/r02_run7_upws_ablation.txt) showed that LDS operand
// reads issued between MFMAs are what holds the matrix core at ~0.6, and that weights kept in registers remove a third
// of them for free.  Here:
//   * compute wave (mh, nh) = 32 pixels (two 16-pixel M-tiles) x 16 output channels, full K = 512.  Its weight slice --
//     16 taps x 32 input channels x 16 output channels -- is 128 VGPRs per lane and stays in registers for the whole kernel;
//     per tap it reads 4 activation fragments (2 M-tiles x 2 channel halves) for 16 MFMAs;
//   * results go to a 64-pixel LDS image; the memory waves (4-7) write it to HBM with 16-byte stores (a unit's output is one
//     contiguous 8 KB block), apply the ReLU mask of the producing layer (16-byte loads, one unit ahead) and keep two input
//     tiles in flight as before;
//   * the 64 KB LDS weight image is only a staging area of the prologue: one input-tile buffer and the output images reuse it.
#include "common.h"
#include "conv_mfma_common.h"

namespace dvae {

typedef float f32x4w __attribute__((ext_vector_type(4)));
#define DWS_OUT_FLOATS 2048            // 64 small pixels x 32 channels

template <int HS, bool MASK>
__global__ __launch_bounds__(512) void k_down32ws2(const float* __restrict__ big, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ mask,
                                                   float* __restrict__ out, int N, int act, int n_units) {
  using G = Geo<HS>;
  static_assert(G::IMGS == 1, "one image per unit");
  static_assert(G::BIG_FLOATS + 2 * DWS_OUT_FLOATS <= 16384, "tile buffer + output images fit the dead weight image");
  constexpr int LNPF = (G::BIG_SLOTS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                                   // 16384 floats, prologue only
  float* btA = smem;                                  // input tiles of the ODD units (reuses the weight image)
  float* ob0 = smem + 16384 - 2 * DWS_OUT_FLOATS;     // two output images at the end of the weight image's space
  float* btB = smem + 16384;                          // input tiles of the EVEN units
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wv < 4;
  const int i16 = lane & 15, kq = lane >> 4;
  const int stride = gridDim.x;
  const int unit0 = blockIdx.x;

  SlotDesc<LNPF> sd;
  f32x4 pfa[LNPF], pfb[LNPF];
  const int ltid = tid - 256;
  if (!is_compute) {
    init_big_slots<HS, 256, LNPF>(sd, ltid);
    if (unit0 < n_units) load_big<HS, LNPF>(pfa, sd, big, unit0, N);
  }
  stage_weights<true>(w, wl, tid);
  if (!is_compute && unit0 < n_units) store_big<HS, LNPF>(pfa, sd, btB);     // unit 0 -> the buffer outside the weight image
  __syncthreads();

  // The two roles separate HERE and never rejoin (every later workgroup barrier is executed once by each role's own code):
  // the 128 weight registers of a compute wave and the two tile register sets of a memory wave are never live together.
  if (is_compute) {
    // ---------------------------------------------------------------- compute wave (mh, nh)
    // its weight slice -> registers: Bq[tap][c2][j] = w[cout = 16 nh + i16][cin = 16 c2 + 4 kq + j][tap]
    const int mh = (wv >> 1) & 1, nh = wv & 1;
    f32x4w Bq[16][2];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
        Bq[t][c2] = *reinterpret_cast<const f32x4w*>(wl + t * 1024 + (kq + 4 * c2) * 128 + (16 * nh + i16) * 4);
    __syncthreads();                                  // the weight image is dead: its space now holds tiles / output images
    const float bv = bias ? bias[16 * nh + i16] : 0.f;
    // per M-tile and (kh >> 1, kw >> 1) variant of swz_big (it depends on (r >> 1, cw >> 1) only): LDS float offset of the
    // FIRST channel half's 16-byte chunk at (row 2 sy, parity 0, column pair sx); the second half is that offset ^ 16 (chunk
    // index ^ 4: the chunk bits sit below the 32-float pixel stride, so the XOR commutes with the linear part)
    int pos[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int p = mh * 32 + mt * 16 + i16;
      const int sy = (p / HS) % G::R, sx = p % HS;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = 2 * sy + 2 * (v >> 1), cw = sx + (v & 1);        // representative (r, cw) of the variant
        pos[mt][v] = ((2 * sy * 2) * G::CW + sx) * 32 + ((kq ^ swz_big<HS>(r, cw)) << 2);
      }
    }
    // D rows 4 kq + v of M-tile mt, column 16 nh + i16 -> float offset in the 64-pixel output image
    const int ooff = (mh * 32 + 4 * kq) * 32 + 16 * nh + i16;          // + (mt * 16 + v) * 32
    __builtin_amdgcn_s_setprio(1);
    int k = 0;
    for (int unit = unit0; unit < n_units; unit += stride, ++k) {
      const float* bt = (k & 1) ? btA : btB;
      f32x4w acc[2][2];                                // [M-tile][chain]: a chain is reused every 4th MFMA (>= the 40-cycle latency)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[mt][c] = f32x4w{0.f, 0.f, 0.f, 0.f};
      // half-taps h2 = 2 tap + channel half: 2 LDS reads + 8 MFMAs each, the reads of the next half-tap issued first
      f32x4w A[2][2];                                  // [slot][mt]
      auto rd = [&](int h2, int slot) {
        const int tap = h2 >> 1, c2 = h2 & 1;
        const int kh = tap >> 2, kw = tap & 3;
        const int v = ((kh >> 1) << 1) | (kw >> 1);    // swz_big's (r >> 1, cw >> 1) with r = 2 sy + kh, cw = sx + (kw >> 1)
        const int c = ((kh * 2 + (kw & 1)) * G::CW + (kw >> 1)) * 32;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          A[slot][mt] = *reinterpret_cast<const f32x4w*>(bt + ((pos[mt][v] ^ (c2 ? 16 : 0)) + c));
      };
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int h2 = 0; h2 < 32; ++h2) {
        const int cur = h2 & 1;
        const int t = h2 >> 1, c2 = h2 & 1;
        if (h2 + 1 < 32) rd(h2 + 1, cur ^ 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc[mt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cur][mt][j], Bq[t][c2][j], acc[mt][j & 1], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // 2 DS reads (next half-tap)
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);    // 8 MFMAs (this half-tap)
      }
      float* ob = ob0 + (k & 1) * DWS_OUT_FLOATS;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f32x4w a = acc[mt][0] + acc[mt][1];
#pragma unroll
        for (int v = 0; v < 4; ++v) ob[ooff + (mt * 16 + v) * 32] = epilogue_act(a[v] + bv, act);
      }
      __syncthreads();
    }
  } else {
    // ---------------------------------------------------------------- memory waves
    if (unit0 + stride < n_units) load_big<HS, LNPF>(pfa, sd, big, unit0 + stride, N);
    if (unit0 + 2 * stride < n_units) load_big<HS, LNPF>(pfb, sd, big, unit0 + 2 * stride, N);
    __syncthreads();                                  // (pairs with the compute waves' barrier after their weight loads)
    f32x4 mk[2];
    auto drain = [&](int u, int b) {
      const float* ob = ob0 + b * DWS_OUT_FLOATS;
      float* dst = out + (long)u * DWS_OUT_FLOATS;     // a unit's 64 pixels x 32 channels are contiguous in NHWC
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = (ltid + 256 * j) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(ob + c);
        if (MASK) {
#pragma unroll
          for (int x = 0; x < 4; ++x) v[x] = mk[j][x] > 0.f ? v[x] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dst + c) = v;
      }
    };
    auto load_mask = [&](int u) {
      const float* src = mask + (long)u * DWS_OUT_FLOATS;
#pragma unroll
      for (int j = 0; j < 2; ++j) mk[j] = *reinterpret_cast<const f32x4*>(src + (ltid + 256 * j) * 4);
    };
    // iteration k: tile of unit k+1 -> its LDS buffer (registers loaded two iterations ago), loads of unit k+3, output image
    // of unit k-1 -> HBM, mask of unit k -> registers.  pfa holds the tiles of units k+1 for even k, pfb for odd k.
    int unit = unit0, k = 0, prev = -1;
    while (unit < n_units) {
      if (unit + stride < n_units) store_big<HS, LNPF>(pfa, sd, btA);               // unit k+1 is odd
      if (unit + 3 * stride < n_units) load_big<HS, LNPF>(pfa, sd, big, unit + 3 * stride, N);
      if (prev >= 0) drain(prev, (k - 1) & 1);
      if (MASK) load_mask(unit);
      __syncthreads();
      prev = unit; unit += stride; ++k;
      if (unit >= n_units) break;
      if (unit + stride < n_units) store_big<HS, LNPF>(pfb, sd, btB);               // unit k+1 is even
      if (unit + 3 * stride < n_units) load_big<HS, LNPF>(pfb, sd, big, unit + 3 * stride, N);
      if (prev >= 0) drain(prev, (k - 1) & 1);
      if (MASK) load_mask(unit);
      __syncthreads();
      prev = unit; unit += stride; ++k;
    }
    if (prev >= 0) drain(prev, (k - 1) & 1);
  }
}

template <int HS>
static int launch_down_ws2_t(const ConvArgs& a, hipStream_t s) {
  using G = Geo<HS>;
  const int n_units = (int)(((long)a.N * HS * HS) / 64);
  const int grid = n_units < 256 ? n_units : 256;
  const size_t lds = (size_t)(16384 + G::BIG_FLOATS) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_down32ws2<HS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_down32ws2<HS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  if (a.mask) hipLaunchKernelGGL((k_down32ws2<HS, true>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
  else hipLaunchKernelGGL((k_down32ws2<HS, false>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
  DVAE_CHECK_LAUNCH();
  return 0;
}

// 32 <-> 32 channels, NHWC everywhere, Hs == Ws in {8, 16}; returns 1 if not applicable
int launch_down_mfma32_ws2(const ConvArgs& a, hipStream_t s) {
  if (!(a.Cb == 32 && a.Cs == 32 && a.Hs == a.Ws && (a.Hs == 8 || a.Hs == 16) && a.big_layout == DVAE_NHWC &&
        a.out_layout == DVAE_NHWC))
    return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  return a.Hs == 16 ? launch_down_ws2_t<16>(a, s) : launch_down_ws2_t<8>(a, s);
}

}  // namespace dvae
