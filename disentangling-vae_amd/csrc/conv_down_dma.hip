// k_down32dma<HS, MASK> (HS = 16, 8): the "down" member of the 32-channel k4/s2/p1 family (Conv2d forward, ConvTranspose2d
// input gradient; reference encoders.py:55-58,73-76, decoders.py:62-64 backward) with
//   * the weights in REGISTERS for the whole kernel (a compute wave owns 16 output channels x the whole contraction
//     K = 16 taps x 32 channels = 128 VGPRs per lane), so no weight image in LDS and two operand reads per 8 MFMAs;
//   * the big-side tiles moved HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): a loader wave executes ~45 instructions per
//     unit instead of the 200-300 of the register-staged loaders (a wave that shares its SIMD with an MFMA-streaming wave
//     issues one instruction per 40-60 cycles, tools/ubench/mfma_mix.hip), with TWO tiles in flight in a ring of three LDS
//     buffers;
//   * the matrix product transposed (A = weights, B = pixels), so that a lane of the D fragment holds 4 consecutive output
//     channels of one pixel: the epilogue is one 16-byte store (and one 16-byte mask load) per 16x16 tile instead of four
//     4-byte ones.
// The accumulator chains are those of k_down32ws (conv_mfma.hip): per output (pixel, channel) four chains j = 0..3, chain j
// sums tap-major over the channels {4 q + j}, then (c0 + c1) + (c2 + c3): the results are bit-identical to that kernel.
//
// LDS tile layout = conv_mfma_common.h (bt[((r * 2 + par) * CW + cw) * 32 + swizzled 16-byte chunk]); the LDS side of an
// LDS-DMA transfer is lane-linear (64 lanes x 16 bytes = 1 KB per wave instruction), so the swizzle and the halo are applied
// on the SOURCE side: lane l of block b fetches the global chunk that belongs at LDS chunk 64 b + l, or 16 bytes of zeros
// (k_zero16) for halo positions.  Which positions are halo is fixed per workgroup for the whole kernel: the persistent
// stride is a multiple of the units per image, so a workgroup always sees the same row range of its images.
#include "conv_mfma_common.h"

namespace dvae {

__device__ __attribute__((aligned(16))) float k_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

template <int HS>
struct DmaGeo {
  using G = Geo<HS>;
  static constexpr int NCHUNK = G::BIG_FLOATS / 4;            // 16-byte chunks of one tile
  static constexpr int NBLK = 44;                             // 1 KB blocks per buffer (4 loader waves x 11)
  static constexpr int NPF = NBLK / 4;
  static constexpr int BUF_FLOATS = NBLK * 256;               // padded: every lane of every block always transfers
  static constexpr int UPI = (HS * HS) / 64;                  // units per image
  static_assert(NCHUNK <= NBLK * 64, "tile fits the padded buffer");
  static_assert(G::IMGS == 1, "one image per unit");
};

__device__ __forceinline__ void barrier_nofence() {
  // the LDS-DMA transfers of the NEXT tile must stay in flight across the barrier: no fence (a __syncthreads() would wait
  // for vmcnt(0)); the orderings that matter are established by explicit s_waitcnt on the loader side and by the data
  // dependences MFMA <- ds_read on the compute side
  asm volatile("s_barrier" ::: "memory");
}

template <int HS, bool MASK>
__global__ __launch_bounds__(512) void k_down32dma(const float* __restrict__ big, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ mask,
                                                   float* __restrict__ out, int act, int n_units, int w_staged) {
  using G = Geo<HS>;
  using D = DmaGeo<HS>;
#ifdef DVAE_DEBUG_SWITCHES
  // timing ablations (DVAE_DMA_ABLATE, debug builds only; results invalid): 1 no output stores, 2 no mask loads, 4 no DMA
  // transfers, 8 no LDS operand reads, 16 no per-unit barriers, 32 no MFMAs
  const int abl = act >> 8;
  act &= 0xff;
#else
  constexpr int abl = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int stride = gridDim.x;
  const int unit0 = blockIdx.x;

  if (wv >= 4) {
    // ---------------------------------------------------------------- loader waves
    const int lw = wv - 4;
    // per slot: the 64-bit source address of this lane's chunk for the workgroup's first unit and its per-unit increment
    const float* src[D::NPF];
    unsigned inc[D::NPF];
    const int n0 = unit0 / D::UPI, sy0 = (unit0 % D::UPI) * G::R;
    const float* base = big + ((long)n0 * G::HB + 2 * sy0 - 1) * G::HB * 32;     // big row 2 sy0 - 1 (never dereferenced if outside)
    const unsigned step = (unsigned)(stride / D::UPI) * (G::HB * G::HB * 32 * 4u);   // bytes per persistent step
#pragma unroll
    for (int k = 0; k < D::NPF; ++k) {
      const int c = ((k * 4 + lw) * 64) + lane;                  // LDS chunk of this lane
      const int q = c >> 3, j = c & 7;                           // pixel slot, chunk within the pixel
      const int cw = q % G::CW, rp = q / G::CW;
      const int par = rp & 1, r = rp >> 1;
      const int by = 2 * sy0 - 1 + r, bx = 2 * cw + par - 1;
      const bool ok = c < D::NCHUNK && by >= 0 && by < G::HB && bx >= 0 && bx < G::HB;
      src[k] = ok ? base + (r * G::HB + bx) * 32 + ((j ^ swz_big<HS>(r, cw)) << 2) : k_zero16;
      inc[k] = ok ? step : 0u;
    }
    auto issue = [&](int buf) {
      float* bt = smem + buf * D::BUF_FLOATS;
#pragma unroll
      for (int k = 0; k < D::NPF; ++k) {
        __builtin_amdgcn_global_load_lds((gas_ptr)src[k], (las_ptr)(bt + (k * 4 + lw) * 256), 16, 0, 0);
        src[k] = (const float*)((const char*)src[k] + inc[k]);
      }
    };
    int unit = unit0;
    if (unit < n_units) issue(0);
    if (unit + stride < n_units) issue(1);
    // tile(unit0) must have landed before the first barrier
    if (unit + stride < n_units) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(D::NPF == 11, "the s_waitcnt immediates below assume 11 transfers per tile");
    barrier_nofence();
    int buf = 0;
    for (; unit < n_units; unit += stride) {
      // compute waves work on `buf`; tile(unit + stride) is in flight into buf + 1; buf + 2 was released by the barrier above
      const int b2 = buf >= 1 ? buf - 1 : 2;        // (buf + 2) % 3
      const bool more = unit + 2 * stride < n_units;
      if (more && !(abl & 4)) issue(b2);
      if (more && !(abl & 4)) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   // all but the 11 newest: tile(unit + stride) has landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(abl & 16)) barrier_nofence();
      buf = buf == 2 ? 0 : buf + 1;
    }
    return;
  }

  // ------------------------------------------------------------------ compute waves
  const int ch = wv & 1, ph = wv >> 1;                 // output-channel half, pixel half of the unit
  const int i16 = lane & 15, kq = lane >> 4;
  // weights: W[tap][h][j] = w(tap, contracted channel 16 h + 4 kq + j, output channel 16 ch + i16)
  f32x4 W[16][2];
  if (w_staged) {                                      // image wl[tap][kc/4][n][kc%4] of dvae_stage_weights
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        W[t][h] = *reinterpret_cast<const f32x4*>(w + (((t * 8 + 4 * h + kq) * 32 + 16 * ch + i16) << 2));
  } else {                                             // raw Conv2d layout w[cs][cb][kh][kw]: n = cs, kc = cb
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) W[t][h][j] = w[(((16 * ch + i16) * 32) + 16 * h + 4 * kq + j) * 16 + t];
  }
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + 16 * ch + 4 * kq);
  // LDS float offsets of this lane's pixel operands: [mt][h][kw >> 1][kh >> 1] (+ a compile-time constant per tap)
  constexpr int NS2 = HS == 8 ? 2 : 1;                 // swz_big<16> does not depend on the row
  int vo[2][2][2][NS2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 32 * ph + 16 * mt + i16;
    const int sy_l = (p / HS) % G::R, sx = p % HS;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int s2 = 0; s2 < NS2; ++s2) {
          const int r = 2 * sy_l + 2 * s2, cw = sx + s;       // r of the kh = 2 s2 tap: swz_big sees (r >> 1) & 1 only
          vo[mt][h][s][s2] = ((2 * sy_l * 2) * G::CW + cw) * 32 + (((4 * h + kq) ^ swz_big<HS>(r, cw)) << 2);
        }
  }
  barrier_nofence();                                   // tile(unit0) is in buffer 0
  __builtin_amdgcn_s_setprio(1);
  int buf = 0;
  f32x4 P[2][2][2];                                    // [slot][mt][h]: pixel operands of a tap
#ifdef DVAE_DEBUG_SWITCHES
  if (abl & 8) {
#pragma unroll
    for (int a = 0; a < 8; ++a) P[a >> 2][(a >> 1) & 1][a & 1] = f32x4{1.f, 2.f, 3.f, 4.f};
  }
#endif
  auto rd = [&](const float* bt, int tap, int slot) {
#ifdef DVAE_DEBUG_SWITCHES
    if (abl & 8) return;
#endif
    const int kh = tap >> 2, kw = tap & 3;
    const int tc = ((kh * 2 + (kw & 1)) * G::CW) * 32;      // row 2 sy_l + kh, parity kw & 1
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        P[slot][mt][h] = *reinterpret_cast<const f32x4*>(bt + vo[mt][h][kw >> 1][NS2 == 2 ? (kh >> 1) : 0] + tc);
  };
  // the first tap's operands of a unit are requested BEFORE the previous unit's epilogue (right behind the barrier that
  // publishes its tile): the LDS round trip runs under the epilogue's arithmetic and stores instead of in front of the unit's
  // first MFMA (round 6: 66.9 -> 66.3 us plain, 74.1 -> 70.0 us masked at 1024 images, the step 1.064 -> 1.056 ms:
  // profiles/r06_v12_ab6.txt)
  if (unit0 < n_units) rd(smem, 0, 0);
  for (int unit = unit0; unit < n_units; unit += stride) {
    const float* bt = smem + buf * D::BUF_FLOATS;
    const long obase = ((long)unit * 64 + 32 * ph + i16) * 32 + 16 * ch + 4 * kq;     // + mt * 16 * 32
    f32x4 mv[2];
    if (MASK) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) mv[mt] = (abl & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(mask + obase + mt * 512);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int cur = t & 1;
      if (t + 1 < 16) rd(bt, t + 1, cur ^ 1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(abl & 32)) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[t][h][j], P[cur][mt][h][j], acc[mt][j], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);    // 4 DS reads (next tap)
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);   // 16 MFMAs (this tap)
    }
    if (!(abl & 16)) barrier_nofence();                // the tile is consumed (every ds_read fed an MFMA above)
    const int nbuf = buf == 2 ? 0 : buf + 1;
    if (unit + stride < n_units) rd(smem + nbuf * D::BUF_FLOATS, 0, 0);     // tile(unit + stride) landed before that barrier
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 a = (acc[mt][0] + acc[mt][1]) + (acc[mt][2] + acc[mt][3]);
      f32x4 o;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float x = epilogue_act(a[v] + bv[v], act);
        if (MASK) x = mv[mt][v] > 0.f ? x : 0.f;
        o[v] = x;
      }
      if (!(abl & 1)) *reinterpret_cast<f32x4*>(out + obase + mt * 512) = o;
    }
    buf = nbuf;
  }
}

template <int HS>
static int launch_down_dma_t(const ConvArgs& a, hipStream_t s) {
  using D = DmaGeo<HS>;
  const int n_units = (int)((long)a.N * HS * HS / 64);
  int grid = n_units < 256 ? n_units : 256;
  grid -= grid % D::UPI;                              // the persistent stride keeps a workgroup on one row range of its images
  const size_t lds = (size_t)3 * D::BUF_FLOATS * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_down32dma<HS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_down32dma<HS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int af = a.act | (env_int("DVAE_DMA_ABLATE", 0) << 8);     // debug builds only
  if (a.mask) hipLaunchKernelGGL((k_down32dma<HS, true>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, af, n_units, a.w_staged);
  else hipLaunchKernelGGL((k_down32dma<HS, false>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.mask, a.out, af, n_units, a.w_staged);
  DVAE_CHECK_LAUNCH();
  return 0;
}

// NHWC in, NHWC out, Hs in {8, 16}; returns 1 if not applicable
int launch_down_mfma32_dma(const ConvArgs& a, hipStream_t s) {
  if (a.big_layout != DVAE_NHWC || a.out_layout != DVAE_NHWC || a.N <= 0) return 1;
  if (a.Hs == 16) return launch_down_dma_t<16>(a, s);
  if (a.Hs == 8) return launch_down_dma_t<8>(a, s);
  return 1;
}

}  // namespace dvae
