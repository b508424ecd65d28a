// Wave-specialised weight-gradient kernel of the 32 <-> 32 channel k4/s2/p1 convolutions at the two large geometries
// (32x32 <-> 16x16 and 16x16 <-> 8x8; conv2 / conv3 / convT1 / convT2 under training.py:157):
//     dw[cs][cb][kh][kw] = sum_p small[p][cs] * big[pix(p, kh, kw)][cb]          M = 32 cs, N = 32 cb per tap, K = pixels
//
// k_wgrad32 (conv_mfma.hip) keeps both activation tiles pixel-major in LDS (as they come from HBM), so every MFMA operand
// is a ds_read_b32 with a swizzled, non-affine address: 192 LDS reads and ~790 VALU instructions next to the 64 MFMAs of a
// unit and wave, 240 VGPRs (nothing else fits on the CU), matrix cores 0.58 busy.  Here
//   * the LOADER waves (4-7) write both tiles into LDS TRANSPOSED -- channel-major: sT[cs][pixel], bT[cb][row][column
//     parity][column pair];
//   * a COMPUTE wave (0-3, one per SIMD) owns the four taps of one kernel row kh: for 8 consecutive small pixels of a row it
//     reads ONE 16-byte A operand (its channel, 4 pixels per lane half) and FOUR aligned 16-byte quads of the big tile
//     (two per column parity); every tap's operand is a choice of registers among them, not a load.  5 LDS reads per 16
//     MFMAs, all addresses lane-base + immediates;
//   * tiles are double-buffered in LDS, one workgroup barrier per unit, the next-but-one tile is in flight in registers.
// Accumulators persist over the workgroup's units (4 taps x 16 registers per compute wave); the per-workgroup partial
// sums go to the workspace in k_wgrad32's format and k_wgrad32_reduce (conv_mfma.hip) finishes them in a fixed order.
//
// Round 6, the loaders.  A wave beside an MFMA-streaming wave on its SIMD gets one instruction per 40-60 cycles whatever it
// is (profiles/r02_run19_mfma_mix.txt), i.e. 130-170 per 8192-cycle unit; the loaders transposed with scalar stores -- 13
// 16-byte loads and 52 ds_write_b32 per thread and unit plus a predicate per load (4-way bank conflicts on top: 53 % of the
// kernel's LDS-active cycles, profiles/r05_final1_pmc_summary.md) -- and the memory side cost 15 % of the launch.  Now a
// thread owns "quads": ONE channel chunk (4 channels) of FOUR column pairs of a row and parity (pixels two columns apart).
// Its four 16-byte loads are a 4 x 4 block [pixel][channel] that leaves transposed as eight ds_write2_b32 (two neighbouring
// column pairs of ONE channel each, from two arbitrary registers): 12 loads + 24 stores per thread and unit, nothing
// predicated per lane.  What makes that possible:
//   * row layout [20 | 12 floats]: parity 0 = a zero quad, then column pairs 1 .. HS (big columns 1, 3, ..); parity 1 = column
//     pairs 0 .. HS - 1 (big columns 0, 2, ..), then a zero quad: every stored quad lies inside the image's columns, the two
//     pad columns (-1 and 2 HS) are the zero quads, written once at kernel start.  The taps: kw = 0 takes pair sx = the
//     element BEFORE the kw = 2 element (pair sx + 1), kw = 1 / kw = 3 likewise in the other parity;
//   * rows outside the image: HS = 8 -- a unit is a whole image, its rows 0 and 17 are zeroed once and never stored;
//     HS = 16 -- slots are ordered row-major and a row is exactly one wave's worth (64 slots), so "row -1 of the image's first
//     unit" / "row 2 HS of its last" are wave-uniform conditions: a scalar branch puts zeros into that wave's registers;
//   * channel c of a tile starts at float c * CH + 4 (c >> 2) (CH = 400 / 432 big, 80 small): the 16 lanes of a ds_read_b128
//     group hit 16 distinct 16-byte bank slots (the stores stay 4-way conflicted -- all addresses of an instruction are
//     congruent mod 4 dwords -- which costs LDS-array cycles nobody waits for; what the loaders are short of is issue slots) --
//     tools/emu/wgrad_ws_lds.py checks the reads and every operand value against the direct definition.  The same quads as
//     four provably aligned ds_write_b128 (conflict-free: the 8 lanes of a store group are the 8 chunks of one quad; 16 v_mov
//     per quad to put the transposed dwords into consecutive registers) measured 77.9 us against 76.5 at 1024 images, 27.6
//     against 27.3 at 16 <-> 8, the same step time (profiles/r06_v11_ab5.txt): bank conflicts of the stores are not what the
//     kernel waits for, and the form with fewer instructions wins.
// Measured (same box, profiles/r06_v7_ab2.txt): 80.6 -> 76.4 us (+ reduce) at 1024 images, 16 <-> 8: 30.0 -> 27.2 us; the
// 1024-image step 1.1035 -> 1.066 ms (inside the step the loaders compete with the other stream's kernels as well).
// The MFMA stream (operands, order) is unchanged: results are bit-identical to rounds 2-5.
#include "common.h"
#include "conv_mfma_common.h"
#include "wgrad_reduce.h"

namespace dvae {

#define WGW_MAX_BLOCKS WG_MAX_BLOCKS
#define WGW_STRIDE WG_STRIDE               // the reduce kernel (and its partial format) is shared with k_wgrad32

template <int HS>
struct WGeo {
  using G = Geo<HS>;
  static constexpr int NQ = HS / 4;                                      // stored quads per row and parity
  static constexpr int CWP = HS + 4;                                     // floats per row and parity: 20 / 12
  static constexpr int BCH = G::BROWS * 2 * CWP;                         // floats per channel of the big tile: 400 / 432
  static constexpr int SCH = 64 + 16;                                    // floats per channel of the small tile
  static constexpr int BT_FLOATS = 31 * BCH + 28 + BCH;                  // channel c at c * CH + 4 (c >> 2)
  static constexpr int ST_FLOATS = 31 * SCH + 28 + 64;
  static constexpr int STORED_ROWS = HS == 16 ? G::BROWS : G::BROWS - 2; // HS = 8: rows 0 and 17 are never inside the image
  static constexpr int NBIG = STORED_ROWS * 2 * NQ * 8;                  // quads of the big tile: 640 / 512
  static constexpr int BUF_FLOATS = BT_FLOATS + ST_FLOATS;
  static_assert(BT_FLOATS % 4 == 0 && ST_FLOATS % 4 == 0 && NBIG % 128 == 0 && NBIG + 128 <= 768, "slot plan");
};

template <int HS>
__global__ __launch_bounds__(512) void k_wgrad32ws(const float* __restrict__ big, const float* __restrict__ small,
                                                   float* __restrict__ ws, int n_units
#ifdef DVAE_DEBUG_SWITCHES
                                                   , int abl   // timing ablations (DVAE_WGWS_ABLATE; results invalid): 1 loaders do not
                                                               // write LDS, 2 no tile loads, 8 no MFMAs, 16 no LDS operand reads, 32 no barrier
#endif
                                                   ) {
#ifndef DVAE_DEBUG_SWITCHES
  constexpr int abl = 0;
#endif
  using G = Geo<HS>;
  using W = WGeo<HS>;
  static_assert(G::IMGS == 1, "one image per unit");
  constexpr int HB = 2 * HS;
  extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 x (bT | sT)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wv < 4;
  const int stride = gridDim.x;
  const int unit0 = blockIdx.x;

  // zero quads / rows outside the image: written here, never again
  for (int e = tid; e < 2 * W::BUF_FLOATS / 4; e += 512) reinterpret_cast<f32x4*>(smem)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  if (is_compute) {
    // ------------------------------------------------------------------ compute wave: kernel row kh, taps kw = 0..3
    const int kh = wv;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float sumS = 0.f, sumK1 = 0.f, sumK2 = 0.f;
    const int abase = i * W::SCH + 4 * (i >> 2) + 4 * h;                              // + sy * HS + 8 gx
    const int bbase = i * W::BCH + 4 * (i >> 2) + kh * 2 * W::CWP + 4 * h;            // + (2 sy * 2 + par) * CWP + 8 gx  (+ 4: second quad)
    __builtin_amdgcn_s_setprio(1);
    __syncthreads();                                                  // tile of the first unit is in buffer 0
    int buf = 0;
    for (int unit = unit0; unit < n_units; unit += stride) {
      const float* bt = smem + buf * W::BUF_FLOATS;
      const float* st = bt + W::BT_FLOATS;
      constexpr int NG = 64 / 8;                                      // groups of 8 pixels: (sy, gx)
      constexpr int GPR = HS / 8;                                     // groups per small row
      f32x4 A[2], P0a[2], P0b[2], P1a[2], P1b[2];
      if (abl & 16) {
#pragma unroll
        for (int c = 0; c < 2; ++c) { A[c] = f32x4{1.f, 2.f, 3.f, 4.f}; P0a[c] = A[c]; P0b[c] = A[c]; P1a[c] = A[c]; P1b[c] = A[c]; }
      }
      auto rd = [&](int g, int slot) {
        if (abl & 16) return;
        const int sy = g / GPR, gx = g % GPR;
        A[slot] = *reinterpret_cast<const f32x4*>(st + abase + sy * HS + 8 * gx);
        const float* bp = bt + bbase + (4 * sy) * W::CWP + 8 * gx;    // row 2 sy + kh, parity 0: [zero | pairs 1..4 | ..]
        P0a[slot] = *reinterpret_cast<const f32x4*>(bp);
        P0b[slot] = *reinterpret_cast<const f32x4*>(bp + 4);
        P1a[slot] = *reinterpret_cast<const f32x4*>(bp + W::CWP);     // parity 1: [pairs 0..3 | .. | zero]
        P1b[slot] = *reinterpret_cast<const f32x4*>(bp + W::CWP + 4);
      };
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      if (!(abl & 8)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int c = g & 1;
        if (g + 1 < NG) rd(g + 1, c ^ 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // small pixel sx = 8 gx + 4 h + j; big column pair cw = sx + (kw >> 1), parity kw & 1; parity 0 is stored from pair
          // 1 on behind a zero quad: pair cw sits at float cw + 3 of its row
          const float a = A[c][j];
          const float b0 = j > 0 ? P0b[c][j - 1] : P0a[c][3];          // kw = 0: pair sx
          const float b1 = P1a[c][j];                                  // kw = 1
          const float b2 = P0b[c][j];                                  // kw = 2: pair sx + 1
          const float b3 = j < 3 ? P1a[c][j + 1] : P1b[c][0];          // kw = 3
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
          sumS += a; sumK1 += b1; sumK2 += b2;
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);            // 5 DS reads (next group)
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);           // 16 MFMAs (this group)
      }
      }
      if (!(abl & 32)) __syncthreads();
      buf ^= 1;
    }
    // partial results of this workgroup, k_wgrad32's layout: ws[block][tap][cs][cb] + 160 bias floats
    float* wsw = ws + (long)blockIdx.x * WGW_STRIDE;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tap = kh * 4 + t;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int cs = (e & 3) + 8 * (e >> 2) + 4 * h;
        wsw[(tap * 32 + cs) * 32 + i] = acc[t][e];
      }
    }
    float* wsb = wsw + 16384;
    sumS += __shfl_xor(sumS, 32, 64);
    sumK1 += __shfl_xor(sumK1, 32, 64);
    sumK2 += __shfl_xor(sumK2, 32, 64);
    if (h == 0) {
      if (kh == 0) wsb[i] = sumS;                                     // sum of the small side per cs
      // taps (1,1) = 5, (1,2) = 6, (2,1) = 9, (2,2) = 10 together cover every big pixel exactly once (bias from the big side)
      if (kh == 1) { wsb[32 + i] = sumK1; wsb[64 + i] = sumK2; }
      if (kh == 2) { wsb[96 + i] = sumK1; wsb[128 + i] = sumK2; }
    }
  } else {
    // ------------------------------------------------------------------ loader waves
    const int lt = tid - 256;
    const int lw = wv - 4;
    // quad slot s = lt + 256 k, k = 0..2: s < NBIG: big tile, s = ((row, parity, quad), chunk) row-major -- a row of the
    // HS = 16 tile is one wave's worth; NBIG <= s < NBIG + 128: small tile (pixel quad, chunk).  k = 0, 1 are big quads for every
    // thread; k = 2: HS = 16 -- loader waves 0, 1 big (rows 8, 9), waves 2, 3 small; HS = 8 -- waves 0, 1 small, 2, 3 nothing
    const bool k2_big = HS == 16 && lw < 2, k2_small = HS == 16 ? lw >= 2 : lw < 2;
    // float offsets: source relative to the unit's base pointer (the big base is row -1 of the unit's rows, so every offset is
    // non-negative: scalar base + 32-bit lane offset addressing); destination of (buffer, channel u) inside the LDS allocation
    unsigned g_of[3];
    int l_of[2][3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int s = lt + 256 * k;
      int l0 = 0, ch = 0;
      g_of[k] = 0;
      if (s < W::NBIG) {
        const int chunk = s & 7, q = s >> 3;
        const int q4 = q % W::NQ, par = (q / W::NQ) & 1, r = q / (2 * W::NQ) + (HS == 16 ? 0 : 1);
        g_of[k] = (unsigned)((r * HB + 8 * q4 + (1 - par)) * 32 + 4 * chunk);       // + 64 t: big columns 8 q4 + 2 t + (1 - par)
        l0 = chunk * (4 * W::BCH + 4) + (r * 2 + par) * W::CWP + (par == 0 ? 4 : 0) + 4 * q4;
        ch = W::BCH;
      } else {                                         // (HS = 8: s >= NBIG + 128 -- loaded like the small quad 128 slots back, never stored)
        const int sp = (s - W::NBIG) & 127, chunk = sp & 7, pq = sp >> 3;
        g_of[k] = (unsigned)(pq * 128 + 4 * chunk);                                 // + 32 t: pixels 4 pq + t
        l0 = W::BT_FLOATS + chunk * (4 * W::SCH + 4) + 4 * pq;
        ch = W::SCH;
      }
#pragma unroll
      for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
        for (int u = 0; u < 4; ++u) l_of[b_][k][u] = b_ * W::BUF_FLOATS + l0 + u * ch;
    }
    // two register sets: tiles k+2 and k+3 are in flight from HBM while tile k+1 sits in LDS (the whole chip issues its tile
    // loads in the same few hundred cycles after a barrier: one unit time of look-ahead does not cover the queueing delay).
    // Every load_unit issues the SAME 12 loads whatever the unit (a unit index past the end is clamped, a wave whose row lies
    // outside the image loads the neighbouring row and stores zeros instead): with loads behind branches the compiler cannot
    // count what is in flight and makes each store wait for ALL of them (s_waitcnt vmcnt(0) in rounds 2-5), the newest set
    // included -- which is the look-ahead gone.
    f32x4 pA[3][4], pB[3][4];
    constexpr unsigned UPI = HS * HS / G::U;                           // units per image: 4 / 1
    const unsigned last_unit = (unsigned)(n_units - 1);
    auto load_unit = [&](int u_, f32x4 (&p)[3][4]) {
      if (abl & 2) return;
      const unsigned u = (unsigned)u_ < last_unit ? (unsigned)u_ : last_unit;
      const unsigned n0 = u / UPI, sy0 = (u % UPI) * G::R;
      const bool top = HS == 16 && sy0 == 0, bot = HS == 16 && sy0 + G::R == HS;   // (scalar: the unit's first / last rows)
      // base = row 2 sy0 - 1 of the image; the wave that holds row -1 (2 HS) reads row 0 (2 HS - 1) instead
      const float* bg = big + ((size_t)n0 * HB + 2 * sy0) * (HB * 32) - HB * 32;
      const float* s0 = bg + ((top && lw == 0) ? HB * 32 : 0) + g_of[0];
      const float* s1 = bg + g_of[1];
      const float* s2 = (k2_big ? bg - ((bot && lw == 1) ? HB * 32 : 0) : small + (size_t)u * (G::U * 32)) + g_of[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) p[0][t] = *reinterpret_cast<const f32x4*>(s0 + 64 * t);
#pragma unroll
      for (int t = 0; t < 4; ++t) p[1][t] = *reinterpret_cast<const f32x4*>(s1 + 64 * t);
      if (k2_big) {
#pragma unroll
        for (int t = 0; t < 4; ++t) p[2][t] = *reinterpret_cast<const f32x4*>(s2 + 64 * t);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) p[2][t] = *reinterpret_cast<const f32x4*>(s2 + 32 * t);
      }
    };
    auto store_unit = [&](int b, int u_, const f32x4 (&p)[3][4]) {
      if (abl & 1) return;
      const unsigned sy0 = ((unsigned)u_ % UPI) * G::R;
      const bool top = HS == 16 && sy0 == 0, bot = HS == 16 && sy0 + G::R == HS;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k == 2 && !k2_big && !k2_small) continue;
        // the 4 x 4 block [pixel t][channel u] leaves transposed: ds_write2_b32 takes its two dwords from two ARBITRARY
        // registers, so two neighbouring pixels of one channel go out in one instruction with no register shuffling (a
        // ds_write_b128 per channel would need its four dwords in consecutive registers: 16 v_mov per quad)
        const bool outside = (k == 0 && lw == 0 && top) || (k == 2 && k2_big && lw == 1 && bot);   // scalar
        if (outside) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) smem[l_of[b][k][u] + t] = 0.f;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) smem[l_of[b][k][u] + t] = p[k][t][u];
        }
      }
    };
    if (unit0 < n_units) { load_unit(unit0, pA); store_unit(0, unit0, pA); }
    load_unit(unit0 + stride, pA);
    load_unit(unit0 + 2 * stride, pB);
    __syncthreads();
    // set A holds tiles k+1 (k even), set B tiles k+1 (k odd); the loop is unrolled by two so that the sets stay static
    int unit = unit0;
    while (unit < n_units) {
      if (unit + stride < n_units) store_unit(1, unit + stride, pA);
      load_unit(unit + 3 * stride, pA);
      if (!(abl & 32)) __syncthreads();
      unit += stride;
      if (unit >= n_units) break;
      if (unit + stride < n_units) store_unit(0, unit + stride, pB);
      load_unit(unit + 3 * stride, pB);
      if (!(abl & 32)) __syncthreads();
      unit += stride;
    }
  }
}

template <int HS>
static int launch_wgrad_ws_t(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N,
                             float* ws, hipStream_t s) {
  using W = WGeo<HS>;
  const int n_units = (int)(((long)N * HS * HS) / 64);
  // Persistent grid.  One workgroup per CU fills the chip -- which is what a step of 129 .. 320 images must NOT do: there the
  // other stream's kernels are the chain of small dependent launches the iteration waits for (8x8 / 4x4 layers, the FC chain),
  // and with every CU taken by a 512-thread, 130 KB workgroup that runs 3-5 units they queue for ~20 us each.  192 workgroups
  // leave a quarter of the chip to them: 192 / 256 images 0.384 -> 0.368, 0.432 -> 0.416 ms, btcvae 64x64x1 B = 256 0.400 ->
  // 0.382 (profiles/r06_s2_cap3.txt: the shipped library against a WGW_FULL_GRID build, same box, three alternations; the
  // sweep over grid sizes: r06_s2_cap1.txt, r06_s2_cap2.txt).  At 128 images and below a workgroup runs one or two units and is
  // gone before anything queues (256: 0.331, 192: 0.335 ms); at 384 images the two are level, at 512 the full grid wins
  // (0.634 vs 0.645 ms) and from 1024 images the main stream's kernels fill the chip themselves: a smaller grid only delays
  // the weight gradients (round 5 measured the same at 1024: profiles/r05_v23_side_cap_ab.txt).
#ifdef WGW_FULL_GRID                 // (variant builds, tools/build_variant.sh: the A/B partner)
  const int full = WGW_MAX_BLOCKS;
#else
  const int full = (N > 128 && N <= 320) ? 192 : WGW_MAX_BLOCKS;
#endif
  int grid = n_units < full ? n_units : full;
  {
    static const int cap = env_int("DVAE_WGRAD_GRID", WGW_MAX_BLOCKS);   // debug builds: A/B of the persistent grid size
    if (cap > 0 && cap < grid) grid = cap;
  }
  const size_t lds = (size_t)2 * W::BUF_FLOATS * sizeof(float);
  static DeviceOnce attr;
  if (attr.first()) { (void)hipFuncSetAttribute((const void*)k_wgrad32ws<HS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
#ifdef DVAE_DEBUG_SWITCHES
  static const int abl = env_int("DVAE_WGWS_ABLATE", 0);
  hipLaunchKernelGGL(k_wgrad32ws<HS>, dim3(grid), dim3(512), lds, s, big, small, ws, n_units, abl);
#else
  hipLaunchKernelGGL(k_wgrad32ws<HS>, dim3(grid), dim3(512), lds, s, big, small, ws, n_units);
#endif
  DVAE_CHECK_LAUNCH();
  return launch_wgrad32_reduce(ws, dw, db, bias_from_big, grid, s, N);
}

// NHWC on both sides, Hs in {8, 16}; returns 1 if not applicable
int launch_wgrad_mfma32_ws(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N, int Hs,
                           float* ws, hipStream_t s) {
  if (Hs == 16) return launch_wgrad_ws_t<16>(big, small, dw, db, bias_from_big, N, ws, s);
  if (Hs == 8) return launch_wgrad_ws_t<8>(big, small, dw, db, bias_from_big, N, ws, s);
  return 1;
}

}  // namespace dvae
