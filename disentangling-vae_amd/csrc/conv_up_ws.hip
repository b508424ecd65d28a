// Wave-specialised "up" MFMA kernel (small -> big: ConvTranspose2d forward = decoders.py:77-80, Conv2d dgrad =
// encoders.py:73-77 under training.py:157) for the two large geometries of the Burgess stack, 16x16 -> 32x32 and
// 8x8 -> 16x16, 32 <-> 32 channels, NHWC on both sides.
//
// Why a second kernel next to k_up32 (conv_mfma.hip): in k_up32 every wave does everything -- stage the input tile,
// prefetch the ReLU mask with sixteen strided 4-byte loads, store its D fragment with sixteen strided 4-byte stores,
// and the MFMAs -- in phases that the two workgroup barriers per unit keep aligned across the waves of a SIMD, so the
// matrix core idles whenever its two waves do address arithmetic or wait for memory (MFMA-busy 0.49-0.58,
// profiles/r01_run31_pmc_summary.md; hoisting the address arithmetic alone changed nothing, profiles/r02_run2_ab.txt).
// Here the roles are split:
//   waves 0-3 (one per SIMD) = compute: LDS operand reads, 128 v_mfma_f32_32x32x2_f32 per unit (one output-parity class
//       x both 32-pixel M-tiles: the B fragments of a tap are shared by the two tiles), bias / ReLU, D fragments
//       written to an LDS image of the unit's output block;
//   waves 4-7 = memory: input tiles two units ahead (registers) and one unit ahead (LDS), the previous unit's output
//       block from LDS to HBM with 16-byte coalesced stores -- a unit's output is ONE contiguous 32 KB block of the NHWC
//       tensor -- masked by the producing layer's activation fetched with 16-byte loads one unit ahead.
// One workgroup barrier per unit; input and output images are double-buffered in LDS.  The weights of a compute wave's
// class live in 64 VGPRs per lane for the whole kernel (they are the same for every unit), so the 64 KB LDS weight image
// is only a staging area of the prologue and the two 32 KB output images reuse it: 64 KB + 2 x 13.5 KB = 91 KB.  The
// bias / ReLU / LDS-write epilogue of unit u is issued inside the MFMA stream of unit u+1 (two accumulator sets).
// Per unit a CU moves 13.5 KB in + 32 KB out (+ 32 KB mask) for 8192 matrix-core cycles: at 100 % MFMA rate that is
// 3.5 (5.9) TB/s over the chip -- the masked variant is HBM- and MFMA-bound at the same time.
#include "common.h"
#include "conv_mfma_common.h"

namespace dvae {

// Input tile of the memory waves (round 6).  Only the pixels INSIDE the image's columns are staged: HS = 16 -- 6 rows x 16
// columns x 8 chunks of 16 bytes = 768 slots, 3 per thread; HS = 8 (a unit is a whole image) -- rows 1..8 x 8 x 8 = 512, 2 per
// thread.  The halo columns (and for HS = 8 the halo rows) are zeroed once at kernel start and never written again.  A slot's
// row is wave-uniform (slot s = thread + 256 k is row-major with 128 / 64 slots per row), so "this row lies outside the image"
// (HS = 16: row 0 of an image's first unit, row 5 of its last) is a scalar condition: such a wave loads the neighbouring row
// (same instruction count, valid address) and stores zeros.  Every unit therefore issues the SAME loads -- no per-lane
// predicate, exec juggling or branch per slot (rounds 2-5: 70 instructions per thread and unit for 4 loads; a wave beside an
// MFMA-streaming wave gets 130-170 instructions per unit in total, profiles/r02_run19_mfma_mix.txt) -- and the compiler can
// count what is in flight instead of waiting for everything.
template <int HS>
struct UpIn {
  using G = Geo<HS>;
  static constexpr int NPF = HS == 16 ? 3 : 2;
  static constexpr int SPR = HS * 8;                  // slots per row: 128 / 64
  unsigned gofs[NPF];                                 // float offset relative to row sy0 - 1 of the unit's image
  int lds[NPF];                                       // (swizzled) float offset inside an input image
  __device__ __forceinline__ void init(int ht) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int s = ht + 256 * k;
      const int chunk = s & 7, col = ((s >> 3) & (HS - 1)) + 1, row = s / SPR + (HS == 16 ? 0 : 1);
      gofs[k] = (unsigned)((row * HS + col - 1) * 32 + 4 * chunk);
      lds[k] = (row * G::SCOLS + col) * 32 + ((chunk ^ swz_small<HS>(row, col)) << 2);
    }
  }
};

#define UPWS_OUT_FLOATS 8192        // one unit's output block: 64 small pixels x 4 parity classes x 32 channels

// MASK: 0 none; 1 the producing layer's fp32 activation (32 KB per unit); 2 its bit plane (`bits`: one uint32 per big-side
// pixel, bit c = [channel c > 0]: 1 KB per unit).  OUTBITS: the forward pass also EMITS the bit plane of its (post-ReLU)
// output into `bits` -- the compute waves take it from the accumulators with one ballot per D-fragment row (lanes 0-31 /
// 32-63 of a row are the 32 channels of two pixels) into a 1 KB LDS image per output image, the memory waves drain it.
template <int HS, int MASK, bool OUTBITS = false>
__global__ __launch_bounds__(512) void k_up32ws(const float* __restrict__ small, const float* __restrict__ w,
                                                const float* __restrict__ bias, const float* __restrict__ mask,
                                                float* __restrict__ out, int act, int n_units, int w_staged,
                                                uint32_t* __restrict__ bits) {
  static_assert(!(MASK && OUTBITS), "a forward pass emits bits, an input-gradient pass consumes a mask");
  static_assert(!OUTBITS || HS == 16, "the bit image's lane-half offset (8 big columns) assumes 16-pixel small rows");
  using G = Geo<HS>;
  static_assert(G::IMGS == 1, "one image per unit");
  constexpr int HB = 2 * HS;
  constexpr int LNPF = UpIn<HS>::NPF;
#ifdef DVAE_DEBUG_SWITCHES
  const int abl = act >> 8;     // timing ablations (DVAE_UPWS_ABLATE, debug builds; results invalid): 1 no output stores,
  act &= 0xff;                  // 2 no mask loads
#else
  constexpr int abl = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                                  // 16384 floats: w[tap][cs/4][cb][cs%4] -- prologue only
  float* out0 = smem;                                // 2 x UPWS_OUT_FLOATS: the output images REUSE the weight image's space
  float* in0 = smem + 16384;                         // 2 x G::SH_FLOATS
  uint32_t* bimg = reinterpret_cast<uint32_t*>(smem + 16384 + 2 * G::SH_FLOATS);   // OUTBITS: 2 x 256 words
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = wv < 4;
  const int stride = gridDim.x;
  const int unit0 = blockIdx.x;

  UpIn<HS> sd;
  f32x4 pf[LNPF];
  const int ht = tid - 256;
  const int mw_ = wv - 4;                            // memory wave 0..3
  constexpr unsigned UPI = HS * HS / G::U;           // units per image: 4 / 1
  const unsigned last_unit = (unsigned)(n_units - 1);
  // rows of this wave's slots that can fall outside the image (HS = 16): slot 0 of waves 0, 1 = tile row 0, slot 2 of waves
  // 2, 3 = tile row 5
  auto load_in = [&](int u_) {
    const unsigned u = (unsigned)u_ < last_unit ? (unsigned)u_ : last_unit;      // (a unit index past the end: clamped, never stored)
    const unsigned n0 = u / UPI, sy0 = (u % UPI) * G::R;
    const float* base = small + ((size_t)n0 * HS + sy0) * (HS * 32) - HS * 32;    // row sy0 - 1 (not dereferenced outside the image)
    const bool top = HS == 16 && sy0 == 0, bot = HS == 16 && sy0 + G::R == HS;
#pragma unroll
    for (int k = 0; k < LNPF; ++k) {
      const float* src = base + sd.gofs[k];
      if (HS == 16 && k == 0) src += (top && mw_ < 2) ? HS * 32 : 0;
      if (HS == 16 && k == 2) src -= (bot && mw_ >= 2) ? HS * 32 : 0;
      pf[k] = *reinterpret_cast<const f32x4*>(src);
    }
  };
  auto store_in = [&](int u_, float* st) {
    const unsigned sy0 = ((unsigned)u_ % UPI) * G::R;
    const bool top = HS == 16 && sy0 == 0, bot = HS == 16 && sy0 + G::R == HS;
#pragma unroll
    for (int k = 0; k < LNPF; ++k) {
      const bool outside = HS == 16 && ((k == 0 && top && mw_ < 2) || (k == 2 && bot && mw_ >= 2));   // scalar
      *reinterpret_cast<f32x4*>(st + sd.lds[k]) = outside ? f32x4{0.f, 0.f, 0.f, 0.f} : pf[k];
    }
  };
  // the halo of both input images: zero, once
  for (int e = tid; e < 2 * G::SH_FLOATS / 4; e += 512) reinterpret_cast<f32x4*>(in0)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (!is_compute) {
    sd.init(ht);
    load_in(unit0);
  }
  // compute-wave constants (the memory waves skip their use)
  const int cls = wv & 3;
  const int py = cls >> 1, px = cls & 1;
  const int i = lane & 31, h = lane >> 5;
  // B operands of this wave's parity class -- 4 taps x 32 contracted channels x 32 output channels = 64 VGPRs per lane --
  // stay in registers for the whole kernel: the weights are the same for every unit, and every LDS read taken out of the
  // MFMA loop shortens it (timing ablations, profiles/r02_run7_upws_ablation.txt: operand reads cost 15 % of the loop)
  f32x4 Bq[4][4];
  const int boff = i * 4 + h * 128;                  // + ((kh*4+kw)*8 + 2q) * 128
  // pre-staged weights (dvae_stage_weights): `w` already is the image, the compute waves take their B fragments straight
  // from it (16 coalesced 16-byte loads per lane, L2-resident, issued before the first barrier) and the LDS staging pass
  // disappears from the prologue
  if (w_staged) {
    if (is_compute) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kh = 1 - py + 2 * (t >> 1), kw = 1 - px + 2 * (t & 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) Bq[t][q] = *reinterpret_cast<const f32x4*>(w + boff + ((kh * 4 + kw) * 8 + 2 * q) * 128);
      }
    }
  } else {
    stage_weights<false>(w, wl, tid);
  }
  if (!is_compute) {
    if (unit0 < n_units) store_in(unit0, in0);
    load_in(unit0 + stride);
  }
  __syncthreads();                                   // weight image and the first input tile are in LDS
  if (is_compute && !w_staged) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kh = 1 - py + 2 * (t >> 1), kw = 1 - px + 2 * (t & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) Bq[t][q] = *reinterpret_cast<const f32x4*>(wl + boff + ((kh * 4 + kw) * 8 + 2 * q) * 128);
    }
  }
  __syncthreads();                                   // the weight image is dead from here on: the output images reuse its 64 KB

  if (is_compute) {
    // ---------------------------------------------------------------- compute waves: class cls, both M-tiles
    const float bv = bias ? bias[i] : 0.f;
    // LDS float offsets of the A operand (input tile) for (M-tile, tap, 8-channel group)
    int aoff[2][4][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int p = mt * 32 + i;
      const int m = p / HS, l = p % HS;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ty = t >> 1, tx = t & 1;
        const int row = m + (py - ty) + 1, col = l + (px - tx) + 1;
        const int sw = swz_small<HS>(row, col);
#pragma unroll
        for (int q = 0; q < 4; ++q) aoff[mt][t][q] = (row * G::SCOLS + col) * 32 + (((2 * q + h) ^ sw) << 2);
      }
    }
    // D-fragment row e of M-tile mt -> float offset in the output image: ((2m+py) * HB + 2l+px) * 32 + i
    int ooff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int pp = mt * 32 + 4 * h;                // + (e & 3) + 8 * (e >> 2)
      const int m = pp / HS, l = pp % HS;
      ooff[mt] = ((2 * m + py) * HB + 2 * l + px) * 32 + i;
    }
    // OUTBITS: lane L of `wordv` collects the bit plane word of big pixel bpix(L): L = mt * 32 + hh * 16 + e <-> D-fragment
    // row e of M-tile mt, lane half hh
    uint32_t wordv = 0;
    int bpix = 0;
    if (OUTBITS) {
      const int e_ = lane & 15, hh = (lane >> 4) & 1, mt_ = lane >> 5;
      const int pp = mt_ * 32 + 4 * hh + (e_ & 3) + 8 * (e_ >> 2);
      bpix = (2 * (pp / HS) + py) * HB + 2 * (pp % HS) + px;
    }
    struct Acc { f32x16 a0, a1; };
    // bias / activation of D-fragment rows [e0, e1) of a finished unit -> its output image (lanes 0-31 / 32-63 of a
    // store: two pixels x 32 channels)
    auto epilogue = [&](const Acc& P, float* ob, int e0, int e1) {
#pragma unroll
      for (int e = e0; e < e1; ++e) {
        // pixel pp = mt*32 + 4h + (e&3) + 8*(e>>2); 4h + (e&3) + 8*((e>>2)&1) never carries into the next small row
        const int dpp = (e & 3) + 8 * (e >> 2);
        const int dm = dpp / HS, dl = dpp % HS;
        const int d = (2 * dm * HB + 2 * dl) * 32;
        const float v0 = epilogue_act(P.a0[e] + bv, act), v1 = epilogue_act(P.a1[e] + bv, act);
        ob[ooff[0] + d] = v0;
        ob[ooff[1] + d] = v1;
        if (OUTBITS) {
          // one ballot per D-fragment row: its two words (lanes 0-31 / 32-63 = the 32 channels of two pixels) go to lanes
          // mt * 32 + e and mt * 32 + 16 + e of `wordv` -- v_writelane_b32 from the scalar pair, no exec juggling and no
          // branch inside the software-pipelined MFMA loop (s_nop: a VALU-written SGPR read by v_writelane)
          const unsigned long long b0 = __builtin_amdgcn_ballot_w64(v0 > 0.f), b1 = __builtin_amdgcn_ballot_w64(v1 > 0.f);
          // (timing ablations, profiles/r04_v20_outbits_abl.txt, B = 1024: plain forward 75.5 us, + the 32 ballots 80.6,
          // + the 64 v_writelane 87.3 -- 90.8 with lane-0 ds_write_b32 under an exec mask instead)
          // (the hazard is real: without wait states behind the v_cmp that wrote the SGPR pair the written words are wrong --
          // tests/test_gpu_mask_bits.py, profiles/r06_v9_ab4.txt; ONE s_nop 3 in front of the four writes of a row instead of one each:
          // 80.4 -> 75.0 us at 1024 images, profiles/r06_v11_ab5.txt)
          asm("s_nop 3\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %0, %3, %7\n\tv_writelane_b32 %0, %4, %8"
              : "+v"(wordv)
              : "s"((uint32_t)b0), "s"((uint32_t)(b0 >> 32)), "s"((uint32_t)b1), "s"((uint32_t)(b1 >> 32)), "n"(e), "n"(16 + e), "n"(32 + e),
                "n"(48 + e));
        }
      }
      // the unit's 64 words of this wave: output image `ob` is out0 + b * UPWS_OUT_FLOATS, its bit image bimg + b * 256
      if (OUTBITS && e1 == 16) bimg[((ob - out0) >> 5) + bpix] = wordv;
    };
    // One unit: 128 MFMAs into C from the input image `in`; the PREVIOUS unit's results P are finished (bias, ReLU, LDS
    // image `pob`) in the shadow of the first 8 MFMA groups -- the matrix core never waits for an epilogue.
    auto unit_body = [&](Acc& C, const Acc& P, const float* in, float* pob, bool have_prev) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { C.a0[e] = 0.f; C.a1[e] = 0.f; }
      f32x4 A0[2], A1[2];
      auto rd = [&](int g, int slot) {
        const int t = g >> 2, q = g & 3;
        A0[slot] = *reinterpret_cast<const f32x4*>(in + aoff[0][t][q]);
        A1[slot] = *reinterpret_cast<const f32x4*>(in + aoff[1][t][q]);
      };
      rd(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int cur = g & 1;
        const int t = g >> 2, q = g & 3;
        if (g + 1 < 16) rd(g + 1, cur ^ 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          C.a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[cur][j], Bq[t][q][j], C.a0, 0, 0, 0);
          C.a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[cur][j], Bq[t][q][j], C.a1, 0, 0, 0);
        }
        if (have_prev && g < 8) epilogue(P, pob, 2 * g, 2 * g + 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads (next group)
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMAs (this group)
        if (have_prev && g < 8) {
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);   // 8 VALU (bias + activation of 4 results)
          __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);   // 4 DS writes
        }
      }
    };
    __builtin_amdgcn_s_setprio(1);
    Acc X, Y;
    int unit = unit0, k = 0;
    // iteration k computes unit k and writes the results of unit k-1 into output image (k-1)&1; X / Y alternate
    unit_body(X, X, in0, nullptr, false);
    __syncthreads();
    unit += stride; k = 1;
    for (;;) {
      if (unit >= n_units) break;
      unit_body(Y, X, in0 + (k & 1) * G::SH_FLOATS, out0 + ((k - 1) & 1) * UPWS_OUT_FLOATS, true);
      __syncthreads();
      unit += stride; ++k;
      if (unit >= n_units) { X = Y; break; }         // (one copy of the tail epilogue: 32 register moves, once per kernel)
      unit_body(X, Y, in0 + (k & 1) * G::SH_FLOATS, out0 + ((k - 1) & 1) * UPWS_OUT_FLOATS, true);
      __syncthreads();
      unit += stride; ++k;
    }
    epilogue(X, out0 + ((k - 1) & 1) * UPWS_OUT_FLOATS, 0, 16);   // the last unit's results
    __syncthreads();                                 // the last unit's image is visible to the memory waves
  } else {
    // ---------------------------------------------------------------- memory waves
    const unsigned bofs = 4 * (ht & 7);
    f32x4 mk[MASK == 1 ? 8 : 1];
    uint32_t mw[MASK == 2 ? 8 : 1];                  // bit plane words of this thread's 8 chunks (chunk = 4 channels of a pixel)
    auto drain = [&](int u, int b) {
      // unit u's output block is contiguous: out + u * 8192 floats
      const float* ob = out0 + b * UPWS_OUT_FLOATS;
      float* dst = out + (long)u * UPWS_OUT_FLOATS;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = (ht + 256 * j) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(ob + c);
        if (MASK == 1) {
#pragma unroll
          for (int x = 0; x < 4; ++x) v[x] = mk[j][x] > 0.f ? v[x] : 0.f;
        }
        if (MASK == 2) {
          // channels 4 (ht & 7) .. + 3 of pixel (ht + 256 j) / 8: bit -> 0 / all-ones (v_bfe_i32 at a per-thread constant
          // offset), then AND: two instructions per element, like the fp32 mask's compare + select
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int on = __builtin_amdgcn_sbfe((int)mw[j], bofs + x, 1);
            const float f = v[x];
            v[x] = __int_as_float(__float_as_int(f) & on);
          }
        }
        if (!(abl & 1)) *reinterpret_cast<f32x4*>(dst + c) = v;
      }
      if (OUTBITS) bits[(long)u * 256 + ht] = bimg[b * 256 + ht];
    };
    // iteration k (k = 0 .. K, K = number of units of this workgroup): input image of unit k+1 <- registers, loads of unit
    // k+2; output image of unit k-2 -> HBM (the compute waves write unit k-1's image during this iteration); mask of
    // unit k-1 -> registers.  Vector-memory operations retire in order (one vmcnt counter for loads and stores on gfx950):
    // the tile is consumed and re-requested BEFORE this iteration's 32 KB of output stores are issued.
    int unit = unit0, k = 0, u1 = -1, u2 = -1;       // u1 / u2: units k-1 / k-2
    for (;;) {
      const bool have = unit < n_units;
      if (have) {
        if (unit + stride < n_units) store_in(unit + stride, in0 + ((k + 1) & 1) * G::SH_FLOATS);
        load_in(unit + 2 * stride);
      }
      if (u2 >= 0) drain(u2, k & 1);
      if (MASK == 1 && u1 >= 0 && !(abl & 2)) {
        const float* src = mask + (long)u1 * UPWS_OUT_FLOATS;
#pragma unroll
        for (int j = 0; j < 8; ++j) mk[j] = *reinterpret_cast<const f32x4*>(src + (ht + 256 * j) * 4);
      }
      if (MASK == 2 && u1 >= 0) {
        const uint32_t* src = bits + (long)u1 * 256;
#pragma unroll
        for (int j = 0; j < 8; ++j) mw[j] = src[(ht >> 3) + 32 * j];
      }
      __syncthreads();
      if (!have) break;
      u2 = u1; u1 = unit; unit += stride; ++k;
    }
    if (u1 >= 0) drain(u1, (k - 1) & 1);
  }
}

template <int HS>
static int launch_up_ws_t(const ConvArgs& a, hipStream_t s) {
  using G = Geo<HS>;
  const int n_units = (int)(((long)a.N * HS * HS) / 64);     // HS*HS is a multiple of 64: every unit is complete
  const int grid = n_units < 256 ? n_units : 256;
  static_assert(2 * UPWS_OUT_FLOATS == 16384, "the two output images occupy exactly the weight image");
  const size_t lds = (size_t)(16384 + 2 * G::SH_FLOATS + 512) * sizeof(float);   // + the two 1 KB bit images (OUTBITS)
  static DeviceOnce attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute((const void*)k_up32ws<HS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k_up32ws<HS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (HS == 16) {
      (void)hipFuncSetAttribute((const void*)k_up32ws<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)k_up32ws<16, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
  }
  static const int abl = env_int("DVAE_UPWS_ABLATE", 0);      // debug builds only
  const int af = a.act | (abl << 8);
  if (a.mask_bits || a.out_bits) {
    // bit planes: the 16 -> 32 geometry only (the two 134 MB activations of the B = 1024 step: conv1's and convT2's outputs)
    if (HS != 16 || (a.mask_bits && (a.mask || a.out_bits)) || (a.out_bits && (a.mask || a.act != DVAE_ACT_RELU))) return 1;
    if (a.mask_bits)
      hipLaunchKernelGGL((k_up32ws<16, 2>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, af, n_units,
                         a.w_staged, const_cast<uint32_t*>(a.mask_bits));
    else
      hipLaunchKernelGGL((k_up32ws<16, 0, true>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, af, n_units,
                         a.w_staged, a.out_bits);
  } else if (a.mask) {
    hipLaunchKernelGGL((k_up32ws<HS, 1>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, af, n_units, a.w_staged, (uint32_t*)nullptr);
  } else {
    hipLaunchKernelGGL((k_up32ws<HS, 0>), dim3(grid), dim3(512), lds, s, a.small, a.w, a.bias, a.mask, a.out, af, n_units, a.w_staged, (uint32_t*)nullptr);
  }
  DVAE_CHECK_LAUNCH();
  return 0;
}

// 32 <-> 32 channels, NHWC on both sides, Hs == Ws in {8, 16}; returns 1 if not applicable
int launch_up_mfma32_ws(const ConvArgs& a, hipStream_t s) {
  if (!(a.Cb == 32 && a.Cs == 32 && a.Hs == a.Ws && (a.Hs == 8 || a.Hs == 16) && a.small_layout == DVAE_NHWC &&
        a.out_layout == DVAE_NHWC))
    return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  return a.Hs == 16 ? launch_up_ws_t<16>(a, s) : launch_up_ws_t<8>(a, s);
}

}  // namespace dvae
