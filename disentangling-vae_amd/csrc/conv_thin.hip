// gfx950 kernels for the two "thin" ends of the Burgess network at 64x64 images:
//   conv1  : x[N,C,64,64] (NCHW, C in {1,3}) -> 32ch 32x32 (NHWC)      encoders.py:54,73
//   convT3 : 32ch 32x32 (NHWC) -> [N,C,64,64] (NCHW) + sigmoid          decoders.py:65,82
// and their backward passes.  With C in {1,3} these layers are bandwidth-shaped:
//   down_thin  (conv1 fwd, convT3 dgrad): K = 16*C; MFMA 32x32x2 f32 with M = 32 pixels of one
//              output row, N = 32 channels; weights live in 8*C VGPRs per lane.
//   up_thin    (convT3 fwd): N = C output channels is too narrow for the matrix core; VALU
//              kernel, one thread per small pixel producing its 2x2xC outputs, weights are
//              wave-uniform (scalar loads -> SGPR operands), inputs staged in swizzled LDS.
//   wgrad_thin (conv1 / convT3 wgrad): M = 32 cs, N = 16*C (cb,tap) columns, K = pixels.
#include <stdlib.h>
#include "common.h"
#include "wgrad_reduce.h"

namespace dvae {

// LDS image of the big (NCHW) tile: [cb][row][par][cw], conflict-free strides
#define TB_ROWS 10                // 4 small rows -> 10 big rows with halo
#define TB_PAR 34                 // floats per parity half-row (33 used)
#define TB_ROW 68                 // floats per row
#define TB_PLANE (TB_ROWS * TB_ROW + 16)

// Big rows [2*sy0-1, 2*sy0+8] of image n, channels [0,C), zero padded.  256 threads: 64 consecutive
// threads read one 256-byte image row (coalesced), 4 (channel,row) pairs per pass.  Loads are
// UNCONDITIONAL (address clamped, value zeroed afterwards) so that the compiler issues them
// back to back instead of one branch + wait per element; load and LDS-store are separate so the
// next unit can be prefetched into registers during the MFMA phase.
template <int C, typename TB = float>
struct BigThinRegs {
  TB v[(C * TB_ROWS + 3) / 4];     // RAW elements as loaded (uint8 pixels are converted when the tile is written to LDS:
  unsigned ok;                     // nothing may depend on a prefetched register before the MFMA phase is over)
};

// Input elements: fp32, or uint8 pixels as the datasets store them (dSprites imgs * 255, CelebA imread:
// utils/datasets.py:204-213,282-291) converted on the fly with ToTensor's arithmetic, float(v) / 255 (IEEE
// division: bit-identical to torchvision's .div(255)).  The batch then stays uint8 in HBM: the three kernels that
// read the input image (conv1 forward, conv1 weight gradient, reconstruction likelihood) fetch 1 byte per pixel.
__device__ __forceinline__ float to_unit(float v) { return v; }
__device__ __forceinline__ float to_unit(uint8_t v) { return (float)v / 255.0f; }
// the 256 possible results as a table in LDS: one ds_read instead of the ~10-instruction IEEE division per pixel
// (the staging code of these kernels is instruction-issue-bound, not bandwidth-bound)
struct UnitLut {
  float t[256];
  __device__ __forceinline__ void init(int tid, int nthr) {
    for (int k = tid; k < 256; k += nthr) t[k] = (float)k / 255.0f;
  }
};
__device__ __forceinline__ float to_unit(float v, const UnitLut*) { return v; }
__device__ __forceinline__ float to_unit(uint8_t v, const UnitLut* lut) { return lut->t[v]; }

template <int C, typename TB>
__device__ __forceinline__ void load_big_thin(BigThinRegs<C, TB>& r, const TB* __restrict__ big, int n, int sy0,
                                              bool valid, int tid) {
  const int tx = tid & 63, ty = tid >> 6;
  unsigned okm = 0;
#pragma unroll
  for (int k = 0; k < (C * TB_ROWS + 3) / 4; ++k) {
    const int pr = ty + 4 * k;
    const int cb = pr / TB_ROWS, rr = pr - cb * TB_ROWS;
    const int by = 2 * sy0 - 1 + rr;
    const bool ok = valid && pr < C * TB_ROWS && by >= 0 && by < 64;
    const long off = ok ? ((((long)n * C + cb) * 64 + by) * 64 + tx) : 0;
    r.v[k] = big[off];
    okm |= ok ? (1u << k) : 0u;
  }
  r.ok = okm;
}

template <int C, typename TB>
__device__ __forceinline__ void store_big_thin(const BigThinRegs<C, TB>& r, float* bt, int tid, const UnitLut* lut) {
  const int tx = tid & 63, ty = tid >> 6;
  const int pc = tx + 1;
  const int dst_col = (pc & 1) * TB_PAR + (pc >> 1);
#pragma unroll
  for (int k = 0; k < (C * TB_ROWS + 3) / 4; ++k) {
    const int pr = ty + 4 * k;
    if (pr < C * TB_ROWS) {
      const int cb = pr / TB_ROWS, rr = pr - cb * TB_ROWS;
      bt[cb * TB_PLANE + rr * TB_ROW + dst_col] = ((r.ok >> k) & 1u) ? to_unit(r.v[k], lut) : 0.f;
    }
  }
  if (tid < C * TB_ROWS * 2) {   // the two zero-padding columns pc = 0 and pc = 65
    const int pr = tid >> 1, side = tid & 1;
    const int cb = pr / TB_ROWS, rr = pr - cb * TB_ROWS;
    bt[cb * TB_PLANE + rr * TB_ROW + (side ? (TB_PAR + 32) : 0)] = 0.f;
  }
}

// ---- down_thin: big NCHW [N,C,64,64] -> small NHWC [N,32,32,32] ----------------------------
// persistent workgroups (4 waves = 4 small rows of 32 pixels per unit); weights go through LDS
// once per workgroup (coalesced read, transposed [k][cs] image) into 8*C VGPRs per lane; the
// next unit's big tile is prefetched into registers during the MFMA phase.
// MODE: 0 plain; 1 output masked by an fp32 activation (`mask`); 2 output masked by its bit plane (`bits`: one uint32 per
// output pixel, bit c = [activation channel c > 0]: 128 bytes per wave and row instead of 4 KB); 3 plain + EMIT the bit
// plane of the output (the forward pass of conv1: consumed by conv2's input-gradient kernel, conv_up_ws.hip).
// (the fp32-mask form keeps 16 mask values in flight over the MFMA phase: at C = 3 it needs more than the 128 registers of four
// waves per SIMD -- it spilled 12 bytes -- so it is built for three; the training step uses MODE 2)
template <int C, int MODE, typename TB = float>
__global__ __launch_bounds__(256, (MODE == 1 && C == 3) ? 3 : 4) void k_down_thin(const TB* __restrict__ big, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ mask,
                                                   float* __restrict__ out, int N, int act, int n_units,
                                                   uint32_t* __restrict__ bits) {
  constexpr bool MASK = MODE == 1;
  __shared__ float bt[C * TB_PLANE];
  __shared__ float wT[16 * C * 32];
  __shared__ UnitLut lut_s[1];
  const UnitLut* lut = lut_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  if (sizeof(TB) == 1) { lut_s[0].init(tid, 256); __syncthreads(); }
  {                                                      // w[cs][k] -> wT[k][cs]; loads first, then LDS writes
    float wv[2 * C];
#pragma unroll
    for (int r = 0; r < 2 * C; ++r) wv[r] = w[tid + r * 256];
#pragma unroll
    for (int r = 0; r < 2 * C; ++r) {
      const int e = tid + r * 256;
      const int cs = e / (16 * C), k = e - cs * (16 * C);
      wT[k * 32 + cs] = wv[r];
    }
  }
  // TWO tiles in flight per workgroup (units u + g and u + 2g while unit u is multiplied): with one, the kernel sat at
  // 3.9 TB/s of algorithmic traffic with every CU's memory queue a latency deep (6 workgroups x 8 KB per CU); the loads
  // are 8 registers per thread and tile
  BigThinRegs<C, TB> pfa, pfb;
  int unit = blockIdx.x;
  const int gstep = gridDim.x;
  if (unit < n_units) load_big_thin<C, TB>(pfa, big, unit >> 3, (unit & 7) * 4, true, tid);
  if (unit + gstep < n_units) load_big_thin<C, TB>(pfb, big, (unit + gstep) >> 3, ((unit + gstep) & 7) * 4, true, tid);
  __syncthreads();
  float wreg[8 * C];                                     // B operand: w[cs = i][k = 2*kk + h]
#pragma unroll
  for (int kk = 0; kk < 8 * C; ++kk) wreg[kk] = wT[(2 * kk + h) * 32 + i];
  const float bv = bias ? bias[i] : 0.f;
  const int sy_l = wv;  // this wave's small row inside the unit; lane i = sx

  auto body = [&](BigThinRegs<C, TB>& pf, int u) {
    const int n = u >> 3, sy0 = (u & 7) * 4;
    __syncthreads();                                     // previous unit's LDS reads are complete
    store_big_thin<C, TB>(pf, bt, tid, lut);
    __syncthreads();
    const int nu = u + 2 * gstep;
    if (nu < n_units) load_big_thin<C, TB>(pf, big, nu >> 3, (nu & 7) * 4, true, tid);
    const long rowbase = ((((long)n * 32 + sy0 + sy_l) * 32)) * 32 + i;
    float mv[16];
    if (MASK) {
#pragma unroll
      for (int e = 0; e < 16; ++e) mv[e] = mask[rowbase + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32];
    }
    const long pixbase = ((long)n * 32 + sy0 + sy_l) * 32;         // this wave's row of 32 pixels
    uint32_t word = 0;                                              // lane L < 32: the bit plane word of pixel L
    if (MODE == 2) word = bits[pixbase + i];
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8 * C; ++kk) {
      // k = 2kk + h : kw = 2*(kk&1) + h -> par = h, cw = sx + (kk&1); kh = (kk>>1)&3; cb = kk>>3
      const int kwb = (kk & 1);
      const int kh = (kk >> 1) & 3, cb = kk >> 3;
      const float a = bt[cb * TB_PLANE + (2 * sy_l + kh) * TB_ROW + h * TB_PAR + i + kwb];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[kk], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int sx0 = (e & 3) + 8 * (e >> 2);                      // lanes 0-31 hold pixel sx0, lanes 32-63 pixel sx0 + 4
      const int sx = sx0 + 4 * h;
      float v = acc[e] + bv;
      if (act == DVAE_ACT_RELU) v = v > 0.f ? v : 0.f;
      if (MASK) v = mv[e] > 0.f ? v : 0.f;
      if (MODE == 2) {
        // the words of the two pixels ARE the lane mask of this register: lane l < 32 = channel l of pixel sx0 (bit l of its
        // word), lane 32 + l = channel l of pixel sx0 + 4 -- two v_readlane and one v_cndmask on the scalar pair
        const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(word, sx0 + 4) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane(word, sx0);
        v = __builtin_amdgcn_inverse_ballot_w64(m) ? v : 0.f;
      }
      if (MODE == 3) {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(v > 0.f);   // bit l = lane l: low half pixel sx0, high half sx0 + 4
        // lane sx0 keeps the low word, lane sx0 + 4 the high one (constant one-hot lane masks: a scalar move + a v_cndmask each)
        word = __builtin_amdgcn_inverse_ballot_w64(0x0000000100000001ull << sx0) ? (uint32_t)b : word;
        word = __builtin_amdgcn_inverse_ballot_w64(0x0000000100000001ull << (sx0 + 4)) ? (uint32_t)(b >> 32) : word;
      }
      out[rowbase + sx * 32] = v;
    }
    if (MODE == 3 && h == 0) bits[pixbase + i] = word;
  };
  while (unit < n_units) {
    body(pfa, unit);
    unit += gstep;
    if (unit >= n_units) break;
    body(pfb, unit);
    unit += gstep;
  }
}

// ---- up_thin: small NHWC [N,32,32,32] -> big NCHW [N,C,64,64], bias + act ------------------
// block = 128 threads = 4 small rows x 32 columns; LDS tile = 6 rows x 34 cols x 32 ch (swizzled);
// persistent over units.  FUSE: the sigmoid output is compared with the target on the spot -- the
// reconstruction likelihood, its per-workgroup partial sum and dL/dlogit come out of the same pass.
#define UT_ROWS 6
#define UT_COLS 34
template <int C, bool FUSE, typename TT = float>
__global__ __launch_bounds__(128) void k_up_thin(const float* __restrict__ small, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ out, int N,
                                                 int act, int n_units, const TT* __restrict__ target,
                                                 float* __restrict__ g, int dist, const float* __restrict__ coef,
                                                 float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float st[UT_ROWS * UT_COLS * 32];
  __shared__ float redl[2];
  const int tid = threadIdx.x;
  const int m = tid >> 5, l = tid & 31;
  const float gs = FUSE ? coef[DVAE_C_INV_B] : 0.f;
  float lsum = 0.f;
  // Consecutive workgroups run on different XCDs (each with its own L2), and the 8 units of an image share halo rows: with the
  // plain round-robin order they were fetched from HBM by two XCDs each (counter traffic 1.32x the algorithmic bytes).  When
  // the grid is a multiple of 64, workgroup b = (xcd = b & 7, slot = b >> 3) takes part slot & 7 of image xcd + 8 (slot >> 3)
  // + (grid / 8) k: the 8 parts of an image run at the same time on ONE XCD.
  const bool xmap = (gridDim.x & 63) == 0;
  const int upi = xmap ? 8 * (int)(gridDim.x >> 3) : (int)gridDim.x;           // units per persistent step
  const int unit0 = xmap ? 8 * (int)((blockIdx.x & 7) + 8 * (blockIdx.x >> 6)) + (int)((blockIdx.x >> 3) & 7) : (int)blockIdx.x;
  for (int unit = unit0; unit < n_units; unit += upi) {
    const int n = unit >> 3, sy0 = (unit & 7) * 4;
    __syncthreads();
    {
      // 16 columns per pass, 8 chunks of 16 bytes per pixel; all loads first (unconditional, clamped
      // address), then the swizzled LDS stores
      const int chunk = tid & 7, cg = tid >> 3;
      f32x4 tv[UT_ROWS * 3];
#pragma unroll
      for (int row = 0; row < UT_ROWS; ++row) {
        const int sy = sy0 - 1 + row;
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
          const int col = cp * 16 + cg;
          const int sx = col - 1;
          const bool ok = n < N && col < UT_COLS && sy >= 0 && sy < 32 && sx >= 0 && sx < 32;
          const long off = ok ? (((((long)n * 32 + sy) * 32) + sx) * 32 + chunk * 4) : 0;
          f32x4 v = *reinterpret_cast<const f32x4*>(small + off);
          if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
          tv[row * 3 + cp] = v;
        }
      }
#pragma unroll
      for (int row = 0; row < UT_ROWS; ++row)
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
          const int col = cp * 16 + cg;
          if (col < UT_COLS)
            *reinterpret_cast<f32x4*>(st + (row * UT_COLS + col) * 32 + ((chunk ^ ((col >> 1) & 7)) << 2)) = tv[row * 3 + cp];
        }
    }
    __syncthreads();
    float acc[4][C];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int cb = 0; cb < C; ++cb) acc[c][cb] = 0.f;

#pragma unroll 1
    for (int q = 0; q < 8; ++q) {   // 4 contracted channels per iteration
      f32x4 x[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int row = m + dy, col = l + dx;  // (m + (dy-1)) + 1
          x[dy][dx] = *reinterpret_cast<const f32x4*>(st + (row * UT_COLS + col) * 32 + ((q ^ ((col >> 1) & 7)) << 2));
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cs = q * 4 + j;
#pragma unroll
        for (int cb = 0; cb < C; ++cb) {
          const float* wp = w + (cs * C + cb) * 16;   // wave-uniform -> scalar loads
#pragma unroll
          for (int kh = 0; kh < 4; ++kh) {
            // big row by = 2*sy - 1 + kh: py = (kh+1)&1, small row offset dsy = (py + 1 - kh) / 2  in {-1,0,1}
            const int py = (kh + 1) & 1;
            const int dsy = (py + 1 - kh) / 2;   // kh=0:+1 (py=1) kh=1:0 (py=0) kh=2:0 (py=1) kh=3:-1 (py=0)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
              const int px = (kw + 1) & 1;
              const int dsx = (px + 1 - kw) / 2;
              acc[py * 2 + px][cb] = fmaf(x[dsy + 1][dsx + 1][j], wp[kh * 4 + kw], acc[py * 2 + px][cb]);
            }
          }
        }
      }
    }
    if (n < N) {
      const int sy = sy0 + m;
#pragma unroll
      for (int cb = 0; cb < C; ++cb) {
        const float bv = bias ? bias[cb] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          float v0 = acc[py * 2 + 0][cb] + bv, v1 = acc[py * 2 + 1][cb] + bv;
          if (act == DVAE_ACT_SIGMOID) { v0 = 1.f / (1.f + expf(-v0)); v1 = 1.f / (1.f + expf(-v1)); }
          else if (act == DVAE_ACT_RELU) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
          const long o = ((((long)n * C + cb) * 64) + 2 * sy + py) * 64 + 2 * l;
          *reinterpret_cast<float2*>(out + o) = make_float2(v0, v1);
          if (FUSE) {
            float xt0, xt1;
            if constexpr (sizeof(TT) == 4) {
              const float2 xt = *reinterpret_cast<const float2*>(target + o);
              xt0 = xt.x; xt1 = xt.y;
            } else {
              const uchar2 xt = *reinterpret_cast<const uchar2*>(target + o);
              xt0 = to_unit(xt.x); xt1 = to_unit(xt.y);      // two IEEE divisions per output pair: noise next to ~100 FMAs
            }
            float gl0, gl1, gr;
            lsum += recon_elem(v0, xt0, dist, &gl0, &gr);
            lsum += recon_elem(v1, xt1, dist, &gl1, &gr);
            *reinterpret_cast<float2*>(g + o) = make_float2(gs * gl0, gs * gl1);
          }
        }
      }
    }
  }
  if (FUSE) {
    const float v = wave_sum(lsum);
    if ((tid & 63) == 0) redl[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = redl[0] + redl[1];
    // unused partial slots must read as zero
    for (int k = gridDim.x + blockIdx.x * 128 + tid; k < DVAE_REC_NPART; k += gridDim.x * 128) partials[k] = 0.f;
  }
}

// ---- up_thin on packed fp32 FMAs (round 3) ---------------------------------------------------------------------------
// Same layer, same decomposition (thread = one small pixel -> its 2 x 2 x C outputs, weights wave-uniform in SGPRs,
// swizzled halo tile in LDS), but the 16 C multiply-adds per contracted channel are issued as v_pk_fma_f32: two fp32 FMAs
// per lane and instruction with the activation broadcast to both halves by op_sel and the two weights an SGPR pair.  k_up_thin
// is bound by VALU issue (48 v_fma_f32 per channel at C = 3 next to the likelihood's arithmetic: 0.35 of the HBM roof,
// profiles/r02_final_pmc_summary.md); paired:
//   * channels (cb0, cb1) of the same output and tap share the activation                         -> 16 packed FMAs,
//   * the remaining channel plane (cb2; the only one for C = 1): two taps that read the SAME source pixel feed two different
//     output classes -- the own pixel 4 taps (2 pairs), the 4 edge neighbours 2 taps each (4 pairs), the 4 corner
//     neighbours one tap each (4 scalar FMAs)                                                    -> 6 packed + 4 scalar,
// 26 instructions per channel instead of 48 (10 instead of 16 for C = 1).  The pairs' weights must sit in adjacent SGPRs, so
// the layer's weights are read from a per-channel record in pair order (DVAE_THIN_PAIR_FLOATS(C) floats per contracted
// channel) that dvae_stage_weights writes once per step.  Every output is still a fixed-order fmaf chain per accumulator;
// the cb2 plane adds three partial accumulators (row pairs, column pairs, corners) at the end.
//   record of channel cs, C = 3:  [2 (4 t + cls) + {0,1}] = w[cs][{0,1}][kh][kw] with cls = 2 py + px, t = 2 ty + tx,
//                                  kh = 1 - py + 2 ty, kw = 1 - px + 2 tx;  then the 16 floats of plane cb2 (C = 1: only these,
//                                  of plane cb0) in the tap order below.
// tap order of the single-plane part (thin_pair_source, common.h): {5, 6, 9, 10, 13, 14, 1, 2, 7, 11, 4, 8, 0, 3, 12, 15} =
// (kh*4+kw): H pairs (cls0,cls1)/(cls2,cls3) from the own pixel: (1,1),(1,2) | (2,1),(2,2); from the pixel above / below:
// (3,1),(3,2) | (0,1),(0,2); V pairs (cls0,cls2) from the left: (1,3),(2,3); (cls1,cls3) from the right: (1,0),(2,0);
// corners: (0,0) -> cls3, (0,3) -> cls2, (3,0) -> cls1, (3,3) -> cls0

typedef float f32x2 __attribute__((ext_vector_type(2)));

// What bounds this kernel (timing ablations of a debug build, profiles/r03_v10_thin_abl.txt, B = 1024, 93 us complete): the
// FMA core alone 44 us; tile loads 16, LDS operand reads 17, the range-reduced expf + IEEE division of the sigmoid 15, the
// scalar loads of the weight records 14, barriers 8, stores 5 -- a wave is latency-bound on all of them in turn (six
// s_waitcnt lgkmcnt(0) per 4 channels behind ~200-cycle scalar loads), which is why halving the FMA instructions with
// v_pk_fma_f32 bought 0-3 % (profiles/r03_v6_kbench.txt; the packed form itself runs at 1.8x the v_fma_f32 rate,
// profiles/r03_v9_valu_fma_rate.txt).  Kept from that analysis: the hardware sigmoid, the likelihood's targets requested
// before the FMA loop instead of one dependent round trip per output pair, tap-major records so that the first 16 weights of
// a channel feed four accumulator chains.  Measured and NOT kept (parity green, not in the tree): the tile in two
// 16-channel halves moved by LDS-DMA (global_load_lds_dwordx4 from inline asm, the next half in flight under the FMAs, same
// 26 KB of LDS) -- 113 / 134 us against 70 / 93 us for this kernel on the same box (profiles/r03_v14_thin_abl.txt): with the
// loads off the critical path a wave is paced by its scalar weight loads alone (8.2 us per unit for ONE resident workgroup
// per CU with tile loads, LDS reads, sigmoid and stores all removed), and twice as many barrier-delimited phases per unit
// leave the three waves of a SIMD fewer chances to cover each other's stalls.
// torch.sigmoid on the hardware transcendentals: common.h's sigmoid_aten (v_exp_f32, v_rcp_f32 + one FMA that lands on ATen's
// rounding of p where the Bernoulli term depends on it; the IEEE division and range-reduced expf of the synchronous kernel
// cost 15 us of 93 at B = 1024)
__device__ __forceinline__ float sigmoid_hw(float v) { return sigmoid_aten(v); }

template <int C, bool FUSE, typename TT = float>
__global__ __launch_bounds__(128) void k_up_thin_pk(const float* __restrict__ small, const float* __restrict__ wrec,
                                                    const float* __restrict__ bias, float* __restrict__ out, int N,
                                                    int act, int n_units, const TT* __restrict__ target,
                                                    float* __restrict__ g, int dist, const float* __restrict__ coef,
                                                    float* __restrict__ partials) {
  constexpr int REC = C == 3 ? 48 : 16;             // DVAE_THIN_PAIR_FLOATS(C)
  constexpr int PB = C == 3 ? 32 : 0;               // offset of the single-plane part in a record
  constexpr int CS = C - 1;                         // the plane handled by the tap-pair scheme
  __shared__ __attribute__((aligned(16))) float st[UT_ROWS * UT_COLS * 32];
  __shared__ float redl[2];
  const int tid = threadIdx.x;
  const int m = tid >> 5, l = tid & 31;
  const float gs = FUSE ? coef[DVAE_C_INV_B] : 0.f;
  const bool bce_logit = FUSE && dist == DVAE_REC_BERNOULLI && act == DVAE_ACT_SIGMOID;
  float lsum = 0.f;
  // Consecutive workgroups run on different XCDs (each with its own L2), and the 8 units of an image share halo rows: with the
  // plain round-robin order they were fetched from HBM by two XCDs each (counter traffic 1.32x the algorithmic bytes).  When
  // the grid is a multiple of 64, workgroup b = (xcd = b & 7, slot = b >> 3) takes part slot & 7 of image xcd + 8 (slot >> 3)
  // + (grid / 8) k: the 8 parts of an image run at the same time on ONE XCD.
  const bool xmap = (gridDim.x & 63) == 0;
  const int upi = xmap ? 8 * (int)(gridDim.x >> 3) : (int)gridDim.x;           // units per persistent step
  const int unit0 = xmap ? 8 * (int)((blockIdx.x & 7) + 8 * (blockIdx.x >> 6)) + (int)((blockIdx.x >> 3) & 7) : (int)blockIdx.x;
  for (int unit = unit0; unit < n_units; unit += upi) {
    const int n = unit >> 3, sy0 = (unit & 7) * 4;
    __syncthreads();
    {
      // 16 columns per pass, 8 chunks of 16 bytes per pixel; all loads first (unconditional, clamped
      // address), then the swizzled LDS stores
      const int chunk = tid & 7, cg = tid >> 3;
      f32x4 tv[UT_ROWS * 3];
#pragma unroll
      for (int row = 0; row < UT_ROWS; ++row) {
        const int sy = sy0 - 1 + row;
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
          const int col = cp * 16 + cg;
          const int sx = col - 1;
          const bool ok = n < N && col < UT_COLS && sy >= 0 && sy < 32 && sx >= 0 && sx < 32;
          const long off = ok ? (((((long)n * 32 + sy) * 32) + sx) * 32 + chunk * 4) : 0;
          f32x4 v = *reinterpret_cast<const f32x4*>(small + off);
          if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
          tv[row * 3 + cp] = v;
        }
      }
#pragma unroll
      for (int row = 0; row < UT_ROWS; ++row)
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
          const int col = cp * 16 + cg;
          if (col < UT_COLS)
            *reinterpret_cast<f32x4*>(st + (row * UT_COLS + col) * 32 + ((chunk ^ ((col >> 1) & 7)) << 2)) = tv[row * 3 + cp];
        }
    }
    __syncthreads();
    float2 tg[C][2];                                 // the likelihood's targets travel under the FMAs
    if (FUSE && n < N) {
#pragma unroll
      for (int cb = 0; cb < C; ++cb)
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const long o = ((((long)n * C + cb) * 64) + 2 * (sy0 + m) + py) * 64 + 2 * l;
          if constexpr (sizeof(TT) == 4) tg[cb][py] = *reinterpret_cast<const float2*>(target + o);
          else {
            const uchar2 xt = *reinterpret_cast<const uchar2*>(target + o);
            tg[cb][py] = make_float2((float)xt.x, (float)xt.y);
          }
        }
    }
    f32x2 accA[4];                                   // (cb0, cb1) of class cls                     (C = 3)
    f32x2 h01 = {0.f, 0.f}, h23 = {0.f, 0.f};        // plane CS: classes (0,1) / (2,3), sources in the own column
    f32x2 v02 = {0.f, 0.f}, v13 = {0.f, 0.f};        // plane CS: classes (0,2) / (1,3), sources left / right
    float sc[4] = {0.f, 0.f, 0.f, 0.f};              // plane CS: corner sources
#pragma unroll
    for (int c = 0; c < 4; ++c) accA[c] = f32x2{0.f, 0.f};

#pragma unroll 1
    for (int q = 0; q < 8; ++q) {   // 4 contracted channels per iteration
      f32x4 x[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int row = m + dy, col = l + dx;  // (m + (dy-1)) + 1
          x[dy][dx] = *reinterpret_cast<const f32x4*>(st + (row * UT_COLS + col) * 32 + ((q ^ ((col >> 1) & 7)) << 2));
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* wp = wrec + (q * 4 + j) * REC;              // wave-uniform -> scalar loads, pairs in adjacent SGPRs
        const f32x2* wp2 = reinterpret_cast<const f32x2*>(wp);
        if (C == 3) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {                // tap-major: consecutive instructions feed four different accumulators
            const int ty = t >> 1, tx = t & 1;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
              const int py = cls >> 1, px = cls & 1;
              const float xs = x[py - ty + 1][px - tx + 1][j];
              accA[cls] = __builtin_elementwise_fma(f32x2{xs, xs}, wp2[t * 4 + cls], accA[cls]);
            }
          }
        }
        const f32x2* wb2 = reinterpret_cast<const f32x2*>(wp + PB);
        const float xc = x[1][1][j], xu = x[0][1][j], xd = x[2][1][j], xl = x[1][0][j], xr = x[1][2][j];
        h01 = __builtin_elementwise_fma(f32x2{xc, xc}, wb2[0], h01);
        h23 = __builtin_elementwise_fma(f32x2{xc, xc}, wb2[1], h23);
        h01 = __builtin_elementwise_fma(f32x2{xu, xu}, wb2[2], h01);
        h23 = __builtin_elementwise_fma(f32x2{xd, xd}, wb2[3], h23);
        v02 = __builtin_elementwise_fma(f32x2{xl, xl}, wb2[4], v02);
        v13 = __builtin_elementwise_fma(f32x2{xr, xr}, wb2[5], v13);
        sc[3] = fmaf(x[2][2][j], wp[PB + 12], sc[3]);
        sc[2] = fmaf(x[2][0][j], wp[PB + 13], sc[2]);
        sc[1] = fmaf(x[0][2][j], wp[PB + 14], sc[1]);
        sc[0] = fmaf(x[0][0][j], wp[PB + 15], sc[0]);
      }
    }
    float acc[4][C];
    {
      const float hh[4] = {h01[0], h01[1], h23[0], h23[1]}, vv[4] = {v02[0], v13[0], v02[1], v13[1]};
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        if (C == 3) { acc[cls][0] = accA[cls][0]; acc[cls][1] = accA[cls][1]; }
        acc[cls][CS] = (hh[cls] + vv[cls]) + sc[cls];
      }
    }
    if (n < N) {
      const int sy = sy0 + m;
#pragma unroll
      for (int cb = 0; cb < C; ++cb) {
        const float bv = bias ? bias[cb] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          float v0 = acc[py * 2 + 0][cb] + bv, v1 = acc[py * 2 + 1][cb] + bv;
          const long o = ((((long)n * C + cb) * 64) + 2 * sy + py) * 64 + 2 * l;
          if (FUSE && bce_logit) {
            // sigmoid + Bernoulli likelihood straight from the logit (common.h: sigmoid_bce_logit), workgroup-uniform branch
            float xt0 = tg[cb][py].x, xt1 = tg[cb][py].y;
            if constexpr (sizeof(TT) != 4) { xt0 = xt0 / 255.0f; xt1 = xt1 / 255.0f; }
            float p0, p1, gl0, gl1;
            lsum += sigmoid_bce_logit(v0, xt0, &p0, &gl0);
            lsum += sigmoid_bce_logit(v1, xt1, &p1, &gl1);
            *reinterpret_cast<float2*>(out + o) = make_float2(p0, p1);
            *reinterpret_cast<float2*>(g + o) = make_float2(gs * gl0, gs * gl1);
            continue;
          }
          if (act == DVAE_ACT_SIGMOID) { v0 = sigmoid_hw(v0); v1 = sigmoid_hw(v1); }
          else if (act == DVAE_ACT_RELU) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
          *reinterpret_cast<float2*>(out + o) = make_float2(v0, v1);
          if (FUSE) {
            float xt0 = tg[cb][py].x, xt1 = tg[cb][py].y;
            if constexpr (sizeof(TT) != 4) { xt0 = xt0 / 255.0f; xt1 = xt1 / 255.0f; }     // = to_unit(uint8_t): ToTensor's division
            float gl0, gl1, gr;
            lsum += recon_elem(v0, xt0, dist, &gl0, &gr);
            lsum += recon_elem(v1, xt1, dist, &gl1, &gr);
            *reinterpret_cast<float2*>(g + o) = make_float2(gs * gl0, gs * gl1);
          }
        }
      }
    }
  }
  if (FUSE) {
    const float v = wave_sum(lsum);
    if ((tid & 63) == 0) redl[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = redl[0] + redl[1];
    // unused partial slots must read as zero
    for (int k = gridDim.x + blockIdx.x * 128 + tid; k < DVAE_REC_NPART; k += gridDim.x * 128) partials[k] = 0.f;
  }
}

// convT3 forward on the staged pair records; target == NULL: no likelihood.  Returns 1 if the shape is not covered.
int launch_up_thin_staged(const float* small, const float* wrec, const float* bias, const void* target, int target_u8,
                          float* out, float* g, int dist, const float* coef, float* partials, int N, int C, int act,
                          hipStream_t s) {
  if (C != 1 && C != 3) return 1;
  // 3 channels: the matrix-core formulation (conv_up_thin_mm.hip) on the operand image behind the pair records
  if (C == 3 && launch_up_thin_mm(small, wrec + 32 * 48, bias, target, target_u8, out, g, dist, coef, partials, N, act, s) == 0)
    return 0;
  const int n_units = N * 8;
  const int grid = n_units < 1536 ? n_units : 1536;
#define UP_PK(CC, FF, TT_) hipLaunchKernelGGL((k_up_thin_pk<CC, FF, TT_>), dim3(grid), dim3(128), 0, s, small, wrec, bias, out, N, act, \
                                             n_units, (const TT_*)target, g, dist, coef, partials)
  if (!target) { if (C == 1) UP_PK(1, false, float); else UP_PK(3, false, float); }
  else if (target_u8) { if (C == 1) UP_PK(1, true, uint8_t); else UP_PK(3, true, uint8_t); }
  else { if (C == 1) UP_PK(1, true, float); else UP_PK(3, true, float); }
#undef UP_PK
  DVAE_CHECK_LAUNCH();
  return 0;
}

// (Round 3 measured a v_mfma_f32_4x4x1_16b_f32 formulation of this layer -- 16 blocks of 4 pixels x 4 output channels, the
// wave's 256 weights in registers, operand reads one step ahead: parity green, but 103.6 us without / 144.9 us with the
// likelihood at B = 1024 against 73 / 98 us here (profiles/r03_v4_kbench.txt).  The 4x4x1 form issues in 16 cycles, not 8:
// 32 FLOP/clk/SIMD, half the rate of the 16x16x4 / 32x32x2 forms, and 3 of its 4 columns carry data -- 4096 matrix-core
// cycles per 128 output pixels x 2 waves, no better than the VALU's 6144 x 2 waves spread over more resident waves.  Not
// shipped; git history: k_up_thin_mfma.)

// ---- wgrad_thin ----------------------------------------------------------------------------
template <int C, typename TB = float>
__global__ __launch_bounds__(256) void k_wgrad_thin(const TB* __restrict__ big, const float* __restrict__ small,
                                                    float* __restrict__ ws, int N, int n_units) {
  constexpr int NT = (16 * C + 31) / 32;   // N-tiles of 32 (cb,tap) columns
  __shared__ __attribute__((aligned(16))) float bt[C * TB_PLANE];
  __shared__ __attribute__((aligned(16))) float sp[8192];  // 128 px x 32 ch tile; reused (8192 floats) for the cross-wave reduction
  __shared__ UnitLut lut_s[1];
  const UnitLut* lut = lut_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  if (sizeof(TB) == 1) { lut_s[0].init(tid, 256); __syncthreads(); }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float sumS = 0.f, sumB[NT];
  int boff[NT];
  bool bval[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    sumB[t] = 0.f;
    const int nidx = t * 32 + i;
    bval[t] = nidx < 16 * C;
    const int cb = bval[t] ? (nidx >> 4) : 0, kh = (nidx >> 2) & 3, kw = nidx & 3;
    boff[t] = cb * TB_PLANE + kh * TB_ROW + (kw & 1) * TB_PAR + (kw >> 1);
  }
  BigThinRegs<C, TB> pfb;
  f32x4 pfs[4];
  auto load_unit = [&](int u) {
    const int n = u >> 3, sy0 = (u & 7) * 4;
    load_big_thin<C, TB>(pfb, big, n, sy0, true, tid);
    const float* src = small + ((((long)n * 32 + sy0) * 32)) * 32;  // 128 pixels x 32 ch contiguous
#pragma unroll
    for (int k = 0; k < 4; ++k) pfs[k] = *reinterpret_cast<const f32x4*>(src + (tid + k * 256) * 4);
  };
  int unit = blockIdx.x;
  if (unit < n_units) load_unit(unit);
  for (; unit < n_units; unit += gridDim.x) {
    __syncthreads();
    store_big_thin<C, TB>(pfb, bt, tid, lut);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(sp + (tid + k * 256) * 4) = pfs[k];
    __syncthreads();
    if (unit + (int)gridDim.x < n_units) load_unit(unit + gridDim.x);
    // wave wv handles small row sy_l = wv (32 pixels = 16 k-steps)
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
      const int sx = 2 * t + h;
      const float a = sp[(wv * 32 + sx) * 32 + i];
      sumS += a;
      const int base = (2 * wv) * TB_ROW + sx;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float b = bt[base + boff[nt]];
        b = bval[nt] ? b : 0.f;
        sumB[nt] += b;
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
      }
    }
  }
  // cross-wave reduction through LDS (reuse sp: 4 waves x NT x 16 x 64 floats <= 8192)
  __syncthreads();
  float* red = sp;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[((wv * NT + nt) * 16 + e) * 64 + lane] = acc[nt][e];
  __syncthreads();
  float* wsw = ws + (long)blockIdx.x * (NT * 1024 + 32 + NT * 32);
  for (int idx = tid; idx < NT * 1024; idx += 256) {
    const int nt = idx >> 10, e = (idx >> 6) & 15, ln = idx & 63;
    float v = red[((0 * NT + nt) * 16 + e) * 64 + ln] + red[((1 * NT + nt) * 16 + e) * 64 + ln] +
              red[((2 * NT + nt) * 16 + e) * 64 + ln] + red[((3 * NT + nt) * 16 + e) * 64 + ln];
    const int cs = (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5);
    const int nidx = nt * 32 + (ln & 31);
    wsw[nt * 1024 + cs * 32 + (ln & 31)] = v;   // [nt][cs][j]
    (void)nidx;
  }
  __syncthreads();
  // bias partial sums: sumS per cs (h halves + 4 waves), sumB per (cb,tap) column
  sumS += __shfl_xor(sumS, 32, 64);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) sumB[nt] += __shfl_xor(sumB[nt], 32, 64);
  float* redb = sp;  // [wv][1+NT][32]
  if (h == 0) {
    redb[(wv * (1 + NT)) * 32 + i] = sumS;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) redb[(wv * (1 + NT) + 1 + nt) * 32 + i] = sumB[nt];
  }
  __syncthreads();
  if (tid < (1 + NT) * 32) {
    float v = redb[tid] + redb[(1 + NT) * 32 + tid] + redb[2 * (1 + NT) * 32 + tid] + redb[3 * (1 + NT) * 32 + tid];
    wsw[NT * 1024 + tid] = v;
  }
}

template <int C>
__global__ __launch_bounds__(256) void k_wgrad_thin_reduce(const float* __restrict__ ws, float* __restrict__ dw,
                                                           float* __restrict__ db, int bias_from_big, int nblk) {
  wgrad_thin_reduce_body<C>(blockIdx.x, ws, dw, db, bias_from_big, nblk);
}

size_t wgrad_thin_ws_floats() { return (size_t)WT_MAX_BLOCKS * (2 * 1024 + 32 + 2 * 32); }

// ---- launchers -------------------------------------------------------------------------------
static bool thin_applicable(const ConvArgs& a) {
  return (a.Cb == 1 || a.Cb == 3) && a.Cs == 32 && a.Hs == 32 && a.Ws == 32;
}

int launch_down_thin(const ConvArgs& a, hipStream_t s) {
  if (!thin_applicable(a) || a.big_layout != DVAE_NCHW || a.out_layout != DVAE_NHWC) return 1;
  if (a.act != DVAE_ACT_NONE && a.act != DVAE_ACT_RELU) return 1;
  if (launch_down_thin_ws(a, s) == 0) return 0;         // large batches: the wave-specialised kernel (conv_thin_ws.hip)
  const int n_units = a.N * 8;
  const int grid = n_units < 1536 ? n_units : 1536;     // 6 resident workgroups per CU
  if ((a.mask_bits && (a.mask || a.out_bits)) || (a.out_bits && a.mask)) return 1;
  uint32_t* bits = a.mask_bits ? const_cast<uint32_t*>(a.mask_bits) : a.out_bits;
#define DVAE_DT(C, MODE) hipLaunchKernelGGL((k_down_thin<C, MODE>), dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units, bits)
  if (a.Cb == 1) {
    if (a.mask_bits) DVAE_DT(1, 2); else if (a.out_bits) DVAE_DT(1, 3); else if (a.mask) DVAE_DT(1, 1); else DVAE_DT(1, 0);
  } else {
    if (a.mask_bits) DVAE_DT(3, 2); else if (a.out_bits) DVAE_DT(3, 3); else if (a.mask) DVAE_DT(3, 1); else DVAE_DT(3, 0);
  }
#undef DVAE_DT
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_thin(const ConvArgs& a, hipStream_t s) {
  if (!thin_applicable(a) || a.small_layout != DVAE_NHWC || a.out_layout != DVAE_NCHW || a.mask) return 1;
  const int n_units = a.N * 8;
  const int grid = n_units;          // one unit per workgroup: the hardware dispatcher balances the load
  if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin<1, false>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, a.act, n_units, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
  else hipLaunchKernelGGL((k_up_thin<3, false>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, a.act, n_units, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_thin_recon(const ConvArgs& a, const float* target, float* g, int dist, const float* coef,
                         float* partials, hipStream_t s) {
  if (!thin_applicable(a) || a.small_layout != DVAE_NHWC || a.out_layout != DVAE_NCHW || a.mask) return 1;
  const int n_units = a.N * 8;
  // persistent (one loss partial per workgroup): 6 workgroups of 128 threads fit a CU (26 KB LDS each)
  const int grid = n_units < 1536 ? n_units : 1536;
  if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin<1, true>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  else hipLaunchKernelGGL((k_up_thin<3, true>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  DVAE_CHECK_LAUNCH();
  return 0;
}

// ---- uint8 input image (conv1 forward, conv1 weight gradient, fused likelihood target) ----------
int launch_down_thin_u8(const uint8_t* x, const float* w, const float* bias, float* out, uint32_t* out_bits, int N, int C, int act,
                        hipStream_t s) {
  const int n_units = N * 8;
  const int grid = n_units < 1536 ? n_units : 1536;
#define DVAE_DT8(C, MODE) hipLaunchKernelGGL((k_down_thin<C, MODE, uint8_t>), dim3(grid), dim3(256), 0, s, x, w, bias, (const float*)nullptr, out, N, act, n_units, out_bits)
  if (C == 1) { if (out_bits) DVAE_DT8(1, 3); else DVAE_DT8(1, 0); }
  else if (C == 3) { if (out_bits) DVAE_DT8(3, 3); else DVAE_DT8(3, 0); }
  else return 1;
#undef DVAE_DT8
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_thin_recon_u8(const ConvArgs& a, const uint8_t* target, float* g, int dist, const float* coef,
                            float* partials, hipStream_t s) {
  if (!thin_applicable(a) || a.small_layout != DVAE_NHWC || a.out_layout != DVAE_NCHW || a.mask) return 1;
  const int n_units = a.N * 8;
  const int grid = n_units < 1536 ? n_units : 1536;
  if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin<1, true, uint8_t>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  else hipLaunchKernelGGL((k_up_thin<3, true, uint8_t>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_wgrad_thin_u8(const uint8_t* x, const float* small, float* dw, float* db, int N, int C, float* ws, hipStream_t s) {
  const int n_units = N * 8;
  const int grid = n_units < WT_MAX_BLOCKS ? n_units : WT_MAX_BLOCKS;
  if (C == 1) {
    hipLaunchKernelGGL((k_wgrad_thin<1, uint8_t>), dim3(grid), dim3(256), 0, s, x, small, ws, N, n_units);
    DVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_wgrad_thin_reduce<1>, dim3(WT_REDUCE_BLOCKS(1)), dim3(256), 0, s, ws, dw, db, 0, grid);
  } else if (C == 3) {
    hipLaunchKernelGGL((k_wgrad_thin<3, uint8_t>), dim3(grid), dim3(256), 0, s, x, small, ws, N, n_units);
    DVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_wgrad_thin_reduce<3>, dim3(WT_REDUCE_BLOCKS(3)), dim3(256), 0, s, ws, dw, db, 0, grid);
  } else {
    return 1;
  }
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_wgrad_thin(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N, int Cb,
                      int Hs, float* ws, hipStream_t s) {
  if (!(Cb == 1 || Cb == 3) || Hs != 32) return 1;
  {                                                     // large batches: the wave-specialised kernel (conv_thin_ws.hip)
    int wgrid = 0;
    if (launch_wgrad_thin_ws(big, small, ws, bias_from_big, N, Cb, &wgrid, s) == 0) {
      if (Cb == 1) hipLaunchKernelGGL(k_wgrad_thin_reduce<1>, dim3(WT_REDUCE_BLOCKS(1)), dim3(256), 0, s, ws, dw, db, bias_from_big, wgrid);
      else hipLaunchKernelGGL(k_wgrad_thin_reduce<3>, dim3(WT_REDUCE_BLOCKS(3)), dim3(256), 0, s, ws, dw, db, bias_from_big, wgrid);
      DVAE_CHECK_LAUNCH();
      return 0;
    }
  }
  const int n_units = N * 8;
  int grid = n_units < WT_MAX_BLOCKS ? n_units : WT_MAX_BLOCKS;
  {
    static const int cap = env_int("DVAE_WGRAD_THIN_GRID", WT_MAX_BLOCKS);   // debug builds: A/B of the persistent grid size
    if (cap > 0 && cap < grid) grid = cap;
  }
  if (Cb == 1) {
    hipLaunchKernelGGL(k_wgrad_thin<1>, dim3(grid), dim3(256), 0, s, big, small, ws, N, n_units);
    DVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_wgrad_thin_reduce<1>, dim3(WT_REDUCE_BLOCKS(1)), dim3(256), 0, s, ws, dw, db, bias_from_big, grid);
  } else {
    hipLaunchKernelGGL(k_wgrad_thin<3>, dim3(grid), dim3(256), 0, s, big, small, ws, N, n_units);
    DVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_wgrad_thin_reduce<3>, dim3(WT_REDUCE_BLOCKS(3)), dim3(256), 0, s, ws, dw, db, bias_from_big, grid);
  }
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
