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
template <int C, bool MASK, typename TB = float>
__global__ __launch_bounds__(256, 4) void k_down_thin(const TB* __restrict__ big, const float* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ mask,
                                                   float* __restrict__ out, int N, int act, int n_units) {
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
      const int sx = (e & 3) + 8 * (e >> 2) + 4 * h;
      float v = acc[e] + bv;
      if (act == DVAE_ACT_RELU) v = v > 0.f ? v : 0.f;
      if (MASK) v = mv[e] > 0.f ? v : 0.f;
      out[rowbase + sx * 32] = v;
    }
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
  for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
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

// ---- up_thin on the matrix core (round 3) ---------------------------------------------------------------------------------
// The same layer (convT3 forward: small NHWC [N,32,32,32] -> big NCHW [N,C,64,64], bias + sigmoid, optionally the fused
// reconstruction likelihood) on v_mfma_f32_4x4x1_16b_f32.  k_up_thin above is a VALU kernel (1536 scalar FMAs per small
// pixel: the C = 1 / 3 output channels are too narrow for a 32x32 or 16x16 MFMA tile) and is issue-bound at 0.35 of the
// HBM roof, with the likelihood arithmetic competing for the same VALU (profiles/r02_final_pmc_summary.md).  The 4x4x1 form
// has 16 independent 4x4 blocks: 4 PIXELS x 4 output channels (3 used) per block = 64 small pixels per instruction and
// contraction step, 75 % of the matrix core's rate for C = 3 -- and the VALU is left to the sigmoid / likelihood epilogue.
//   workgroup = 4 waves, persistent over units of 4 small rows x 32 columns (8 per image); wave (rp, py) owns the two small
//   rows 2rp, 2rp+1 (lane = pixel) and the output row parity py, both column parities px:
//     out[2r+py][2l+px][cb] = sum_{ty,tx,cs} small[r+py-ty][l+px-tx][cs] * w[cs][cb][1-py+2ty][1-px+2tx]
//   A operand: the lane's source pixel, four contracted channels per ds_read_b128 from the halo tile in LDS (6 distinct
//   source pixels per lane and channel quad serve 32 MFMAs); B operand: w for cb = lane % 4, the wave's 8 taps x 32
//   channels = 256 VGPRs per lane for the whole kernel (one wave per SIMD: the register file is this kernel's to use);
//   D: lane (block b, cb) holds 4 pixels x 2 column parities = 8 consecutive output columns of one row: two 16-byte stores.
// Exact fp32 (k-ordered fmaf chains, four accumulators per output -- contracted channel mod 4 -- added at the end).
#define UM_GRID 256                // persistent: one workgroup per CU (a wave keeps 256 VGPRs of weights)
#define UM_ROWS 6
#define UM_COLS 34
#define UM_TILE (UM_ROWS * UM_COLS * 32)
template <int C, bool FUSE, typename TT = float>
__global__ __launch_bounds__(256) void k_up_thin_mfma(const float* __restrict__ small, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int N, int act,
                                                      int n_units, const TT* __restrict__ target, float* __restrict__ g,
                                                      int dist, const float* __restrict__ coef, float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float st[2][UM_TILE];
  __shared__ float wsh[32 * C * 16];
  __shared__ float redl[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wv >> 1, py = wv & 1;
  const int cb = lane & 3;
  const float gs = FUSE ? coef[DVAE_C_INV_B] : 0.f;
  // ---- staging slots of this thread: 16-byte chunk `ch` of tile pixels (row, col), constant over the units
  constexpr int NSLOT = (UM_ROWS * UM_COLS * 8 + 255) / 256;
  int s_lds[NSLOT], s_gofs[NSLOT], s_row[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int sidx = tid + 256 * k;
    s_lds[k] = -1; s_gofs[k] = 0; s_row[k] = -100;
    if (sidx < UM_ROWS * UM_COLS * 8) {
      const int ch = sidx & 7, pix = sidx >> 3;
      const int col = pix % UM_COLS, row = pix / UM_COLS;
      s_lds[k] = (row * UM_COLS + col) * 32 + ((ch ^ (col & 7)) << 2);
      const bool inside = col >= 1 && col <= 32;
      s_gofs[k] = ((row - 1) * 32 + (col - 1)) * 32 + ch * 4;
      s_row[k] = inside ? row : -100;                    // -100: halo column, always zero
    }
  }
  f32x4 pf[NSLOT];
  auto load_tile = [&](int u) {
    const int n = u >> 3, sy0 = (u & 7) * 4;
    const float* base = small + (((long)n * 32 + sy0) * 32) * 32;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int sy = sy0 - 1 + s_row[k];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (s_row[k] >= 0 && sy >= 0 && sy < 32) v = *reinterpret_cast<const f32x4*>(base + s_gofs[k]);
      pf[k] = v;
    }
  };
  auto store_tile = [&](float* t) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k)
      if (s_lds[k] >= 0) *reinterpret_cast<f32x4*>(t + s_lds[k]) = pf[k];
  };
  int unit = blockIdx.x;
  if (unit < n_units) load_tile(unit);
  for (int e = tid; e < 32 * C * 16; e += 256) wsh[e] = w[e];
  __syncthreads();
  // B operands: Bw[ty][kw][cs] = w[cs][cb][1 - py + 2 ty][kw] for cb = lane % 4 (0 beyond C)
  float Bw[2][4][32];
#pragma unroll
  for (int ty = 0; ty < 2; ++ty)
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
      for (int cs = 0; cs < 32; ++cs)
        Bw[ty][kw][cs] = cb < C ? wsh[(cs * C + (cb < C ? cb : 0)) * 16 + (1 - py + 2 * ty) * 4 + kw] : 0.f;
  const float bv = (bias && cb < C) ? bias[cb] : 0.f;
  // A operand addresses: pixel = lane: small row 2rp + lane/32, column lane%32; source (row + py - ty, col + dc), dc in -1..1
  const int rr = 2 * rp + (lane >> 5), l = lane & 31;
  int aoff[2][3], asw[3];
#pragma unroll
  for (int dc = 0; dc < 3; ++dc) {
    const int tcol = l + dc;                             // (l + 1) + (dc - 1)
    asw[dc] = tcol & 7;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty) aoff[ty][dc] = ((rr + 1 + py - ty) * UM_COLS + tcol) * 32;
  }
  if (unit < n_units) store_tile(st[0]);
  __syncthreads();
  if (unit + (int)gridDim.x < n_units) load_tile(unit + gridDim.x);
  float lsum = 0.f;
  int buf = 0;
  for (; unit < n_units; unit += gridDim.x) {
    const float* t = st[buf];
    // lane (block b = lane/4, cb): pixels 4b .. 4b+3 of the wave's two rows x both column parities = output columns
    // 8 (b%8) .. +7 of row 2 (sy0 + 2rp + b/8) + py of channel cb
    const int b_ = lane >> 2;
    const long o = ((((long)(unit >> 3) * C + (cb < C ? cb : 0)) * 64) + 2 * ((unit & 7) * 4 + 2 * rp + (b_ >> 3)) + py) * 64 + 8 * (b_ & 7);
    // the likelihood target of this unit: requested now, consumed after the MFMA phase
    f32x4 tg0 = {0.f, 0.f, 0.f, 0.f}, tg1 = {0.f, 0.f, 0.f, 0.f};
    uint2 tgb = {0u, 0u};
    if (FUSE && cb < C) {
      if constexpr (sizeof(TT) == 4) {
        tg0 = *reinterpret_cast<const f32x4*>(target + o);
        tg1 = *reinterpret_cast<const f32x4*>(target + o + 4);
      } else {
        tgb = *reinterpret_cast<const uint2*>(target + o);
      }
    }
    f32x4 acc[2][4];                                     // [px][channel % 4]: an accumulator is re-used every 8th MFMA
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[px][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 8 steps of 4 contracted channels: 6 operand reads (2 source rows x 3 source columns) + 32 MFMAs each.  The reads of
    // step q+1 are issued before the MFMAs of step q (register ping-pong, order pinned by sched_group_barrier): with one
    // wave per SIMD nothing else hides the LDS latency
    f32x4 x[2][2][3];
    auto rd = [&](int q, int slot) {
#pragma unroll
      for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc)
          x[slot][ty][dc] = *reinterpret_cast<const f32x4*>(t + aoff[ty][dc] + ((q ^ asw[dc]) << 2));
    };
    rd(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cur = q & 1;
      if (q + 1 < 8) rd(q + 1, cur ^ 1);
#pragma unroll
      for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            const int kw = 1 - px + 2 * tx, dc = 1 + px - tx;          // source column l + px - tx
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[px][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[cur][ty][dc][j], Bw[ty][kw][4 * q + j], acc[px][j], 0, 0, 0);
          }
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);    // 6 DS reads (next step)
      __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);   // 32 MFMAs (this step)
    }
    // hand over: the next unit's tile (prefetched during the MFMAs) goes to the other buffer
    if (unit + (int)gridDim.x < n_units) store_tile(st[buf ^ 1]);
    __syncthreads();
    if (unit + 2 * (int)gridDim.x < n_units) load_tile(unit + 2 * gridDim.x);
    buf ^= 1;
    // epilogue
    if (cb < C) {
      const f32x4 s0 = (acc[0][0] + acc[0][1]) + (acc[0][2] + acc[0][3]), s1 = (acc[1][0] + acc[1][1]) + (acc[1][2] + acc[1][3]);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = s0[e] + bv; v[2 * e + 1] = s1[e] + bv; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (act == DVAE_ACT_SIGMOID) v[e] = 1.f / (1.f + expf(-v[e]));
        else if (act == DVAE_ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(out + o) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(out + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
      if (FUSE) {
        float xt[8];
        if constexpr (sizeof(TT) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { xt[e] = tg0[e]; xt[4 + e] = tg1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xt[e] = to_unit((uint8_t)((tgb.x >> (8 * e)) & 0xff));
            xt[4 + e] = to_unit((uint8_t)((tgb.y >> (8 * e)) & 0xff));
          }
        }
        float gl[8], gr;
#pragma unroll
        for (int e = 0; e < 8; ++e) { lsum += recon_elem(v[e], xt[e], dist, &gl[e], &gr); gl[e] *= gs; }
        *reinterpret_cast<f32x4*>(g + o) = f32x4{gl[0], gl[1], gl[2], gl[3]};
        *reinterpret_cast<f32x4*>(g + o + 4) = f32x4{gl[4], gl[5], gl[6], gl[7]};
      }
    }
  }
  if (FUSE) {
    const float v = wave_sum(lsum);
    if (lane == 0) redl[wv] = v;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = (redl[0] + redl[1]) + (redl[2] + redl[3]);
    // unused partial slots must read as zero
    for (int k = gridDim.x + blockIdx.x * 256 + tid; k < DVAE_REC_NPART; k += gridDim.x * 256) partials[k] = 0.f;
  }
}

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
  const int n_units = a.N * 8;
  const int grid = n_units < 1536 ? n_units : 1536;     // 6 resident workgroups per CU
  if (a.Cb == 1) {
    if (a.mask) hipLaunchKernelGGL((k_down_thin<1, true>), dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
    else hipLaunchKernelGGL((k_down_thin<1, false>), dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
  } else {
    if (a.mask) hipLaunchKernelGGL((k_down_thin<3, true>), dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
    else hipLaunchKernelGGL((k_down_thin<3, false>), dim3(grid), dim3(256), 0, s, a.big, a.w, a.bias, a.mask, a.out, a.N, a.act, n_units);
  }
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_thin(const ConvArgs& a, hipStream_t s) {
  if (!thin_applicable(a) || a.small_layout != DVAE_NHWC || a.out_layout != DVAE_NCHW || a.mask) return 1;
  const int n_units = a.N * 8;
  static const bool valu = env_on("DVAE_UP_THIN_VALU");     // debug builds: the round-1/2 VALU kernel (A/B)
  if (valu) {
    const int grid = n_units;        // one unit per workgroup: the hardware dispatcher balances the load
    if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin<1, false>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, a.act, n_units, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL((k_up_thin<3, false>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, a.act, n_units, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
  } else {
    const int grid = n_units < UM_GRID ? n_units : UM_GRID;
    if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin_mfma<1, false>), dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.out, a.N, a.act, n_units, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
    else hipLaunchKernelGGL((k_up_thin_mfma<3, false>), dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.out, a.N, a.act, n_units, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (float*)nullptr);
  }
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_thin_recon(const ConvArgs& a, const float* target, float* g, int dist, const float* coef,
                         float* partials, hipStream_t s) {
  if (!thin_applicable(a) || a.small_layout != DVAE_NHWC || a.out_layout != DVAE_NCHW || a.mask) return 1;
  const int n_units = a.N * 8;
  static const bool valu = env_on("DVAE_UP_THIN_VALU");     // debug builds: the round-1/2 VALU kernel (A/B)
  if (valu) {
    // persistent (one loss partial per workgroup): 6 workgroups of 128 threads fit a CU (26 KB LDS each)
    const int grid = n_units < 1536 ? n_units : 1536;
    if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin<1, true>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
    else hipLaunchKernelGGL((k_up_thin<3, true>), dim3(grid), dim3(128), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  } else {
    const int grid = n_units < UM_GRID ? n_units : UM_GRID;
    if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin_mfma<1, true>), dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
    else hipLaunchKernelGGL((k_up_thin_mfma<3, true>), dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  }
  DVAE_CHECK_LAUNCH();
  return 0;
}

// ---- uint8 input image (conv1 forward, conv1 weight gradient, fused likelihood target) ----------
int launch_down_thin_u8(const uint8_t* x, const float* w, const float* bias, float* out, int N, int C, int act, hipStream_t s) {
  const int n_units = N * 8;
  const int grid = n_units < 1536 ? n_units : 1536;
  if (C == 1) hipLaunchKernelGGL((k_down_thin<1, false, uint8_t>), dim3(grid), dim3(256), 0, s, x, w, bias, (const float*)nullptr, out, N, act, n_units);
  else if (C == 3) hipLaunchKernelGGL((k_down_thin<3, false, uint8_t>), dim3(grid), dim3(256), 0, s, x, w, bias, (const float*)nullptr, out, N, act, n_units);
  else return 1;
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_up_thin_recon_u8(const ConvArgs& a, const uint8_t* target, float* g, int dist, const float* coef,
                            float* partials, hipStream_t s) {
  if (!thin_applicable(a) || a.small_layout != DVAE_NHWC || a.out_layout != DVAE_NCHW || a.mask) return 1;
  const int n_units = a.N * 8;
  const int grid = n_units < UM_GRID ? n_units : UM_GRID;
  if (a.Cb == 1) hipLaunchKernelGGL((k_up_thin_mfma<1, true, uint8_t>), dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
  else hipLaunchKernelGGL((k_up_thin_mfma<3, true, uint8_t>), dim3(grid), dim3(256), 0, s, a.small, a.w, a.bias, a.out, a.N, DVAE_ACT_SIGMOID, n_units, target, g, dist, coef, partials);
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
