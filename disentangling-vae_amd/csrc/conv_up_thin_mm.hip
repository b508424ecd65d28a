// convT3 forward (decoders.py:65,82: ConvTranspose2d(32, 3, k4, s2, p1) + sigmoid) with the reconstruction likelihood and
// dL/dlogit fused (losses.py:429-444), on the matrix cores -- round 5, 64x64x3 images (fp32 or uint8 targets).
//
// Why the VALU kernel (k_up_thin_pk, conv_thin.hip: 88-91 us at 1024 images, 0.39 of the HBM roof, not bandwidth-bound) could
// not move here before: seen per output-parity class the layer is a contraction of 128 with THREE output columns -- a 16-wide
// MFMA tile is 81 % padding (round 3 measured the 4x4x1 form at half rate: slower than the VALU).  The product below is the
// same layer seen per 2 x 2 INPUT window instead: a transposed convolution with k4 s2 p1 sends the window of small pixels
// (qy + a, qx + b), a, b in {0, 1}, to exactly the 2 x 2 block of big pixels (2 qy + 1 + dy, 2 qx + 1 + dx), dy, dx in {0, 1},
// through the taps kh = 2 - 2a + dy, kw = 2 - 2b + dx -- every tap of the kernel exactly once.  So
//     D[(c, dy, dx)][q] = sum_{a, b, cs} W'[(c, dy, dx)][(a, b, cs)] * in[q + (a, b)][cs]
// is ONE dense product per base position q in [-1, 31]^2: M = 12 output values (padded to 16), K = 4 x 32 = 128, N = base
// positions; 33 x 33 of them per image instead of 32 x 32 (+6 %), 12 of 16 rows used: 29 us of v_mfma_f32_16x16x4_f32 at 1024
// images against 44 us of packed FMAs + their operand traffic.
//
// Workgroup: 512 threads = 8 waves, two per CU, persistent over units; a unit = 11 base rows x 33 base columns of one image (3
// units per image) = 363 positions = 23 tiles of 16, three per wave.  Per unit:
//   * the 12 x 32 input pixels (NHWC, 48 KB) go through registers into a swizzled LDS tile with a zero column on either side
//     (rows outside the image are stored as zeros);
//   * B operand (positions): a lane = (position j = lane % 16, channel group k = lane / 16) reads channels 8k .. 8k+7 of its
//     window pixel with two ds_read_b128 per tap -> 8 MFMAs; A operand (weights): 32 registers per lane for the whole
//     kernel, from the operand image dvae_stage_weights writes behind the pair records;
//   * D fragment: lane (j, g) holds the 2 x 2 block of channel g at position j -> + bias -> LDS stage [channel][22 rows][64];
//   * second phase, all threads, 16 bytes per access and every lane busy: stage + target -> sigmoid (hardware exp / rcp),
//     likelihood term, dL/dlogit -> recon and g_logit rows written whole.
// Every output is a fixed-order sum (tap-major, 8 contraction steps per tap, 4 channels per step inside the MFMA); results
// differ from k_up_thin_pk's in summation order only (tests: rtol 1e-5 of the layer scale against fp64).
#include "common.h"

namespace dvae {

#define UTM_ROWS 12                 // input rows of a unit's tile
#define UTM_COLS 34                 // 32 pixels + a zero column on either side
#define UTM_TILE (UTM_ROWS * UTM_COLS * 32)
#define UTM_BR 11                   // base rows per unit
#define UTM_NPOS (UTM_BR * 33)      // 363 base positions per unit
#define UTM_OR 22                   // output rows per unit (before clipping to the image)
#define UTM_STAGE (3 * UTM_OR * 64)

__device__ __forceinline__ float sigmoid_hw_mm(float v) { return sigmoid_aten(v); }

// experiment switches of variant builds (tools/build_variant.sh; the shipped library is built with the defaults below)
#ifndef UTM_VARIANT      // 1: next tile requested before the matrix phase, 2: targets requested before the matrix phase,
#define UTM_VARIANT 13   // 4: LDS-only barriers (global loads / stores stay in flight across them)
#endif
#ifndef UTM_ABL          // timing ablations (results invalid): 1 no MFMAs, 2 no likelihood arithmetic, 4 no target loads,
#define UTM_ABL 0        // 8 no output stores, 16 no tile loads
#endif
__device__ __forceinline__ void utm_barrier() {
  if (UTM_VARIANT & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// DIST: the reconstruction distribution (FUSE) as a compile-time constant -- with a run-time code the compiler evaluates all
// three likelihoods per output and selects (measured: 1360 vector instructions per unit and wave instead of ~500).
// TT: target type, float or uint8_t pixels (ToTensor's x / 255 on the fly, as the other image-reading kernels do).
// Two workgroups per CU (69 KB of LDS, <= 128 registers): one streams its tile in / its rows out while the other multiplies;
// a first version with ONE 121 KB workgroup per CU, a double-buffered tile and the next unit prefetched into registers ran all
// eight waves through every phase in lock step -- 57 / 83 us without / with the likelihood at 1024 images, no better than the
// packed-FMA kernel (measured in round 5; that visit's raw file did not survive a replaced build container).
template <bool FUSE, int DIST, typename TT>
__global__ __launch_bounds__(512, 4) void k_up_thin_mm(const float* __restrict__ small, const float* __restrict__ wimg,
                                                       const float* __restrict__ bias, float* __restrict__ out, int N,
                                                       int n_units, const TT* __restrict__ target, float* __restrict__ g,
                                                       const float* __restrict__ coef, float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tin = smem;                             // [UTM_TILE]
  float* stage = smem + UTM_TILE;                // [3][22][64]
  float* redl = stage + UTM_STAGE;               // [8]
  float* wl = redl + 8;                          // [32][64]: the A-operand image (register q of lane l at q * 64 + l)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kg = lane >> 4;
  const float gs = FUSE ? coef[DVAE_C_INV_B] : 0.f;
  float lsum = 0.f;

  // zero columns of the tile (never written again) -- and everything else once, so that no read ever sees garbage
  for (int e = tid; e < UTM_TILE / 4; e += 512) reinterpret_cast<f32x4*>(tin)[e] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A operand: W'[(c, dy, dx)][(tap, 8 kg + i)] for MFMA (tap, i), this lane's row m = lane % 16 and k-slot kg.  Round 6: the
  // image lives in LDS (8 KB) and a wave reads a tap's 8 registers per unit (8 conflict-free ds_read_b32 per 24 MFMAs)
  // instead of holding all 32 for the whole kernel: those registers now carry the NEXT unit's input tile through the second
  // phase (below)
  for (int e = tid; e < 32 * 64 / 4; e += 512) reinterpret_cast<f32x4*>(wl)[e] = reinterpret_cast<const f32x4*>(wimg)[e];
  const float bv = (kg < 3 && bias) ? bias[kg] : 0.f;

  // ---- everything that does not depend on the unit, once -----------------------------------------------------------------
  // this wave's three tiles: the window's first pixel (floats inside the tile; the taps are fixed strides from it) with the
  // swizzle keys of its two columns, and the stage offset
  int xb[3], soff[3];
  int key0[3];
  bool x0ok[3], x1ok[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int pos0 = 16 * (3 * wv + s) + j;
    const bool pvalid = pos0 < UTM_NPOS;
    const int pos = pvalid ? pos0 : UTM_NPOS - 1;
    const int rq = pos / 33, cq = pos - 33 * rq;
    xb[s] = (rq * UTM_COLS + cq) * 32;
    key0[s] = cq & 7;
    soff[s] = (kg * UTM_OR + 2 * rq) * 64 + 2 * cq - 1;
    x0ok[s] = pvalid && kg < 3 && cq > 0;                 // X = 2 cq - 1 >= 0
    x1ok[s] = pvalid && kg < 3 && cq < 32;                // X = 2 cq < 64
  }
  const bool tile3 = 16 * (3 * wv + 2) < UTM_NPOS;        // wave-uniform: the last wave's third tile is all padding
  // input tile: 12 rows x 256 chunks of 16 bytes, 6 per thread: thread (row half, chunk) -> rows ri = 2 k + (tid >> 8)
  const int within = tid & 255, rhalf = tid >> 8;
  const int pgo = rhalf * 1024 + within * 4;              // + 2048 k
  const int ci_ld = (within >> 3) + 1;
  const int plo = (rhalf * UTM_COLS + ci_ld) * 32 + (((within & 7) ^ (ci_ld & 7)) << 2);     // + 2 * UTM_COLS * 32 k
  // second phase: chunk e = tid + 512 k of the stage -> (channel, row, 16-byte column)
  int bso[3], bgo[3], brow[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = tid + 512 * k;
    const int c = e / (UTM_OR * 16), rem = e - c * (UTM_OR * 16), r = rem >> 4, x4 = rem & 15;
    brow[k] = e < 3 * UTM_OR * 16 ? r : -1000;            // (never a valid row)
    bso[k] = (c * UTM_OR + r) * 64 + 4 * x4;
    bgo[k] = (c * 64 + r) * 64 + 4 * x4;
  }
  __syncthreads();                                        // the zero fill is complete

  // Round 6: the NEXT unit's input tile is requested while this unit's second phase runs (registers pf: the 24 the weights
  // used to occupy).  Alone the kernel is no faster for it (75.2 us at 1024 images against 72.5-73.2 with the weights in
  // registers and the tile requested where it is stored; requesting the targets early as well: 96 us, the registers spill) --
  // but the STEP is: 1.066 against 1.072-1.078 ms on the same box, three alternations (profiles/r06_v8_ab3.txt,
  // r06_v9_ab4.txt): beside the estimator's backward kernels on the other stream, what counts is that the launch does not
  // stand still on a round trip.
  f32x4 pf[6];
  auto load_tile = [&](int unit) {
    const int n = unit / 3, u3 = unit - 3 * n;
    const int sy0 = UTM_BR * u3 - 1;
    const float* base = small + ((long)n * 32 + sy0) * 1024 + pgo;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int sy = sy0 + 2 * k + rhalf;
      pf[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sy >= 0 && sy < 32 && !((UTM_ABL & 16) && N > 0)) pf[k] = *reinterpret_cast<const f32x4*>(base + 2048 * k);
    }
  };
  if ((int)blockIdx.x < n_units) load_tile(blockIdx.x);
  for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    const int n = unit / 3, u3 = unit - 3 * n;
    const int Y0 = UTM_OR * u3 - 1;                       // image row of stage row 0
    const long img0 = ((long)n * 3 * 64 + Y0) * 64;       // (element offset of stage row 0 of channel 0; may be negative)
    // ---- targets of the second phase: requested now
    f32x4 tg[3];
    uchar4 tg8[3];
    auto load_targets_k = [&](int k) {
      const int Y = Y0 + brow[k];
      tg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      tg8[k] = make_uchar4(0, 0, 0, 0);
      if (Y >= 0 && Y < 64) {
        if constexpr (sizeof(TT) == 4) tg[k] = *reinterpret_cast<const f32x4*>(target + img0 + bgo[k]);
        else tg8[k] = *reinterpret_cast<const uchar4*>(target + img0 + bgo[k]);
      }
    };
    auto load_targets = [&]() {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int Y = Y0 + brow[k];
        tg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        tg8[k] = make_uchar4(0, 0, 0, 0);
        if (Y >= 0 && Y < 64 && !((UTM_ABL & 4) && N > 0)) {
          if constexpr (sizeof(TT) == 4) tg[k] = *reinterpret_cast<const f32x4*>(target + img0 + bgo[k]);
          else tg8[k] = *reinterpret_cast<const uchar4*>(target + img0 + bgo[k]);
        }
      }
    };
    // ---- tile in: registers -> swizzled LDS (rows outside the image: zeros)
#pragma unroll
    for (int k = 0; k < 6; ++k) *reinterpret_cast<f32x4*>(tin + plo + 2 * UTM_COLS * 32 * k) = pf[k];
    if ((UTM_VARIANT & 1) && unit + (int)gridDim.x < n_units) load_tile(unit + gridDim.x);
    if (FUSE && (UTM_VARIANT & 2)) load_targets();
    if (FUSE && (UTM_VARIANT & 16)) load_targets_k(0);
    utm_barrier();                                        // the tile is complete; the stage is free (second phase of the last unit)
    // ---- matrix phase
    f32x4 acc[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int a = tap >> 1, b = tap & 1;
      f32x4 x[3][2];
      float wr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) wr[i] = wl[(tap * 8 + i) * 64 + lane];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (s == 2 && !tile3) continue;
        const int key = (key0[s] + b) & 7;
        const float* px = tin + xb[s] + (a * UTM_COLS + b) * 32;
        x[s][0] = *reinterpret_cast<const f32x4*>(px + (((2 * kg) ^ key) << 2));
        x[s][1] = *reinterpret_cast<const f32x4*>(px + (((2 * kg + 1) ^ key) << 2));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (s == 2 && !tile3) continue;
          if (!((UTM_ABL & 1) && N > 0)) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[i], x[s][i >> 2][i & 3], acc[s], 0, 0, 0);
        }
      }
    }
    if (FUSE && (UTM_VARIANT & 8)) load_targets();        // (variant: requested behind the last MFMA, in front of the stage hand-over)
    if (FUSE && (UTM_VARIANT & 16)) { load_targets_k(1); load_targets_k(2); }
    // ---- D fragment -> stage: lane (j, kg < 3) holds the 2 x 2 block of channel kg at its position
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s == 2 && !tile3) continue;
      float* sp = stage + soff[s];
      if (x0ok[s]) { sp[0] = acc[s][0] + bv; sp[64] = acc[s][2] + bv; }
      if (x1ok[s]) { sp[1] = acc[s][1] + bv; sp[65] = acc[s][3] + bv; }
    }
    utm_barrier();                                        // the stage is complete; the tile may be overwritten
    // ---- the next unit's tile: in flight during the second phase
    if (!(UTM_VARIANT & 1) && unit + (int)gridDim.x < n_units) load_tile(unit + gridDim.x);
    // ---- second phase: rows of the stage, 16 bytes per access, every lane busy
    if (FUSE && !(UTM_VARIANT & (2 | 8 | 16))) load_targets();
    if (FUSE && sizeof(TT) != 4) {
#pragma unroll
      for (int k = 0; k < 3; ++k)     // ToTensor
        tg[k] = f32x4{(float)tg8[k].x / 255.0f, (float)tg8[k].y / 255.0f, (float)tg8[k].z / 255.0f, (float)tg8[k].w / 255.0f};
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int Y = Y0 + brow[k];
      if (Y < 0 || Y >= 64) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(stage + bso[k]);
      f32x4 gl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((UTM_ABL & 2) && N > 0) {
          gl[q] = v[q] + tg[k][q];
        } else if (FUSE && DIST == DVAE_REC_BERNOULLI) {
          float y, glq;
          lsum += sigmoid_bce_logit(v[q], tg[k][q], &y, &glq);
          v[q] = y;
          gl[q] = gs * glq;
        } else {
          const float y = sigmoid_hw_mm(v[q]);
          v[q] = y;
          if (FUSE) {
            float glq, gr;
            lsum += recon_elem(y, tg[k][q], DIST, &glq, &gr);
            gl[q] = gs * glq;
          }
        }
      }
      if ((UTM_ABL & 8) && N > 0) { lsum += v[0] + v[1] + v[2] + v[3] + gl[0] + gl[1] + gl[2] + gl[3]; continue; }
      *reinterpret_cast<f32x4*>(out + img0 + bgo[k]) = v;
      if (FUSE) *reinterpret_cast<f32x4*>(g + img0 + bgo[k]) = gl;
    }
  }
  if (FUSE) {
    const float v = wave_sum(lsum);
    __syncthreads();
    if (lane == 0) redl[wv] = v;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += redl[w];
      partials[blockIdx.x] = s;
    }
    // unused partial slots must read as zero
    for (int k = gridDim.x + blockIdx.x * 512 + tid; k < DVAE_REC_NPART; k += gridDim.x * 512) partials[k] = 0.f;
  }
}

// returns 1 if the shape is not covered (the caller takes k_up_thin_pk)
int launch_up_thin_mm(const float* small, const float* wimg, const float* bias, const void* target, int target_u8, float* out,
                      float* g, int dist, const float* coef, float* partials, int N, int act, hipStream_t s) {
  static const bool off = env_off("DVAE_UP_THIN_MM");     // A/B switch, debug builds only
  static const int min_n = env_int("DVAE_UP_THIN_MM_MIN_N", 1);
  if (off || N < min_n || act != DVAE_ACT_SIGMOID) return 1;
  if ((((uintptr_t)small | (uintptr_t)out | (uintptr_t)g) & 15) != 0 || ((uintptr_t)target & (target_u8 ? 3 : 15)) != 0) return 1;
  const int n_units = 3 * N;
  static const int cap = env_int("DVAE_UP_THIN_MM_GRID", 512);         // two workgroups per CU
  const int grid = n_units < cap ? n_units : cap;
  constexpr size_t lds = (size_t)(UTM_TILE + UTM_STAGE + 8 + 32 * 64) * sizeof(float);
#define UTM_GO(FUSE_, DIST_, TT_)                                                                                         \
  do {                                                                                                                    \
    static DeviceOnce attr;                                                                                               \
    if (attr.first())                                                                                                     \
      (void)hipFuncSetAttribute((const void*)k_up_thin_mm<FUSE_, DIST_, TT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_up_thin_mm<FUSE_, DIST_, TT_>), dim3(grid), dim3(512), lds, s, small, wimg, bias, out, N, n_units,      \
                       (const TT_*)target, g, coef, partials);                                                           \
  } while (0)
  if (!target) UTM_GO(false, 0, float);
  else if (target_u8) {
    if (dist == DVAE_REC_BERNOULLI) UTM_GO(true, DVAE_REC_BERNOULLI, uint8_t);
    else if (dist == DVAE_REC_GAUSSIAN) UTM_GO(true, DVAE_REC_GAUSSIAN, uint8_t);
    else UTM_GO(true, DVAE_REC_LAPLACE, uint8_t);
  } else {
    if (dist == DVAE_REC_BERNOULLI) UTM_GO(true, DVAE_REC_BERNOULLI, float);
    else if (dist == DVAE_REC_GAUSSIAN) UTM_GO(true, DVAE_REC_GAUSSIAN, float);
    else UTM_GO(true, DVAE_REC_LAPLACE, float);
  }
#undef UTM_GO
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
