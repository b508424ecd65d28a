// The fully-connected core of the Burgess VAE as ONE launch per direction:
//   forward : lin1 -> lin2 -> mu_logvar_gen -> reparameterise (+ per-dim KL partials) -> lin1 -> lin2 -> lin3
//             (encoders.py:81-87, vae.py:52-71, losses.py:452-480, decoders.py:71-73)
//   backward: the mirror chain of input gradients with the ReLU masks of the saved activations and the reparameterisation /
//             KL backward in the middle (training.py:157 through the same lines).
// Why: as separate launches these are 7 + 7 kernels of 4-11 us each at ANY batch size (profiles/r02_final_kbench*.txt: the
// arithmetic is < 1 us, the rest is launch + dependent-load latency); they are the longest part of the latency chain that
// bounds the step at the 128 images per GPU of the 8-GPU headline configuration (profiles/r02_final_timeline_b128.md).
//
// Formulation.  Batch rows are independent, so a workgroup (4 waves, one per SIMD) owns FCC_R = 8 rows and walks the whole
// chain with the activations in LDS; nothing but the weights is read from memory between the first and the last layer, and
// the weight streams do not depend on the data, so they are prefetched across layer boundaries (8 chunks in flight per
// wave).  A layer is  y[8][O] = x[8][C] W  on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks per instruction =
// 4 batch rows x 64 output columns per wave, one contraction step; two row groups share the B operand.  The 4x4x1 form
// issues in 16 cycles (measured: profiles/r03_v2_fcc_ab.txt, r03_v4_kbench.txt) = 32 FLOP/clk/SIMD, half the rate of the
// 16x16x4 / 32x32x2 forms -- but with M = 4 nothing is wasted on 8 rows, where a 16x16x4 tile idles half of the matrix
// core (same net rate) and a 32x32x2 tile three quarters.  B operand: the lane's output column, four contraction steps per
// 16-byte load from the k-chunked images of dvae_stage_weights ([C/4][O][4]: a wave's load = 1 KB contiguous).  A operand:
// one broadcast ds_read_b128 per row group and 4 steps.  Bounds per workgroup and direction: 12.8 k MFMAs on 4 SIMDs =
// 51 k cycles = 21 us of matrix core, and 1.6 MB of weights at the ~65-90 GB/s one CU draws from L2 = 18-24 us: the launch
// takes 24-28 us at any batch size up to 2048 rows (256 workgroups), against 50 / 67 us for the 7 + 7 launches it replaces
// at 128 rows.  Fewer rows per workgroup would only move the bound from the matrix core to the weight stream.
// Exact fp32 (k-ordered fmaf chains per accumulator; two accumulators per output, even / odd steps, summed at the end).
//
// Round 6, CONV = true: the 4x4 end of the conv stacks inside the same launches.  The 4 (8) batch rows of a workgroup ARE whole
// images, and conv_64's output IS the chain's first input row (encoders.py:76-81), lin3's output IS convT_64's input
// (decoders.py:73-75) -- and the mirror of both in the backward pass.  As launches of their own those four layers (k_down32<4> /
// k_up32<4>, 0.5 GFLOP each) cost 8-12 us apiece at ANY batch size, 10-80 us inside a step where they queue behind the other
// stream's chip-filling kernels (profiles/r06_final1_timeline*.md).  Here a workgroup runs conv4_end.h's unit bodies (the
// arithmetic of those kernels, bit for bit) as prologue and epilogue of its chain: the staged 64 KB weight images arrive by
// LDS-DMA while the chain's first ring is in flight / while the last layers multiply, and the chain's own LDS buffers overlay
// the prologue's weight image once it is consumed (150 KB of LDS in all, one workgroup per CU as before).
#include "common.h"
#include "conv4_end.h"

namespace dvae {

#define FCC_XS 516        // LDS row stride of an activation tile (floats): rows r, r+1 are 4 banks apart -> the four distinct
                          // 16-byte addresses of a broadcast ds_read_b128 never share a bank
#define FCC_HID 256
#define FCC_FLAT 512

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  // A: lane l -> (block l/4, row l%4); B: lane l -> (block l/4, column l%4); D[v] on lane l = (block l/4, row v, column l%4)
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// accumulators of one 64-column group: [row group of 4 rows][even / odd contraction step]; RG row groups = 4 RG rows per workgroup
template <int RG>
struct AccR { f32x4 a[RG][2]; };
template <int RG>
__device__ __forceinline__ void acc_zero(AccR<RG>& c) {
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int p = 0; p < 2; ++p) c.a[g][p] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// the lane's column, rows 0 .. 4 RG - 1
template <int RG>
__device__ __forceinline__ void acc_rows(const AccR<RG>& c, float (&v)[4 * RG]) {
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const f32x4 s = c.a[g][0] + c.a[g][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[4 * g + r] = s[r];
  }
}

template <int DEPTH, int NG>
struct Ring { f32x4 v[DEPTH][NG]; };

// first DEPTH chunks of a weight stream: wp = image + (lane's column) * 4, chunk c of group g at wp + g*gstride + c*cstride
template <int DEPTH, int NG, int NCH>
__device__ __forceinline__ void ring_fill(Ring<DEPTH, NG>& ring, const float* __restrict__ wp, int gstride, int cstride) {
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int c = d < NCH ? d : NCH - 1;
#pragma unroll
    for (int g = 0; g < NG; ++g) ring.v[d][g] = *reinterpret_cast<const f32x4*>(wp + g * gstride + c * cstride);
  }
}

template <int RG>
__device__ __forceinline__ void mac_chunk(const f32x4 (&a)[RG], const f32x4 w, AccR<RG>& acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int g = 0; g < RG; ++g) acc.a[g][j & 1] = mfma4(a[g][j], w[j], acc.a[g][j & 1]);
  }
}

// NCH chunks (4 contraction steps each) of NG column groups.  `ring` holds chunks 0..DEPTH-1 on entry; while the last DEPTH
// chunks are consumed the first chunks of the NEXT layer's stream are requested into `nring` (NGN = 0: none), so a layer
// boundary costs no exposed weight latency.  xr = activation tile + (lane & 3) * FCC_XS (row group 1 is 4 rows further).
template <int DEPTH, int NCH, int NG, int NCHN, int NGN, int RG>
__device__ __forceinline__ void gemm_run(Ring<DEPTH, NG>& ring, const float* __restrict__ wp, int gstride, int cstride,
                                         const float* xr, AccR<RG> (&acc)[NG], Ring<DEPTH, (NGN > 0 ? NGN : 1)>& nring,
                                         const float* __restrict__ wn, int gstride_n, int cstride_n) {
  static_assert(NCH % DEPTH == 0, "chunk count must be a multiple of the ring depth");
  // A operands one chunk ahead (register ping-pong): the LDS latency hides under the 4 RG MFMAs of the current chunk.  The
  // read for chunk NCH lands in the tile row's padding (FCC_XS >= C + 4) and is never used.
  f32x4 an[RG], nn[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) an[g] = *reinterpret_cast<const f32x4*>(xr + 4 * g * FCC_XS);
  for (int c0 = 0; c0 < NCH - DEPTH; c0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int c = c0 + d;
      f32x4 w[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        w[g] = ring.v[d][g];
        ring.v[d][g] = *reinterpret_cast<const f32x4*>(wp + g * gstride + (c + DEPTH) * cstride);
      }
#pragma unroll
      for (int g = 0; g < RG; ++g) nn[g] = *reinterpret_cast<const f32x4*>(xr + 4 * g * FCC_XS + 4 * (c + 1));
#pragma unroll
      for (int g = 0; g < NG; ++g) mac_chunk<RG>(an, w[g], acc[g]);
#pragma unroll
      for (int g = 0; g < RG; ++g) an[g] = nn[g];
      // pin the software pipeline: the re-request of the slot just consumed stays HERE (DEPTH chunks ahead of its use) and
      // the next chunk's operand reads precede this chunk's MFMAs; left alone the scheduler gathers all DEPTH loads at the
      // end of the loop body, where their latency is fully exposed
      __builtin_amdgcn_sched_group_barrier(0x020, NG, 0);            // NG VMEM reads
      __builtin_amdgcn_sched_group_barrier(0x100, RG, 0);            // RG DS reads (next chunk)
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * RG * NG, 0);   // 4 RG MFMAs per column group (this chunk)
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int c = NCH - DEPTH + d;
    if (NGN > 0) {
      const int cn = d < NCHN ? d : NCHN - 1;
#pragma unroll
      for (int g = 0; g < NGN; ++g) nring.v[d][g] = *reinterpret_cast<const f32x4*>(wn + g * gstride_n + cn * cstride_n);
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) nn[g] = *reinterpret_cast<const f32x4*>(xr + 4 * g * FCC_XS + 4 * (c + 1));
#pragma unroll
    for (int g = 0; g < NG; ++g) mac_chunk<RG>(an, ring.v[d][g], acc[g]);
#pragma unroll
    for (int g = 0; g < RG; ++g) an[g] = nn[g];
    __builtin_amdgcn_sched_group_barrier(0x020, NGN, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, RG, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * RG * NG, 0);
  }
}

// layer with a SMALL contraction (C <= 32, run-time: the latent side): chunks 0..c4-1 of one column group, all requested at
// once (c4 <= 8)
template <int RG>
__device__ __forceinline__ void gemm_small_c(const float* __restrict__ wp, int cstride, int c4, const float* xr, AccR<RG>& acc) {
  f32x4 w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) w[c] = *reinterpret_cast<const f32x4*>(wp + (c < c4 ? c : 0) * cstride);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < c4) {
      f32x4 a[RG];
#pragma unroll
      for (int g = 0; g < RG; ++g) a[g] = *reinterpret_cast<const f32x4*>(xr + 4 * g * FCC_XS + 4 * c);
      mac_chunk<RG>(a, w[c], acc);
    }
  }
}

// R x C rows of a row-major [n][C] tensor -> LDS tile (zero rows beyond n)
template <int C, int NT, int R>
__device__ __forceinline__ void load_rows(const float* __restrict__ x, int row0, int n, float* tile, int tid) {
  constexpr int Q = C / 4;                      // 16-byte chunks per row
  constexpr int NP = R * Q / NT;
  static_assert(NP >= 1 && NP * NT == R * Q, "the tile must be a whole number of 16-byte chunks per thread");
  f32x4 v[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = tid + NT * p;
    const int r = idx / Q, q = idx % Q;
    v[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row0 + r < n) v[p] = *reinterpret_cast<const f32x4*>(x + (long)(row0 + r) * C + q * 4);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int idx = tid + NT * p;
    *reinterpret_cast<f32x4*>(tile + (idx / Q) * FCC_XS + (idx % Q) * 4) = v[p];
  }
}

// L2 warm-up of a weight image.  Inside a training step the FC images were written ~200 us (and several hundred MB of conv
// traffic) before this kernel runs: on boxes whose caches do not retain them the chain's dependent weight stream pays HBM
// latency per ring refill -- 45-64 us per launch instead of 24-26 (profiles/r03_v15_fcc_cold.txt; in the step:
// profiles/r03_final_timeline.md 61 / 69 us, profiles/r03_v4_timeline.md 27 / 37 us on another box).  Every workgroup
// therefore touches, at its start and all at once, every 64-byte sector of its share of the four large images with LDS-DMA
// transfers into a 1 KB scratch block (no destination registers, so nothing to keep live; the data is never used): one parallel HBM round trip,
// after which the stream runs from L2.  Workgroups b, b + 8, b + 16 .. share an XCD (and its L2): they split each image
// among themselves.  Vector-memory operations return in order, so the wave's real weight loads queue behind its prefetches.
__device__ __forceinline__ void l2_warm(const float* __restrict__ p, int nfloats, int share, int nshares, int wv, int nwaves,
                                        int lane, unsigned lds_byte_addr) {
  // one transfer touches 64 different 64-byte sectors (16 bytes of each): 4 KB of the image per instruction
  const int nblk = nfloats >> 10;                   // 4 KB blocks
  for (int b = share + nshares * wv; b < nblk; b += nshares * nwaves)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p + (long)b * 1024 + lane * 16),
                 "s"(lds_byte_addr)
                 : "memory", "m0");
}

struct FwdArgs { dvae_fc_chain_fwd_args a; };
struct BwdArgs { dvae_fc_chain_bwd_args a; };

// Work split of a workgroup.  KS = 1: 4 waves, wave w owns output columns 64w .. 64w+63 of a 256-wide layer over the whole
// contraction (both 256-column halves of a 512-wide one).  KS = 2: 8 waves (two per SIMD, twice the weight bytes in flight
// per CU -- the launch is bound by the latency of its L2 -> register weight stream, see the file header): wave (cg, kh) owns
// columns 64cg .. of contraction half kh, the two halves are added through LDS in a fixed order (kh = 0 first); of a 512-wide
// layer it owns the 256-column half kh over the whole contraction.  DEPTH = weight chunks in flight per wave.
template <int KS>
struct Lane {
  int tid, lane, wv, cg, kh, col, xo;
  __device__ __forceinline__ Lane() {
    tid = threadIdx.x; lane = tid & 63;
    wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    cg = wv & 3; kh = KS == 2 ? wv >> 2 : 0;
    col = cg * 64 + lane;
    xo = (lane & 3) * FCC_XS;
  }
};

// LDS plan (floats inside the dynamic allocation).  Plain chain: the buffers below, one after the other.  CONV: region A =
// [0, 16384) is the prologue's weight image and THEN the same buffers (the image is dead once the prologue's matrix phase is
// over); region B = [16384, +20992) is the prologue's big tile + cross-wave reduction buffer and THEN the epilogue's weight
// image + small tile; the L2 warm-up sink follows.
template <int KS, int RG, bool CONV>
struct ChainLds {
  static constexpr int R = 4 * RG, NWV = 4 * KS;
  static constexpr int TA = 0;
  static constexpr int PART = TA + R * FCC_XS;
  static constexpr int TB = PART + (KS == 2 ? R * FCC_HID : 4);
  static constexpr int RED = TB + R * FCC_XS;                  // [NWV][R][64]
  static constexpr int MLT = RED + NWV * R * 64;               // [R][64]
  static constexpr int END = MLT + R * 64;
  static constexpr int WL = 0;                                 // prologue: "down" image
  static constexpr int BT = C4_WL_FLOATS, RD = BT + C4_BT_FLOATS;           // prologue: big tile, reduction buffer
  static constexpr int WL2 = C4_WL_FLOATS, ST = WL2 + C4_WL_FLOATS;         // epilogue: "up" image, small tile
  static constexpr int SINK = CONV ? RD + C4_RED_FLOATS : END;
  static constexpr int TOTAL = SINK + 256;
  static_assert(!CONV || (END <= C4_WL_FLOATS && ST + C4_ST_FLOATS <= SINK), "the chain's buffers overlay the prologue's image");
};

// 256-wide layer, contraction of NCH chunks: complete sums in v[] of the waves with kh == 0 (one workgroup barrier inside
// when KS == 2).  wp = this layer's image + col * 4; tile = activation tile.  The next layer's first chunks are requested
// into nring from wn (already offset for this wave) while the last DEPTH chunks are consumed.
template <int DEPTH, int KS, int NCH, int NCHN, int NGN, int RG>
__device__ __forceinline__ void layer256(const Lane<KS>& L, Ring<DEPTH, 1>& ring, const float* __restrict__ wp, int cstride,
                                         const float* tile, Ring<DEPTH, (NGN > 0 ? NGN : 1)>& nring,
                                         const float* __restrict__ wn, int gstride_n, int cstride_n, float* part,
                                         float (&v)[4 * RG]) {
  constexpr int NW = NCH / KS;                  // chunks per wave
  AccR<RG> acc[1];
  acc_zero(acc[0]);
  gemm_run<DEPTH, NW, 1, NCHN, NGN, RG>(ring, wp + (long)L.kh * NW * cstride, 0, cstride, tile + L.xo + L.kh * NW * 4, acc, nring,
                                        wn, gstride_n, cstride_n);
  acc_rows(acc[0], v);
  if (KS == 2) {
    if (L.kh == 1) {
#pragma unroll
      for (int r = 0; r < 4 * RG; ++r) part[r * FCC_HID + L.col] = v[r];
    }
    __syncthreads();
    if (L.kh == 0) {
#pragma unroll
      for (int r = 0; r < 4 * RG; ++r) v[r] += part[r * FCC_HID + L.col];
    }
  }
}

// first chunks of a 256-wide layer's stream for this wave
template <int DEPTH, int KS, int NCH>
__device__ __forceinline__ void fill256(const Lane<KS>& L, Ring<DEPTH, 1>& ring, const float* __restrict__ wp, int cstride) {
  ring_fill<DEPTH, 1, NCH / KS>(ring, wp + (long)L.kh * (NCH / KS) * cstride, 0, cstride);
}

// ---------------------------------------------------------------------------------------------- forward
template <int DEPTH, int KS, int RG, bool CONV>
__global__ __launch_bounds__(256 * KS) void k_fc_chain_fwd(const FwdArgs P) {
  const dvae_fc_chain_fwd_args& a = P.a;
  constexpr int R = 4 * RG;                         // batch rows per workgroup
  constexpr int NT = 256 * KS, NWV = 4 * KS;
  constexpr int SD = 64 / NWV;                      // chunks per wave of a layer whose contraction is split over ALL waves
  constexpr int G512 = KS == 2 ? 1 : 2;             // 256-column halves of a 512-wide layer per wave
  using LD = ChainLds<KS, RG, CONV>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tA = smem + LD::TA;
  float* tB = smem + LD::TB;
  float* red = smem + LD::RED;                      // [NWV][R][64]
  float* mlt = smem + LD::MLT;                      // [R][64]
  float* part = smem + LD::PART;
  float* warm_sink = smem + LD::SINK;
  static_assert(!CONV || KS == 2, "the conv ends are 512-thread bodies");
  const Lane<KS> L;
  const int tid = L.tid, lane = L.lane, wv = L.wv, col = L.col, xo = L.xo;
  const bool own = L.kh == 0;                       // this wave finishes the 256-wide layers (bias, activation, stores)
  const int row0 = blockIdx.x * R;
  const int D = a.D, D2 = 2 * a.D;
  constexpr int CS = FCC_HID * 4;                   // chunk stride of a 256-wide image

  Ring<DEPTH, 1> r1, r2, dummy;
  Ring<SD, 1> dummy_s;
  fill256<DEPTH, KS, 128>(L, r1, a.w_e1 + col * 4, CS);
  {
    const int nsh = gridDim.x >= 8 ? gridDim.x >> 3 : 1, sh = (blockIdx.x >> 3) % nsh;
    const unsigned sink = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)warm_sink;
    l2_warm(a.w_e1, FCC_FLAT * FCC_HID, sh, nsh, wv, NWV, lane, sink);
    l2_warm(a.w_e2, FCC_HID * FCC_HID, sh, nsh, wv, NWV, lane, sink);
    l2_warm(a.w_d2, FCC_HID * FCC_HID, sh, nsh, wv, NWV, lane, sink);
    l2_warm(a.w_d3, FCC_HID * FCC_FLAT, sh, nsh, wv, NWV, lane, sink);
  }
  const bool dec = row0 < a.n_dec;                  // workgroup-uniform
  if constexpr (CONV) {
    // ---- conv_64 (encoders.py:76-80): unit u of this workgroup = images row0 + 4 u .. + 3 -> a_flat (global: the backward pass
    // reads it) and, with one unit per workgroup, straight into the chain's input tile
    SlotDesc<Geo<4>::BIG_NPF> sd;
    init_big_slots<4>(sd, tid);
    f32x4 pf[Geo<4>::BIG_NPF];
    load_big<4>(pf, sd, a.conv_in, blockIdx.x * RG, a.n_enc);
    c4_weight_image_issue(a.conv_w, smem + LD::WL, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the barrier inside the unit publishes the image)
    float* a_out = const_cast<float*>(a.a_flat);
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      if (u > 0) load_big<4>(pf, sd, a.conv_in, blockIdx.x * RG + u, a.n_enc);
      // (RG == 2: the tile would overlay the image the second unit still multiplies with -- its rows come back from a_flat below)
      c4_down_unit<false, RG == 1>(a.conv_b, nullptr, a_out, tA, FCC_XS, a.n_enc, DVAE_ACT_RELU, blockIdx.x * RG + u,
                                   smem + LD::WL, smem + LD::BT, smem + LD::RD, pf, sd);
    }
    if (RG != 1) {
      __syncthreads();
      load_rows<FCC_FLAT, NT, R>(a.a_flat, row0, a.n_enc, tA, tid);
    }
  } else {
    load_rows<FCC_FLAT, NT, R>(a.a_flat, row0, a.n_enc, tA, tid);
  }
  const float be1 = a.b_e1[col], be2 = a.b_e2[col];
  __syncthreads();
  if constexpr (CONV) {
    // convT_64's "up" image: region B is free now (the prologue's last reads were in front of that barrier)
    if (dec && a.convT_w) {
      c4_weight_image_issue(a.convT_w, smem + LD::WL2, tid);
      if (RG == 1)     // one unit per workgroup: lin3's outputs go straight into the epilogue's small tile; its halo is zero
        for (int e = tid; e < C4_ST_FLOATS / 4; e += NT) reinterpret_cast<f32x4*>(smem + LD::ST)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  float v[R];
  // ---- encoder lin1: 512 -> 256, ReLU
  layer256<DEPTH, KS, 128, 64 / KS, 1, RG>(L, r1, a.w_e1 + col * 4, CS, tA, r2, a.w_e2 + col * 4 + (long)L.kh * (64 / KS) * CS, 0, CS, part, v);
  if (own) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = fmaxf(v[r] + be1, 0.f);
      tB[r * FCC_XS + col] = v[r];
      if (row0 + r < a.n_enc) a.h1[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  __syncthreads();
  // ---- encoder lin2: 256 -> 256, ReLU; meanwhile request this wave's slice of mu_logvar_gen (contraction split over ALL waves)
  Ring<SD, 1> rml;
  const int cml = lane < D2 ? lane : D2 - 1;
  const float* wml = a.w_ml + ((long)wv * SD * D2 + cml) * 4;
  layer256<DEPTH, KS, 64, 0, 0, RG>(L, r2, a.w_e2 + col * 4, CS, tB, dummy, nullptr, 0, 0, part, v);
  ring_fill<SD, 1, SD>(rml, wml, 0, D2 * 4);
  if (own) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = fmaxf(v[r] + be2, 0.f);
      tA[r * FCC_XS + col] = v[r];
      if (row0 + r < a.n_enc) a.h2[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  __syncthreads();
  // ---- mu_logvar_gen: 256 -> 2D (no activation): wave w contracts k in [4 SD w, 4 SD (w+1)), partial sums through LDS
  {
    AccR<RG> acc[1];
    acc_zero(acc[0]);
    gemm_run<SD, SD, 1, 0, 0, RG>(rml, wml, 0, D2 * 4, tA + xo + wv * SD * 4, acc, dummy_s, nullptr, 0, 0);
    acc_rows(acc[0], v);
#pragma unroll
    for (int r = 0; r < R; ++r) red[((wv) * R + (r)) * 64 + (lane)] = v[r];
  }
  // the decoder's second weight stream does not depend on anything computed here: request it now
  Ring<DEPTH, 1> rd2;
  if (dec) fill256<DEPTH, KS, 64>(L, rd2, a.w_d2 + col * 4, CS);
  __syncthreads();
  for (int t = tid; t < R * D2; t += NT) {
    const int r = t / D2, j = t % D2;
    float m = red[((0) * R + (r)) * 64 + (j)];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m += red[((w) * R + (r)) * 64 + (j)];
    m += a.b_ml[j];
    mlt[(r) * 64 + (j)] = m;
    if (row0 + r < a.n_enc) a.ml[(long)(row0 + r) * D2 + j] = m;
  }
  __syncthreads();
  // ---- reparameterise (vae.py:66-68) + per-dim KL terms (losses.py:470); z -> tB (zero padded to a multiple of 4 columns)
  {
    float* klt = red;                     // [R][16]
    const int dp = (D + 3) & ~3;
    for (int t = tid; t < R * 16; t += NT) {
      const int r = t >> 4, d = t & 15;
      float kl = 0.f;
      if (d < dp) {
        float zz = 0.f;
        if (d < D) {
          const float m = mlt[(r) * 64 + (2 * d)], lv = mlt[(r) * 64 + (2 * d + 1)];
          zz = m;
          const long o = (long)(row0 + r) * D + d;
          if (row0 + r < a.n_enc) {
            if (a.eps) zz = m + expf(0.5f * lv) * a.eps[o];
            a.mu[o] = m; a.logvar[o] = lv; a.z[o] = zz;
            if (row0 + r < a.n_kl) kl = 0.5f * (-1.f - lv + m * m + expf(lv));
          }
        }
        tB[r * FCC_XS + d] = zz;
      }
      klt[r * 16 + d] = kl;
    }
    __syncthreads();
    if (a.kl_part && tid < 16) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) s += klt[r * 16 + tid];
      a.kl_part[(long)blockIdx.x * 16 + tid] = s;
    }
  }
  if (!dec) return;
  const float bd1 = a.b_d1[col], bd2 = a.b_d2[col];
  // ---- decoder lin1: D -> 256, ReLU
  if (own) {
    AccR<RG> acc;
    acc_zero(acc);
    gemm_small_c<RG>(a.w_d1 + col * 4, CS, (D + 3) >> 2, tB + xo, acc);
    acc_rows(acc, v);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = fmaxf(v[r] + bd1, 0.f);
      tA[r * FCC_XS + col] = v[r];
      if (row0 + r < a.n_dec) a.d1[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  __syncthreads();
  // ---- decoder lin2: 256 -> 256, ReLU; its tail requests lin3's stream (512 wide: column half kh when KS == 2, both else)
  Ring<DEPTH, G512> rd3;
  const float* wd3 = a.w_d3 + (col + (KS == 2 ? L.kh * FCC_HID : 0)) * 4;
  layer256<DEPTH, KS, 64, 64, G512, RG>(L, rd2, a.w_d2 + col * 4, CS, tA, rd3, wd3, FCC_HID * 4, FCC_FLAT * 4, part, v);
  if (own) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = fmaxf(v[r] + bd2, 0.f);
      tB[r * FCC_XS + col] = v[r];
      if (row0 + r < a.n_dec) a.d2[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  __syncthreads();
  // ---- decoder lin3: 256 -> 512, ReLU
  {
    AccR<RG> acc[G512];
#pragma unroll
    for (int g = 0; g < G512; ++g) acc_zero(acc[g]);
    gemm_run<DEPTH, 64, G512, 0, 0, RG>(rd3, wd3, FCC_HID * 4, FCC_FLAT * 4, tB + xo, acc, dummy, nullptr, 0, 0);
#pragma unroll
    for (int g = 0; g < G512; ++g) {
      const int c512 = col + (KS == 2 ? L.kh : g) * FCC_HID;
      const float b = a.b_d3[c512];
      acc_rows(acc[g], v);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float y = fmaxf(v[r] + b, 0.f);
        if (row0 + r < a.n_dec) a.d3[(long)(row0 + r) * FCC_FLAT + c512] = y;
        if (CONV && RG == 1) smem[LD::ST + c4_st_index(r, c512 >> 4, c512 & 15)] = y;
      }
    }
  }
  if constexpr (CONV) {
    // ---- convT_64 (decoders.py:74-76): this workgroup's rows of d3 = the 4x4x32 inputs of its images -> [n_dec][8][8][32], ReLU
    if (!a.convT_w) return;
    __syncthreads();                                 // d3 is visible to the workgroup; the "up" image has landed (vmcnt(0))
    if constexpr (RG == 1) {
      c4_up_unit<false>(smem + LD::ST, smem + LD::WL2, a.convT_b, nullptr, a.convT_out, a.n_dec, DVAE_ACT_RELU, blockIdx.x);
      return;
    }
    SlotDesc<Geo<4>::SH_NPF> ss;
    init_small_slots<4>(ss, tid, 1);
    f32x4 ps[Geo<4>::SH_NPF];
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int unit = blockIdx.x * RG + u;
      if (unit * 4 >= a.n_dec) break;                // (workgroup-uniform)
      load_small_halo<4>(ps, ss, a.d3, unit, a.n_dec, 1);
      if (u > 0) __syncthreads();                    // the previous unit's reads of the tile
      store_small_halo<4>(ps, ss, smem + LD::ST);
      __syncthreads();
      c4_up_unit<false>(smem + LD::ST, smem + LD::WL2, a.convT_b, nullptr, a.convT_out, a.n_dec, DVAE_ACT_RELU, unit);
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward
template <int DEPTH, int KS, int RG, bool CONV>
__global__ __launch_bounds__(256 * KS) void k_fc_chain_bwd(const BwdArgs P) {
  const dvae_fc_chain_bwd_args& a = P.a;
  constexpr int R = 4 * RG;
  constexpr int NT = 256 * KS, NWV = 4 * KS;
  constexpr int SD = 64 / NWV;
  constexpr int G512 = KS == 2 ? 1 : 2;
  using LD = ChainLds<KS, RG, CONV>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tA = smem + LD::TA;
  float* tB = smem + LD::TB;
  float* red = smem + LD::RED;                      // [NWV][R][64]
  float* part = smem + LD::PART;
  float* warm_sink = smem + LD::SINK;
  static_assert(!CONV || KS == 2, "the conv ends are 512-thread bodies");
  const Lane<KS> L;
  const int tid = L.tid, lane = L.lane, wv = L.wv, col = L.col, xo = L.xo;
  const bool own = L.kh == 0;
  const int row0 = blockIdx.x * R;
  const int n = a.n;
  const int D = a.D, D2 = 2 * a.D;
  constexpr int CS = FCC_HID * 4;

  Ring<DEPTH, 1> r3, r2, re2, dummy;
  Ring<SD, 1> dummy_s;
  fill256<DEPTH, KS, 128>(L, r3, a.w_d3 + col * 4, CS);
  {
    const int nsh = gridDim.x >= 8 ? gridDim.x >> 3 : 1, sh = (blockIdx.x >> 3) % nsh;
    const unsigned sink = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)warm_sink;
    l2_warm(a.w_d3, FCC_HID * FCC_FLAT, sh, nsh, wv, NWV, lane, sink);
    l2_warm(a.w_d2, FCC_HID * FCC_HID, sh, nsh, wv, NWV, lane, sink);
    l2_warm(a.w_e2, FCC_HID * FCC_HID, sh, nsh, wv, NWV, lane, sink);
    l2_warm(a.w_e1, FCC_FLAT * FCC_HID, sh, nsh, wv, NWV, lane, sink);
  }
  if constexpr (CONV) {
    // ---- convT_64's input gradient (decoders.py:74-76 under training.py:157), masked by lin3's ReLU (d3): unit u = images
    // row0 + 4 u .. + 3 -> gd3 (global: lin3's weight gradient reads it) and, with one unit per workgroup, the chain's input tile
    SlotDesc<Geo<4>::BIG_NPF> sd;
    init_big_slots<4>(sd, tid);
    f32x4 pf[Geo<4>::BIG_NPF];
    load_big<4>(pf, sd, a.convT_gout, blockIdx.x * RG, n);
    c4_weight_image_issue(a.convT_w, smem + LD::WL, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* g_out = const_cast<float*>(a.gd3);
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      if (u > 0) load_big<4>(pf, sd, a.convT_gout, blockIdx.x * RG + u, n);
      c4_down_unit<true, RG == 1>(nullptr, a.d3, g_out, tA, FCC_XS, n, DVAE_ACT_NONE, blockIdx.x * RG + u,
                                  smem + LD::WL, smem + LD::BT, smem + LD::RD, pf, sd);
    }
    if (RG != 1) {
      __syncthreads();
      load_rows<FCC_FLAT, NT, R>(a.gd3, row0, n, tA, tid);
    }
  } else {
    load_rows<FCC_FLAT, NT, R>(a.gd3, row0, n, tA, tid);
  }
  float mk[R], v[R];
  // ReLU mask of a 256-wide layer = its saved post-activation output (zero rows beyond n: their gradients are not stored)
  auto load_mask = [&](const float* __restrict__ act) {
    if (own) {
#pragma unroll
      for (int r = 0; r < R; ++r) mk[r] = row0 + r < n ? act[(long)(row0 + r) * FCC_HID + col] : 0.f;
    }
  };
  load_mask(a.d2);
  __syncthreads();
  if constexpr (CONV) {
    if (a.conv_w) {                                   // conv_64's "up" image: region B is free now
      c4_weight_image_issue(a.conv_w, smem + LD::WL2, tid);
      if (RG == 1)
        for (int e = tid; e < C4_ST_FLOATS / 4; e += NT) reinterpret_cast<f32x4*>(smem + LD::ST)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- decoder lin3 input gradient: 512 -> 256, mask d2
  layer256<DEPTH, KS, 128, 64 / KS, 1, RG>(L, r3, a.w_d3 + col * 4, CS, tA, r2, a.w_d2 + col * 4 + (long)L.kh * (64 / KS) * CS, 0, CS, part, v);
  if (own) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = mk[r] > 0.f ? v[r] : 0.f;
      tB[r * FCC_XS + col] = v[r];
      if (row0 + r < n) a.gd2[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  load_mask(a.d1);
  __syncthreads();
  // ---- decoder lin2 input gradient: 256 -> 256, mask d1; request this wave's slice of lin1's (256 -> D, split over ALL waves)
  Ring<SD, 1> r1;
  const int c1 = lane < D ? lane : D - 1;
  const float* w1 = a.w_d1 + ((long)wv * SD * D + c1) * 4;
  layer256<DEPTH, KS, 64, 0, 0, RG>(L, r2, a.w_d2 + col * 4, CS, tB, dummy, nullptr, 0, 0, part, v);
  ring_fill<SD, 1, SD>(r1, w1, 0, D * 4);
  if (own) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = mk[r] > 0.f ? v[r] : 0.f;
      tA[r * FCC_XS + col] = v[r];
      if (row0 + r < n) a.gd1[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  __syncthreads();
  // ---- decoder lin1 input gradient: 256 -> D (dL/dz through the decoder)
  {
    AccR<RG> acc[1];
    acc_zero(acc[0]);
    gemm_run<SD, SD, 1, 0, 0, RG>(r1, w1, 0, D * 4, tA + xo + wv * SD * 4, acc, dummy_s, nullptr, 0, 0);
    acc_rows(acc[0], v);
#pragma unroll
    for (int r = 0; r < R; ++r) red[((wv) * R + (r)) * 64 + (lane)] = v[r];
  }
  fill256<DEPTH, KS, 64>(L, re2, a.w_e2 + col * 4, CS);   // encoder lin2's stream: independent of the latent glue below
  load_mask(a.h2);
  __syncthreads();
  // ---- reparameterisation + KL backward (k_reparam_kl_bwd's arithmetic) -> dml[8][2D] (interleaved) -> tB, zero padded
  {
    const float klw = a.scal[DVAE_S_KLW] * a.coef[DVAE_C_INV_B];
    const int dp2 = (D2 + 3) & ~3;
    for (int t = tid; t < R * 32; t += NT) {
      const int r = t >> 5, d = t & 31;
      if (d < D) {
        float dm = 0.f, dl = 0.f;
        if (row0 + r < n) {
          const long o = (long)(row0 + r) * D + d;
          float g = red[((0) * R + (r)) * 64 + (d)];
#pragma unroll
          for (int w = 1; w < NWV; ++w) g += red[((w) * R + (r)) * 64 + (d)];
          if (a.dz) a.dz[o] = g;
          if (a.dz2) g += a.dz2[o];
          if (a.dz3) g += a.dz3[o];
          const float m = a.mu[o], lv = a.logvar[o];
          dm = g + klw * m;
          dl = klw * 0.5f * (expf(lv) - 1.f);
          if (a.eps) dl += g * a.eps[o] * 0.5f * expf(0.5f * lv);
          if (a.dmu_x) dm += a.dmu_x[o];
          if (a.dlv_x) dl += a.dlv_x[o];
          a.dml[(long)(row0 + r) * D2 + 2 * d] = dm;
          a.dml[(long)(row0 + r) * D2 + 2 * d + 1] = dl;
        }
        tB[r * FCC_XS + 2 * d] = dm;
        tB[r * FCC_XS + 2 * d + 1] = dl;
      }
    }
    if (tid < R * 4) {                           // padding columns D2 .. dp2-1 (at most 2: D2 is even)
      const int r = tid >> 2, c = D2 + (tid & 3);
      if (c < dp2) tB[r * FCC_XS + c] = 0.f;
    }
  }
  __syncthreads();
  // ---- mu_logvar_gen input gradient: 2D -> 256, mask h2
  if (own) {
    AccR<RG> acc;
    acc_zero(acc);
    gemm_small_c<RG>(a.w_ml + col * 4, CS, (D2 + 3) >> 2, tB + xo, acc);
    acc_rows(acc, v);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = mk[r] > 0.f ? v[r] : 0.f;
      tA[r * FCC_XS + col] = v[r];
      if (row0 + r < n) a.gh2[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  load_mask(a.h1);
  __syncthreads();
  // ---- encoder lin2 input gradient: 256 -> 256, mask h1; its tail requests lin1's stream (512 wide)
  Ring<DEPTH, G512> re1;
  const int half = KS == 2 ? L.kh : 0;
  const float* we1 = a.w_e1 + (col + half * FCC_HID) * 4;
  layer256<DEPTH, KS, 64, 64, G512, RG>(L, re2, a.w_e2 + col * 4, CS, tA, re1, we1, FCC_HID * 4, FCC_FLAT * 4, part, v);
  if (own) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = mk[r] > 0.f ? v[r] : 0.f;
      tB[r * FCC_XS + col] = v[r];
      if (row0 + r < n) a.gh1[(long)(row0 + r) * FCC_HID + col] = v[r];
    }
  }
  // masks of the 512-wide output: the conv stack's flattened activation (encoders.py:80)
  float mk2[G512][R];
#pragma unroll
  for (int g = 0; g < G512; ++g)
#pragma unroll
    for (int r = 0; r < R; ++r)
      mk2[g][r] = row0 + r < n ? a.a_flat[(long)(row0 + r) * FCC_FLAT + (KS == 2 ? half : g) * FCC_HID + col] : 0.f;
  __syncthreads();
  // ---- encoder lin1 input gradient: 256 -> 512, mask a_flat
  {
    AccR<RG> acc[G512];
#pragma unroll
    for (int g = 0; g < G512; ++g) acc_zero(acc[g]);
    gemm_run<DEPTH, 64, G512, 0, 0, RG>(re1, we1, FCC_HID * 4, FCC_FLAT * 4, tB + xo, acc, dummy, nullptr, 0, 0);
#pragma unroll
    for (int g = 0; g < G512; ++g) {
      const int c512 = col + (KS == 2 ? half : g) * FCC_HID;
      acc_rows(acc[g], v);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float y = mk2[g][r] > 0.f ? v[r] : 0.f;
        if (row0 + r < n) a.ga_flat[(long)(row0 + r) * FCC_FLAT + c512] = y;
        if (CONV && RG == 1) smem[LD::ST + c4_st_index(r, c512 >> 4, c512 & 15)] = y;
      }
    }
  }
  if constexpr (CONV) {
    // ---- conv_64's input gradient (encoders.py:76-77 under training.py:157), masked by conv3's ReLU: ga_flat rows -> [n][8][8][32]
    if (!a.conv_w) return;
    __syncthreads();                                 // ga_flat is visible to the workgroup; the "up" image has landed (vmcnt(0))
    if constexpr (RG == 1) {
      c4_up_unit<true>(smem + LD::ST, smem + LD::WL2, nullptr, a.conv_act, a.conv_gin, n, DVAE_ACT_NONE, blockIdx.x);
      return;
    }
    SlotDesc<Geo<4>::SH_NPF> ss;
    init_small_slots<4>(ss, tid, 1);
    f32x4 ps[Geo<4>::SH_NPF];
#pragma unroll
    for (int u = 0; u < RG; ++u) {
      const int unit = blockIdx.x * RG + u;
      if (unit * 4 >= n) break;                      // (workgroup-uniform)
      load_small_halo<4>(ps, ss, a.ga_flat, unit, n, 1);
      if (u > 0) __syncthreads();
      store_small_halo<4>(ps, ss, smem + LD::ST);
      __syncthreads();
      c4_up_unit<true>(smem + LD::ST, smem + LD::WL2, nullptr, a.conv_act, a.conv_gin, n, DVAE_ACT_NONE, unit);
    }
  }
}

// ring depth x contraction split of the shipped library.  Measured (profiles/r03_v2_fcc_ab.txt, B = 128 / 1024, forward |
// backward, us, 8 rows per workgroup): <8,1> 28.1 | 33.6 / 30.5 | 40.2; <16,1> 28.0 | 33.4 / 30.3 | 38.4; <8,2> 23.6 | 25.3 /
// 26.1 | 27.7; <16,2> 24.4 | 26.0 / 26.8 | 28.4 -- the second wave per SIMD helps, a deeper ring does not: with 8 rows the
// launch is bound by its 12.8 k MFMAs per SIMD (21.6 us).  Round 5: FOUR rows per workgroup (RG = 1: half the MFMAs per
// workgroup, twice the workgroups; the bound becomes the L2 -> register weight stream, ~84 GB/s per CU) up to
// FCC_R4_MAX_ROWS batch rows, where the doubled workgroup count still fits the chip in one wave; 8 rows above.
// Debug builds can A/B the other instantiations with DVAE_FCC_VARIANT = 100 * rows + 10 * DEPTH + KS (rows = 0: by batch)
#ifndef FCC_DEFAULT_VARIANT
#define FCC_DEFAULT_VARIANT 82
#endif
#ifndef FCC_R4_DEPTH
#define FCC_R4_DEPTH 8            // same box, alone, 128 / 256 / 1024 rows: depth 8 18.3 / 17.9 / 20.6 us forward, depth 16 26.1 / 25.3 /
#endif                            // 26.5 (8 rows per workgroup: 26.2 / 25.3 / 26.6): profiles/r05_final1_fcc_ab.txt
#define FCC_R4_MAX_ROWS 1024

// batch rows per workgroup of the chain kernels for a launch over n rows (= the granularity of the forward's KL partial blocks)
int fc_chain_rows(int n) {
  static const int variant = env_int("DVAE_FCC_VARIANT", 0);
  if (variant >= 100) return variant / 100 == 4 ? 4 : 8;
  return n <= FCC_R4_MAX_ROWS ? 4 : 8;
}

template <int DEPTH, int KS, int RG>
static int launch_fwd_t(const FwdArgs& P, hipStream_t s) {
  if (P.a.conv_in) {
    if constexpr (KS != 2) {                          // (debug-build variants with 256 threads)
      set_error("dvae_fc_chain_fwd: the conv ends need the 512-thread variants");
      return -1;
    }
    if constexpr (KS == 2) {
      constexpr int lds = ChainLds<KS, RG, true>::TOTAL * (int)sizeof(float);
      static DeviceOnce attr;
      if (attr.first())
        (void)hipFuncSetAttribute((const void*)k_fc_chain_fwd<DEPTH, KS, RG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((k_fc_chain_fwd<DEPTH, KS, RG, true>), dim3((P.a.n_enc + 4 * RG - 1) / (4 * RG)), dim3(256 * KS), lds, s, P);
      return 0;
    }
  }
  constexpr int lds = ChainLds<KS, RG, false>::TOTAL * (int)sizeof(float);
  hipLaunchKernelGGL((k_fc_chain_fwd<DEPTH, KS, RG, false>), dim3((P.a.n_enc + 4 * RG - 1) / (4 * RG)), dim3(256 * KS), lds, s, P);
  return 0;
}
template <int DEPTH, int KS, int RG>
static int launch_bwd_t(const BwdArgs& P, hipStream_t s) {
  if (P.a.convT_gout) {
    if constexpr (KS != 2) {
      set_error("dvae_fc_chain_bwd: the conv ends need the 512-thread variants");
      return -1;
    }
    if constexpr (KS == 2) {
      constexpr int lds = ChainLds<KS, RG, true>::TOTAL * (int)sizeof(float);
      static DeviceOnce attr;
      if (attr.first())
        (void)hipFuncSetAttribute((const void*)k_fc_chain_bwd<DEPTH, KS, RG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((k_fc_chain_bwd<DEPTH, KS, RG, true>), dim3((P.a.n + 4 * RG - 1) / (4 * RG)), dim3(256 * KS), lds, s, P);
      return 0;
    }
  }
  constexpr int lds = ChainLds<KS, RG, false>::TOTAL * (int)sizeof(float);
  hipLaunchKernelGGL((k_fc_chain_bwd<DEPTH, KS, RG, false>), dim3((P.a.n + 4 * RG - 1) / (4 * RG)), dim3(256 * KS), lds, s, P);
  return 0;
}

int launch_fc_chain_fwd(const dvae_fc_chain_fwd_args* a, hipStream_t s) {
  FwdArgs P;
  P.a = *a;
  static const int variant = env_int("DVAE_FCC_VARIANT", 0) % 100;
  const bool r4 = fc_chain_rows(a->n_enc) == 4;
  int rc = 0;
  switch (variant) {
#ifdef DVAE_DEBUG_SWITCHES
    case 81: rc = r4 ? launch_fwd_t<8, 1, 1>(P, s) : launch_fwd_t<8, 1, 2>(P, s); break;
    case 82: rc = r4 ? launch_fwd_t<8, 2, 1>(P, s) : launch_fwd_t<8, 2, 2>(P, s); break;
    case 161: rc = r4 ? launch_fwd_t<16, 1, 1>(P, s) : launch_fwd_t<16, 1, 2>(P, s); break;
    case 162: rc = r4 ? launch_fwd_t<16, 2, 1>(P, s) : launch_fwd_t<16, 2, 2>(P, s); break;
#endif
    default:
      rc = r4 ? launch_fwd_t<FCC_R4_DEPTH, 2, 1>(P, s) : launch_fwd_t<FCC_DEFAULT_VARIANT / 10, FCC_DEFAULT_VARIANT % 10, 2>(P, s);
      break;
  }
  if (rc != 0) return rc;
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_fc_chain_bwd(const dvae_fc_chain_bwd_args* a, hipStream_t s) {
  BwdArgs P;
  P.a = *a;
  static const int variant = env_int("DVAE_FCC_VARIANT", 0) % 100;
  const bool r4 = fc_chain_rows(a->n) == 4;
  int rc = 0;
  switch (variant) {
#ifdef DVAE_DEBUG_SWITCHES
    case 81: rc = r4 ? launch_bwd_t<8, 1, 1>(P, s) : launch_bwd_t<8, 1, 2>(P, s); break;
    case 82: rc = r4 ? launch_bwd_t<8, 2, 1>(P, s) : launch_bwd_t<8, 2, 2>(P, s); break;
    case 161: rc = r4 ? launch_bwd_t<16, 1, 1>(P, s) : launch_bwd_t<16, 1, 2>(P, s); break;
    case 162: rc = r4 ? launch_bwd_t<16, 2, 1>(P, s) : launch_bwd_t<16, 2, 2>(P, s); break;
#endif
    default:
      rc = r4 ? launch_bwd_t<FCC_R4_DEPTH, 2, 1>(P, s) : launch_bwd_t<FCC_DEFAULT_VARIANT / 10, FCC_DEFAULT_VARIANT % 10, 2>(P, s);
      break;
  }
  if (rc != 0) return rc;
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
