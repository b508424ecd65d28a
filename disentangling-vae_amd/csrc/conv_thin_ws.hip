// k_down_thin_ws<C, MODE>: the "down" kernel of the thin ends (conv1 forward, reference encoders.py:54,73; convT3 input
// gradient, decoders.py:65,82 backward) for fp32 NCHW images at 64x64, wave-specialised.
//
// Why: k_down_thin (conv_thin.hip) ran at 52-59 us per launch at B = 1024 where a streaming kernel with the same traffic
// (50 MB in, 138 MB out) takes 28 us (tools/ubench/hbm_mix.hip, profiles/r04_v22_hbm_mix.txt) and the matrix core needs 23.
// Its waves load, multiply AND store, so every wait for a prefetched tile (s_waitcnt vmcnt counts loads and stores in one
// in-order counter) also waits for the previous unit's 16 output stores, and the register allocator's reuse of a prefetch
// register put a full vmcnt(0) between the MFMA phase and the epilogue.  And: an fp32 MFMA runs at the vector rate ON the
// vector ALUs -- every VALU instruction a wave of the SIMD issues takes 2-3 cycles away from the matrix pipe
// (tools/ubench/mfma_chain.hip, profiles/r04_v26_mfma_chain.txt: 61.5 cycles per chained MFMA alone, 75 with 6 VALU beside it,
// whichever wave issues them) -- so the epilogue is kept as short as the data allows.
// Roles in a 512-thread workgroup (two workgroups per CU):
//   * waves 0-3, compute: one small row of 32 pixels each; per unit the 8 C chained MFMAs of the row, TRANSPOSED (A = weights,
//     B = pixels: a lane of the D fragment holds 4 consecutive channels of one pixel); the epilogue of the previous unit (bias,
//     ReLU, mask / bit collection, four 16-byte LDS writes into an output stage), the operand reads of the next unit and
//     nothing else ride in the chain's issue shadow.  No vector memory instruction at all after the prologue.
//   * waves 4-5, loaders: the next tiles HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB per instruction: a wave that
//     shares its SIMD with MFMA-streaming waves issues one instruction per 40-60 cycles, so a loader may cost ~30 per unit),
//     three tiles ahead in a ring of four stages (a ring of three measured the same: profiles/r04_v30_thin_ab.txt); lanes whose
//     row lies outside the image are masked off and find the zeros the stage was initialised with.
//   * waves 6-7, drainers: output stage -> HBM, 16 bytes per lane, every 8 lanes one full 128-byte line; their stores are
//     the only thing that ever waits for the memory system, and nothing waits for them.
//   * one s_barrier per unit: "tile u + 1 has landed; tile u is read; output stage u - 1 is written; stage u - 2 is drained".
//   * the 8 units of an image run at the same time on ONE XCD (workgroup b -> XCD b & 7), so the halo rows two units share
//     come out of that XCD's L2 instead of HBM a second time.
// MODE: 0 plain; 2 output masked by a bit plane (128 words per unit, fetched with the tile); 3 plain + emit the output's bit
// plane.  The accumulation order is k_down_thin's (one chain over k = 16 cb + 4 kh + kw): results are bit-identical to it.
#include <type_traits>
#include "common.h"
#include "wgrad_reduce.h"

#pragma clang diagnostic ignored "-Winline-asm"     // dma16s names m0 in its clobber list on purpose

namespace dvae {

template <int C, int MODE>
struct ThinWsGeo {
  // A channel plane in LDS = the tile's 10 image rows as they lie in memory (64 floats each: 16-byte LDS-DMA is lane-linear,
  // rows cannot be padded) followed by a ZERO ZONE (floats 640..895, never written).  The two operand positions outside the
  // image -- column -1: lane 0 at kw = 0; column 64: lane 63 at kw = 3 -- are read from the zero zone: those two lanes carry a
  // different base address, no lane select is executed.
  static constexpr int PLANE = 896;                        // 3 transfers of 256 floats + 128 more zeros
  static constexpr int ZZ = 640;                           // ZZ + 64 kh + 2 < PLANE for kh <= 3
  static constexpr int NS = 4;                             // ring of tile stages: one being read, three in flight
  static constexpr int BITS_OFF = C * PLANE;               // MODE 2: the unit's 128 mask words
  static constexpr int STAGE = BITS_OFF + (MODE == 2 ? 128 : 0);
  static constexpr int NDMA = 3 * C + (MODE == 2 ? 1 : 0); // 1 KB transfers per tile (64 lanes x 16 bytes)
  static constexpr int NPF0 = (NDMA + 1) / 2, NPF1 = NDMA / 2;   // per loader wave
  static constexpr int OST = 128 * 32 + 128;               // output stage: 128 pixels x 32 channels (+ 128 bit-plane words)
  static constexpr int OUT0 = NS * STAGE;                  // the two output stages
  static constexpr int BIAS = OUT0 + 2 * OST;              // the 32 biases
  static constexpr int TOTAL = BIAS + 32;
  static_assert(STAGE % 4 == 0 && OST % 4 == 0, "16-byte alignment of the stages");
};

// one LDS-DMA transfer: every ACTIVE lane l of the wave moves 16 bytes from base + its own 32-bit offset to LDS byte
// lds_addr + 16 l (inactive lanes leave their LDS bytes alone)
// (called with all 64 lanes active: the lane mask is applied to EXEC around the instruction and EXEC is set back to all ones)
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_addr, unsigned long long lanes) {
  asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff),
               "s"(sbase), "s"(lds_addr), "s"(lanes)
               : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void thin_barrier() { asm volatile("s_barrier" ::: "memory"); }

// ABL (debug builds only, results invalid): timing ablations -- 1 no output stores, 2 no MFMAs, 4 no LDS-DMA transfers,
// 16 no epilogue at all; variants with valid results: 32 compute waves at s_setprio 3, 64 helper waves at s_setprio 3
template <int C, int MODE, int ABL = 0>
__global__ __launch_bounds__(512, 4) void k_down_thin_ws(const float* __restrict__ big, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int N, int act,
                                                         uint32_t* __restrict__ bits) {
  using G = ThinWsGeo<C, MODE>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (image lane, part of the image): the 8 parts of an image sit in 8 neighbouring slots of one XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int part = slot & 7;
  const int ipi = gridDim.x >> 3;                          // images per persistent step (gridDim.x is a multiple of 64)
  const int n0 = xcd + 8 * (slot >> 3);
  const int sy0 = part * 4;
  auto nxt = [](int b) { return (b + 1) & (G::NS - 1); };

  // the tile stages start as zeros: zero zones, and the image rows that lie outside the image for this workgroup's part
  // (row -1 of part 0, row 64 of part 7), which no transfer ever writes
  for (int e = tid; e < G::NS * G::STAGE / 4; e += 512) reinterpret_cast<f32x4*>(smem)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 32) smem[G::BIAS + tid] = (bias && MODE != 2) ? bias[tid] : 0.f;
  __syncthreads();

  if (wv >= 4 && (ABL & 64)) __builtin_amdgcn_s_setprio(3);
  if (wv < 4 && (ABL & 32)) __builtin_amdgcn_s_setprio(3);
  if (wv >= 6) {
    // ---------------------------------------------------------------- drainers: output stage -> HBM
    const int dl = (wv - 6) * 64 + lane;                   // 0..127
    // 16 KB = 1024 chunks of 16 bytes, 8 per lane: chunk X = j * 128 + dl -> pixel X >> 3, slot X & 7 (slot = channel chunk
    // XOR (pixel >> 1) & 7: the compute waves' bank swizzle; a line is still written whole, by 8 neighbouring lanes)
    unsigned goff[8];                                      // byte offsets into the unit's 16 KB output block
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int X = j * 128 + dl, p = X >> 3, c = (X & 7) ^ ((p >> 1) & 7);
      goff[j] = (unsigned)(p * 32 + c * 4) * 4u;
    }
    auto drain = [&](int ob, int n) {
      const float* os = smem + G::OUT0 + ob * G::OST;
      char* obase = reinterpret_cast<char*>(out + (((long)n * 32 + sy0) * 32) * 32);   // wave-uniform
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(os + (j * 128 + dl) * 4);
      f32x4 bw = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 3 && dl < 32) bw = *reinterpret_cast<const f32x4*>(os + 4096 + dl * 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!(ABL & 1)) *reinterpret_cast<f32x4*>(obase + goff[j]) = v[j];
        else asm volatile("" ::"v"(v[j][0]), "v"(v[j][3]));
      }
      if (MODE == 3 && dl < 32) {
        char* bbase = reinterpret_cast<char*>(bits + ((long)n * 32 + sy0) * 32);
        if (!(ABL & 1)) *reinterpret_cast<f32x4*>(bbase + (unsigned)dl * 16u) = bw;
        else asm volatile("" ::"v"(bw[0]));
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the stage is read before the next barrier
    };
    thin_barrier();                                        // prologue barrier
    int m = 0;
    for (int n = n0; n < N; n += ipi, ++m) {
      thin_barrier();                                      // barrier m: output stage of unit m - 2 is complete
      if (m >= 2) drain(m & 1, n - 2 * ipi);
    }
    thin_barrier();                                        // final barrier: both output stages are complete
    if (m >= 2) drain(m & 1, n0 + (m - 2) * ipi);
    if (m >= 1) drain((m - 1) & 1, n0 + (m - 1) * ipi);
    return;
  }

  if (wv >= 4) {
    // ---------------------------------------------------------------- loaders
    // transfer d = lw + 2 k: d < 3 C: channel plane d / 3, 1 KB block d % 3 of its 10 rows (lane l: row 4 q + l / 16, columns
    // 4 (l % 16) ..+3); d = 3 C (MODE 2): the unit's 128 mask words.  Source = a wave-uniform base that moves with the image
    // + a per-lane 32-bit offset that never changes; lanes whose row is outside the tile or outside the image are inactive.
    const int lw = wv - 4;
    constexpr int NPFM = G::NPF0;
    unsigned voff[NPFM];
    unsigned long long lanes[NPFM];                        // active lanes of transfer k (never empty for d < NDMA)
#pragma unroll
    for (int k = 0; k < NPFM; ++k) {
      const int d = lw + 2 * k;
      voff[k] = 0u;
      bool on = false;
      if (d < 3 * C) {
        const int c = d / 3, q = d - 3 * c;
        const int r = 4 * q + (lane >> 4), col = 4 * (lane & 15);
        const int by = 2 * sy0 - 1 + r;
        on = r < 10 && by >= 0 && by < 64;
        voff[k] = on ? (unsigned)((c * 64 + by) * 64 + col) * 4u : 0u;
      } else if (MODE == 2 && d == 3 * C) {
        on = lane < 32;
        voff[k] = on ? (unsigned)lane * 16u : 0u;
      }
      lanes[k] = __builtin_amdgcn_ballot_w64(on);
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const float* ibase = big + (long)n0 * C * 4096;        // image n0 (wave-uniform); + ipi images per issue
    const uint32_t* bbase = MODE == 2 ? bits + ((long)n0 * 32 + sy0) * 32 : nullptr;
    auto issue = [&](int buf) {
      if (ABL & 4) return;
      const unsigned base = lds0 + (unsigned)buf * (G::STAGE * 4u);
#pragma unroll
      for (int k = 0; k < NPFM; ++k) {
        const int d = lw + 2 * k;                          // wave-uniform
        if (d < 3 * C) {
          dma16s(ibase, voff[k], __builtin_amdgcn_readfirstlane(base + (unsigned)((d / 3) * G::PLANE + (d % 3) * 256) * 4u), lanes[k]);
        } else if (MODE == 2 && d == 3 * C) {
          dma16s(bbase, voff[k], __builtin_amdgcn_readfirstlane(base + G::BITS_OFF * 4u), lanes[k]);
        }
      }
      ibase += (long)ipi * C * 4096;
      if (MODE == 2) bbase += (long)ipi * 1024;
    };
    auto wait_landed = [&](int newer) {                    // all transfers but those of the `newer` newest tiles have landed
      if (newer <= 0) wait_vmcnt<0>();
      else if (newer == 1) { if (lw == 0) wait_vmcnt<G::NPF0>(); else wait_vmcnt<G::NPF1>(); }
      else { if (lw == 0) wait_vmcnt<2 * G::NPF0>(); else wait_vmcnt<2 * G::NPF1>(); }
    };
    // a tile is issued three units before its operands are read, and has to have landed one unit before: an LDS-DMA transfer
    // takes ~1 us from issue to landing under load, a unit ~1 us
    int n = n0;
    if (n < N) issue(0);
    if (n + ipi < N) issue(1);
    if (n + 2 * ipi < N) issue(2);
    wait_landed((n + ipi < N) + (n + 2 * ipi < N));        // tile(n0) has landed
    thin_barrier();                                        // prologue barrier
    int buf = 0;
    for (; n < N; n += ipi) {
      // the compute waves read tile(n + ipi) next (stage buf + 1); stage buf + 3 = buf - 1 was released by the last barrier
      if (n + 3 * ipi < N) issue((buf + 3) & 3);
      wait_landed((n + 2 * ipi < N) + (n + 3 * ipi < N));  // tile(n + ipi) has landed
      thin_barrier();
      buf = nxt(buf);
    }
    thin_barrier();                                        // final barrier
    return;
  }

  // ------------------------------------------------------------------ compute waves: wave = small row sy0 + wv
  const int i = lane & 31, h = lane >> 5;
  float wreg[8 * C];                                       // A operand: w[cs = i][k = 2 kk + h], k = 16 cb + 4 kh + kw
#pragma unroll
  for (int kk = 0; kk < 8 * C; ++kk) wreg[kk] = w[i * (16 * C) + 2 * kk + h];
  // D register e = 4 g + q of lane (i, h): channel 8 g + 4 h + q of pixel i
  // B operand of step kk: plane cb, tile row 2 wv + kh, column 2 i + (2 kwb + h) - 1; lane 0 (kwb = 0) and lane 63
  // (kwb = 1) would read columns -1 / 64: they read the plane's zero zone instead
  const int abase_e = lane == 0 ? G::ZZ : (2 * wv) * 64 + 2 * i + h - 1;
  const int abase_o = lane == 63 ? G::ZZ : (2 * wv) * 64 + 2 * i + h + 1;
  const bool relu = MODE == 3 || (MODE == 0 && act == DVAE_ACT_RELU);   // MODE 2 is an input gradient: no activation, no bias
  // output stage: pixel (wv * 32 + i) row of 32 floats, 16-byte chunk (2 g + h) at slot (2 g + h) ^ ((i >> 1) & 7)
  const int obase = (wv * 32 + i) * 32;
  const int osw = (i >> 1) & 7;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the only vector loads of these waves
  thin_barrier();                                          // prologue barrier: tile(n0) is in stage 0

  struct Ops {                                             // operands of one unit, as read from its tile
    float a[8 * C];
    uint32_t word;
  };
  struct Pend {                                            // what the deferred epilogue needs
    f32x16 acc;
    uint32_t word;
  };
  constexpr int NM = 8 * C;                                // MFMAs per unit
  auto read_op = [&](int buf, Ops& o, int kk) {             // operand kk of the unit whose tile is in stage `buf`
    const float* st = smem + buf * G::STAGE;
    const int kh = (kk >> 1) & 3, cb = kk >> 3;
    o.a[kk] = st[((kk & 1) ? abase_o : abase_e) + cb * G::PLANE + kh * 64];
  };
  auto read_word = [&](int buf, Ops& o) {
    o.word = 0;
    if (MODE == 2) o.word = reinterpret_cast<const uint32_t*>(smem + buf * G::STAGE)[G::BITS_OFF + wv * 32 + i];
  };
  // epilogue of D registers 4 g .. 4 g + 3 (channels 8 g + 4 h + 0..3 of pixel i): one 16-byte LDS write
  auto epi_group = [&](const Pend& p, int g, int ob, uint32_t& oword) {
    f32x4 o, bq = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2) bq = *reinterpret_cast<const f32x4*>(smem + G::BIAS + 8 * g + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = 4 * g + q;
      float v = p.acc[e];
      if (MODE != 2) v += bq[q];
      if (MODE == 3) {
        const bool pos = v > 0.f;
        oword = (oword << 1) | (pos ? 1u : 0u);            // register e -> bit 15 - e; spread below
        v = pos ? v : 0.f;
      } else if (relu) {
        v = v > 0.f ? v : 0.f;
      }
      if (MODE == 2) {                                     // p.word = the pixel's word >> 4 h: bit 8 g + q is this register's channel
        const int mbit = __builtin_amdgcn_sbfe((int)p.word, 8 * g + q, 1);      // 0 or -1
        v = __int_as_float(__float_as_int(v) & mbit);
      }
      o[q] = v;
    }
    float* os = smem + G::OUT0 + ob * G::OST;
    *reinterpret_cast<f32x4*>(os + obase + (((2 * g + h) ^ osw) << 2)) = o;
  };
  auto epi_tail = [&](int ob, uint32_t oword) {
    if (MODE == 3) {
      // oword bit 15 - e = [register e > 0], e = 4 g + q -> channel 8 g + 4 h + q: reverse the 16 bits, spread the nibbles
      // to bytes, shift by 4 h, OR with the other half's lane (i, 1 - h), lane (i, 0) writes the pixel's word
      uint32_t r = __builtin_bitreverse32(oword) >> 16;    // bit e = register e
      r = (r & 0xFu) | ((r & 0xF0u) << 4) | ((r & 0xF00u) << 8) | ((r & 0xF000u) << 12);
      r <<= 4 * h;
      r |= (uint32_t)__builtin_amdgcn_ds_bpermute(((lane ^ 32) << 2), (int)r);
      uint32_t* ow = reinterpret_cast<uint32_t*>(smem + G::OUT0 + ob * G::OST + 4096);
      if (h == 0) ow[wv * 32 + i] = r;
    }
  };
  // chain of the unit whose operands are in `oc`; reads the next unit's (tile in `nbuf`) into `on`; finishes `old` into
  // output stage `ob`
  // (the next tile's stage is read even when there is no next unit: stale LDS, never used)
  auto unit = [&](Ops& oc, Ops& on, int nbuf, Pend& cur, const Pend& old, int ob, auto have_old) {
    constexpr bool HAVE_OLD = decltype(have_old)::value;
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): this unit's operands have returned (its tile is read)
    thin_barrier();                                        // ... and the next tile has landed
    cur.word = oc.word >> (4 * h);
    read_word(nbuf, on);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t oword = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) cur.acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NM; ++kk) {
      if (!(ABL & 2)) cur.acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[kk], oc.a[kk], cur.acc, 0, 0, 0);
      else asm volatile("" ::"v"(oc.a[kk]));
      read_op(nbuf, on, kk);                               // the next unit's operands, one per slot
      if (HAVE_OLD && !(ABL & 16)) {
        // the four groups spread over the chain (C = 3: slots 2, 8, 14, 20; C = 1: slots 1, 3, 5, 7)
        constexpr int STEP = NM / 4, FIRST = NM >= 24 ? 2 : STEP - 1;
        if (kk >= FIRST && (kk - FIRST) % STEP == 0 && (kk - FIRST) / STEP < 4) epi_group(old, (kk - FIRST) / STEP, ob, oword);
        if (kk == FIRST + 3 * STEP) epi_tail(ob, oword);
      }
      __builtin_amdgcn_sched_barrier(0);                   // keep the slots in this order
    }
  };
  auto finish = [&](const Pend& p, int ob) {
    if (ABL & 16) { asm volatile("" ::"v"(p.acc[0]), "v"(p.acc[15])); return; }
    uint32_t oword = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) epi_group(p, g, ob, oword);
    epi_tail(ob, oword);
  };
  Pend pa, pb;
  Ops oa, ob_;
  int n = n0, buf = 0, m = 0;                              // buf: the tile of unit m (image n)
  if (n < N) {
#pragma unroll
    for (int kk = 0; kk < NM; ++kk) read_op(0, oa, kk);
    read_word(0, oa);
    unit(oa, ob_, nxt(buf), pa, pb, 0, std::false_type{});
    n += ipi, buf = nxt(buf), ++m;
    while (true) {
      if (n >= N) { finish(pa, (m - 1) & 1); break; }
      unit(ob_, oa, nxt(buf), pb, pa, (m - 1) & 1, std::true_type{});
      n += ipi, buf = nxt(buf), ++m;
      if (n >= N) { finish(pb, (m - 1) & 1); break; }
      unit(oa, ob_, nxt(buf), pa, pb, (m - 1) & 1, std::true_type{});
      n += ipi, buf = nxt(buf), ++m;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the last output stage is written
  thin_barrier();                                          // final barrier
}

// fp32 NCHW images only; returns 1 if the shape is not covered (the caller falls back to k_down_thin)
int launch_down_thin_ws(const ConvArgs& a, hipStream_t s) {
  static const bool off = env_off("DVAE_THIN_WS");                   // A/B switch, debug builds only
  static const int min_n = env_int("DVAE_THIN_WS_MIN_N", 192);       // below: fewer than 3 tiles per workgroup
  static const int grid = env_int("DVAE_THIN_WS_GRID", 512);
  if (off || a.N < min_n || a.mask) return 1;
  if (a.mask_bits && (a.out_bits || a.bias || a.act != DVAE_ACT_NONE)) return 1;   // MODE 2 is a bare masked input gradient
  if (a.out_bits && a.act != DVAE_ACT_RELU) return 1;                               // MODE 3 = conv1 forward: bias + ReLU + bits
  if (((uintptr_t)a.big | (uintptr_t)a.out) & 15) return 1;          // tiles by 16-byte LDS-DMA, output in 16-byte stores
  uint32_t* bits = a.mask_bits ? const_cast<uint32_t*>(a.mask_bits) : a.out_bits;
  if (bits && ((uintptr_t)bits & 15)) return 1;
#define DVAE_DTW(C, MODE)                                                                                                   \
  do {                                                                                                                     \
    constexpr int lds = ThinWsGeo<C, MODE>::TOTAL * 4;                                                                     \
    static DeviceOnce attr;                                                                                                \
    if (attr.first()) (void)hipFuncSetAttribute((const void*)k_down_thin_ws<C, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((k_down_thin_ws<C, MODE>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.out, a.N, a.act, bits); \
  } while (0)
#ifdef DVAE_DEBUG_SWITCHES
#define DVAE_DTWA(MODE, ABL)                                                                                               \
  do {                                                                                                                     \
    constexpr int lds = ThinWsGeo<3, MODE>::TOTAL * 4;                                                                     \
    static DeviceOnce attr;                                                                                                \
    if (attr.first()) (void)hipFuncSetAttribute((const void*)k_down_thin_ws<3, MODE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((k_down_thin_ws<3, MODE, ABL>), dim3(grid), dim3(512), lds, s, a.big, a.w, a.bias, a.out, a.N, a.act, bits); \
  } while (0)
  static const int abl = env_int("DVAE_THIN_WS_ABLATE", 0);
  if (abl && a.Cb == 3 && (a.mask_bits || a.out_bits)) {
    const bool m2 = a.mask_bits != nullptr;
    switch (abl) {
      case 1: if (m2) DVAE_DTWA(2, 1); else DVAE_DTWA(3, 1); break;
      case 2: if (m2) DVAE_DTWA(2, 2); else DVAE_DTWA(3, 2); break;
      case 3: if (m2) DVAE_DTWA(2, 3); else DVAE_DTWA(3, 3); break;
      case 4: if (m2) DVAE_DTWA(2, 4); else DVAE_DTWA(3, 4); break;
      case 6: if (m2) DVAE_DTWA(2, 6); else DVAE_DTWA(3, 6); break;
      case 7: if (m2) DVAE_DTWA(2, 7); else DVAE_DTWA(3, 7); break;
      case 17: if (m2) DVAE_DTWA(2, 17); else DVAE_DTWA(3, 17); break;
      case 21: if (m2) DVAE_DTWA(2, 21); else DVAE_DTWA(3, 21); break;
      case 32: if (m2) DVAE_DTWA(2, 32); else DVAE_DTWA(3, 32); break;
      case 64: if (m2) DVAE_DTWA(2, 64); else DVAE_DTWA(3, 64); break;
      default: return 1;
    }
    DVAE_CHECK_LAUNCH();
    return 0;
  }
#undef DVAE_DTWA
#endif
  if (a.Cb == 1) {
    if (a.mask_bits) DVAE_DTW(1, 2); else if (a.out_bits) DVAE_DTW(1, 3); else DVAE_DTW(1, 0);
  } else if (a.Cb == 3) {
    if (a.mask_bits) DVAE_DTW(3, 2); else if (a.out_bits) DVAE_DTW(3, 3); else DVAE_DTW(3, 0);
  } else {
    return 1;
  }
#undef DVAE_DTW
  DVAE_CHECK_LAUNCH();
  return 0;
}

// =====================================================================================================================
// k_wgrad_thin_ws<C, BIAS_BIG>: the weight gradient of the thin ends (conv1: encoders.py:54 under training.py:157, dw[cs][cb]
// [kh][kw] = sum over pixels of dy[n][sy][sx][cs] x[n][cb][2 sy - 1 + kh][2 sx - 1 + kw]; convT3: decoders.py:65, the same sum
// with the roles of the tensors swapped by the caller) for fp32 images at 64x64, wave-specialised like k_down_thin_ws.
// k_wgrad_thin (conv_thin.hip) took 58 us per launch at B = 1024 for 26 us of HBM reads (184 MB) and 31 us of matrix-core work
// (it pads the 48 (cb, tap) columns to two 32-column tiles), one after the other (its waves load, stage and multiply in turn).
// Here:
//   * v_mfma_f32_16x16x4_f32: M = 16 cs x N = 16 taps of ONE input channel x K = 4 pixels: 2 x C tiles, nothing padded
//     (48 instead of 64 MFMA-cycles per pixel pair), six independent accumulators per wave;
//   * waves 0-3, compute: one small row of 32 pixels each = 8 steps of 4 pixels; per step one ds_read_b64 (the dy of channels
//     2 i, 2 i + 1: the two M tiles interleave the channels so that a lane's pair is one 8-byte read) and C ds_read_b32 (x
//     patch values), the NEXT unit's operands read under the current unit's MFMAs; no vector memory instruction at all;
//   * waves 4-7, loaders: both tiles HBM -> LDS by LDS-DMA, 16 + 3 C transfers of 1 KB per unit, two units ahead in a ring of
//     three stages (one workgroup per CU: see the launcher); image rows outside the image are masked lanes over LDS zeros, the two columns outside the image are read
//     from a zero zone by the lanes concerned (as in k_down_thin_ws);
//   * the accumulators live in registers for the whole kernel; at the end the four waves' partial sums are added in a fixed
//     order through LDS and written in k_wgrad_thin's partial-buffer layout, so k_wgrad_thin_reduce finishes the job.
// Bias gradient partials: the sum of the small-side tensor per channel (conv1), or the per-column sums of the big side
// (BIAS_BIG, convT3: the reduction uses the four taps that cover every pixel once) -- VALU adds on the operands already read.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int C, bool BIAS_BIG>
struct ThinWgGeo {
  static constexpr int PLANE = 768;                        // 3 transfers of 256 floats; rows 10, 11 are never written: zeros
  static constexpr int ZZ = 640;
  static constexpr int SM = 128 * 32;                      // small tile: 128 pixels x 32 channels
  static constexpr int STAGE = SM + C * PLANE;
  static constexpr int NS = 3;
  static constexpr int NDMA = 16 + 3 * C;
  static constexpr int TOTAL = NS * STAGE;
  static constexpr int NACC = 2 * C * 4;                   // accumulator registers per lane
  static_assert(4 * (NACC + 2 + C) * 64 <= TOTAL, "the final cross-wave reduction fits the stages");
};

template <int C, bool BIAS_BIG, int ABL = 0>
__global__ __launch_bounds__(512, 4) void k_wgrad_thin_ws(const float* __restrict__ big, const float* __restrict__ small,
                                                          float* __restrict__ ws, int N) {
  using G = ThinWgGeo<C, BIAS_BIG>;
  constexpr int NT32 = (16 * C + 31) / 32;                 // k_wgrad_thin's 32-column tiles (partial-buffer layout)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int part = slot & 7;
  const int ipi = gridDim.x >> 3;
  const int n0 = xcd + 8 * (slot >> 3);
  const int sy0 = part * 4;
  auto nxt = [](int b) { return b == 2 ? 0 : b + 1; };

  for (int e = tid; e < G::NS * G::STAGE / 4; e += 512) reinterpret_cast<f32x4*>(smem)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  if (wv >= 4) {
    // ---------------------------------------------------------------- loaders
    // transfer d = lw + 4 k: d < 16: 1 KB block d of the small tile (contiguous in memory); d >= 16: big tile, channel plane
    // (d - 16) / 3, block (d - 16) % 3 of its 10 rows (lane l: row 4 q + l / 16, columns 4 (l % 16) ..+3)
    const int lw = wv - 4;
    constexpr int NPFM = (G::NDMA + 3) / 4;
    unsigned voff[NPFM];
    unsigned long long lanes[NPFM];
#pragma unroll
    for (int k = 0; k < NPFM; ++k) {
      const int d = lw + 4 * k;
      voff[k] = 0u;
      bool on = false;
      if (d < 16) {
        on = true;
        voff[k] = (unsigned)(d * 1024 + lane * 16);
      } else if (d < G::NDMA) {
        const int c = (d - 16) / 3, q = (d - 16) - 3 * c;
        const int r = 4 * q + (lane >> 4), col = 4 * (lane & 15);
        const int by = 2 * sy0 - 1 + r;
        on = r < 10 && by >= 0 && by < 64;
        voff[k] = on ? (unsigned)((c * 64 + by) * 64 + col) * 4u : 0u;
      }
      lanes[k] = __builtin_amdgcn_ballot_w64(on);
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const float* ibase = big + (long)n0 * C * 4096;                  // wave-uniform; + ipi images per issue
    const float* sbase = small + (((long)n0 * 32 + sy0) * 32) * 32;
    auto issue = [&](int buf) {
      if (ABL & 4) return;
      const unsigned base = lds0 + (unsigned)buf * (G::STAGE * 4u);
#pragma unroll
      for (int k = 0; k < NPFM; ++k) {
        const int d = lw + 4 * k;                          // wave-uniform
        if (d < 16) dma16s(sbase, voff[k], __builtin_amdgcn_readfirstlane(base + (unsigned)d * 1024u), lanes[k]);
        else if (d < G::NDMA)
          dma16s(ibase, voff[k], __builtin_amdgcn_readfirstlane(base + (unsigned)(G::SM + ((d - 16) / 3) * G::PLANE + ((d - 16) % 3) * 256) * 4u), lanes[k]);
      }
      ibase += (long)ipi * C * 4096;
      sbase += (long)ipi * 32768;
    };
    auto wait_landed = [&](bool newest_in_flight) {        // all transfers but the newest tile's have landed
      if (!newest_in_flight) { wait_vmcnt<0>(); return; }
      // transfers per tile of this wave: the d = lw + 4 k below NDMA
      if (lw == 0) wait_vmcnt<(G::NDMA + 3) / 4>();
      else if (lw == 1) wait_vmcnt<(G::NDMA + 2) / 4>();
      else if (lw == 2) wait_vmcnt<(G::NDMA + 1) / 4>();
      else wait_vmcnt<G::NDMA / 4>();
    };
    int n = n0;
    if (n < N) issue(0);
    if (n + ipi < N) issue(1);
    wait_landed(n + ipi < N);                              // tile(n0) has landed
    thin_barrier();                                        // prologue barrier
    int buf = 0;
    for (; n < N; n += ipi) {
      // the compute waves read tile(n + ipi) next (stage buf + 1); stage buf + 2 was released by the last barrier
      const bool more = n + 2 * ipi < N;
      if (more) issue(buf >= 1 ? buf - 1 : 2);
      wait_landed(more);                                   // tile(n + ipi) has landed
      thin_barrier();
      buf = nxt(buf);
    }
    thin_barrier();                                        // final barrier: every tile is consumed
    thin_barrier();                                        // the reduction's barriers (the loaders take part: s_barrier counts
    thin_barrier();                                        // every wave of the workgroup)
    return;
  }

  // ------------------------------------------------------------------ compute waves: wave = small row sy0 + wv
  const int i16 = lane & 15, kq = lane >> 4;
  const int kh = i16 >> 2, kw = i16 & 3;
  // A (small side): pixel 4 t + kq of the row, channels 2 i16, 2 i16 + 1: float offset (wv * 32 + 4 t + kq) * 32 + 2 i16
  const int aoff = (wv * 32 + kq) * 32 + 2 * i16;          // + 128 t
  // B (big side), plane nt: row 2 wv + kh, column 2 (4 t + kq) - 1 + kw; lanes that would read column -1 (t = 0, kq = 0,
  // kw = 0) or column 64 (t = 7, kq = 3, kw = 3) read the plane's zero zone instead
  const int bmid = G::SM + (2 * wv + kh) * 64 + 2 * kq - 1 + kw;    // + 8 t
  const int bfirst = (kq == 0 && kw == 0) ? G::SM + G::ZZ : bmid;
  const int blast = (kq == 3 && kw == 3) ? G::SM + G::ZZ : bmid + 56;
  f32x4 acc[2][C];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < C; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 sumS = {0.f, 0.f};
  float sumB[C];
#pragma unroll
  for (int nt = 0; nt < C; ++nt) sumB[nt] = 0.f;
  thin_barrier();                                          // prologue barrier: tile(n0) is in stage 0

  // Operands: the NEXT unit's are read under the current unit's MFMAs, step for step (the barrier in front of a unit says
  // that the next tile has landed).  A tile is therefore read completely one unit BEFORE its MFMAs run: the loaders refill
  // the stage of tile u - 1 while unit u - 1 runs, which a shorter look-ahead (reading a tile during its own unit) would race with.
  struct Ops {                                             // operands of one unit
    f32x2 a[8];
    float b[8][C];
  };
  auto read_step = [&](int buf, Ops& o, int t) {
    const float* st = smem + buf * G::STAGE;
    o.a[t] = *reinterpret_cast<const f32x2*>(st + aoff + 128 * t);
#pragma unroll
    for (int nt = 0; nt < C; ++nt) {
      const int base = t == 0 ? bfirst : (t == 7 ? blast : bmid + 8 * t);
      o.b[t][nt] = st[base + nt * G::PLANE];
    }
  };
  // the 8 steps of the unit whose operands are in `oc`; every step refills its registers with the next unit's operands (stage
  // `nbuf`; stale and unused if there is none)
  auto unit = [&](Ops& oc, int nbuf) {
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): this unit's operands have returned (its tile is read)
    thin_barrier();                                        // ... and the next tile has landed
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (BIAS_BIG) {
#pragma unroll
        for (int nt = 0; nt < C; ++nt) sumB[nt] += oc.b[t][nt];
      } else {
        sumS += oc.a[t];
      }
#pragma unroll
      for (int nt = 0; nt < C; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          if (!(ABL & 2)) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(oc.a[t][mt], oc.b[t][nt], acc[mt][nt], 0, 0, 0);
      read_step(nbuf, oc, t);                              // into the registers this step has just freed
      __builtin_amdgcn_sched_barrier(0);                   // keep the steps in this order
    }
  };
  Ops oa;
  int n = n0, buf = 0;
  if (n < N) {
#pragma unroll
    for (int t = 0; t < 8; ++t) read_step(0, oa, t);
    for (; n < N; n += ipi) {
      unit(oa, nxt(buf));
      buf = nxt(buf);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  thin_barrier();                                          // final barrier: every tile is consumed, the stages are free

  // cross-wave reduction in a fixed order through LDS: red[wave][slot][lane], slot = (mt * C + nt) * 4 + r, then the two
  // (one + C) bias partials
  constexpr int NSLOT = G::NACC + 2 + C;
  float* red = smem;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < C; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wv * NSLOT + (mt * C + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
  red[(wv * NSLOT + G::NACC + 0) * 64 + lane] = sumS[0];
  red[(wv * NSLOT + G::NACC + 1) * 64 + lane] = sumS[1];
#pragma unroll
  for (int nt = 0; nt < C; ++nt) red[(wv * NSLOT + G::NACC + 2 + nt) * 64 + lane] = sumB[nt];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  thin_barrier();
  constexpr int STRIDE = NT32 * 1024 + 32 + NT32 * 32;     // k_wgrad_thin's partial block: [nt32][cs][32] + sumS[32] + sumB[nt32][32]
  float* wsw = ws + (long)blockIdx.x * STRIDE;
  const int t4 = tid;                                      // 256 compute threads
  auto wsum = [&](int sl, int ln) {
    return (red[(0 * NSLOT + sl) * 64 + ln] + red[(1 * NSLOT + sl) * 64 + ln]) + (red[(2 * NSLOT + sl) * 64 + ln] + red[(3 * NSLOT + sl) * 64 + ln]);
  };
  // D register r of lane (i16, kq), tile (mt, nt): row 4 kq + r -> cs = 2 (4 kq + r) + mt; column i16 -> tap i16 of channel nt
  for (int idx = t4; idx < NT32 * 1024; idx += 256) {
    const int nt32 = idx >> 10, cs = (idx >> 5) & 31, j = idx & 31;
    const int nidx = nt32 * 32 + j;                        // cb * 16 + tap
    float v = 0.f;
    if (nidx < 16 * C) {
      const int nt = nidx >> 4, tap = nidx & 15, mt = cs & 1, row = cs >> 1;
      v = wsum((mt * C + nt) * 4 + (row & 3), (row >> 2) * 16 + tap);
    }
    wsw[idx] = v;
  }
  if (t4 < 32) {                                           // sumS[cs]: channel 2 i16 + q of lanes (i16, kq = 0..3)
    const int i = t4 >> 1, q = t4 & 1;
    wsw[NT32 * 1024 + t4] = BIAS_BIG ? 0.f : (wsum(G::NACC + q, i) + wsum(G::NACC + q, 16 + i)) + (wsum(G::NACC + q, 32 + i) + wsum(G::NACC + q, 48 + i));
  }
  if (t4 < NT32 * 32) {                                    // sumB[column nidx]
    const int nidx = t4;
    float v = 0.f;
    if (BIAS_BIG && nidx < 16 * C) {
      const int nt = nidx >> 4, tap = nidx & 15;
      v = (wsum(G::NACC + 2 + nt, tap) + wsum(G::NACC + 2 + nt, 16 + tap)) + (wsum(G::NACC + 2 + nt, 32 + tap) + wsum(G::NACC + 2 + nt, 48 + tap));
    }
    wsw[NT32 * 1024 + 32 + nidx] = v;
  }
  thin_barrier();
}

// fp32 images only; writes gridDim.x = *grid_out partial blocks in k_wgrad_thin's layout; returns 1 if not covered
int launch_wgrad_thin_ws(const float* big, const float* small, float* ws, int bias_from_big, int N, int Cb, int* grid_out, hipStream_t s) {
  static const bool off = env_off("DVAE_THIN_WS");                   // A/B switch, debug builds only
  static const int min_n = env_int("DVAE_THIN_WS_MIN_N", 192);
  if (off || N < min_n || (Cb != 1 && Cb != 3)) return 1;
  if (((uintptr_t)big | (uintptr_t)small) & 15) return 1;
  // ONE workgroup per CU (the LDS of two would fit): alone the kernel takes the same 34 us either way, but two leave 6 KB of LDS
  // per CU and the small kernels of the other stream's critical path then wait for the whole launch (k_down32<4> ran 56 us
  // beside it); with one, the training step is 12 us shorter (profiles/r04_v44_thin_wgrad_grid.txt)
  static const int grid_dbg = env_int("DVAE_THIN_WGRAD_GRID", 256);  // debug builds: 512 = two workgroups per CU
  const int grid = grid_dbg == 512 ? 512 : 256;
  static_assert(512 <= WT_MAX_BLOCKS, "the partial buffer holds the grid");
#define DVAE_WTW(C, BB)                                                                                                    \
  do {                                                                                                                     \
    constexpr int lds = ThinWgGeo<C, BB>::TOTAL * 4;                                                                       \
    static DeviceOnce attr;                                                                                                \
    if (attr.first()) (void)hipFuncSetAttribute((const void*)k_wgrad_thin_ws<C, BB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((k_wgrad_thin_ws<C, BB>), dim3(grid), dim3(512), lds, s, big, small, ws, N);                        \
  } while (0)
  if (Cb == 1) { if (bias_from_big) DVAE_WTW(1, true); else DVAE_WTW(1, false); }
  else { if (bias_from_big) DVAE_WTW(3, true); else DVAE_WTW(3, false); }
#undef DVAE_WTW
  DVAE_CHECK_LAUNCH();
  *grid_out = grid;
  return 0;
}

}  // namespace dvae
