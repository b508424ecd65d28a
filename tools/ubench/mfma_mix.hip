// What does a second wave per SIMD cost a wave that only issues fp32 MFMAs?  (The three wave-specialised conv kernels lose
// ~15 % to their loader waves and 7-17 % to their own LDS operand reads: profiles/r02_run7_*, r02_run14_15_*.)
// One workgroup per CU, 512 threads: waves 0-3 = "compute" (one per SIMD), waves 4-7 = "helper" (one per SIMD).
// compute variants (per "unit" = 128 32x32x2 MFMAs or 256 16x16x4 MFMAs = 8192 matrix-pipe cycles):
//   c0  32x32x2, operands in registers
//   c1  32x32x2, 2 ds_read_b128 per 8 MFMAs (k_up32ws)            c2  32x32x2, 5 ds_read_b128 per 16 MFMAs (k_wgrad32ws)
//   c3  16x16x4, 6 ds_read_b128 per 16 MFMAs (k_down32ws)
// helper variants (per unit, sized like a loader wave of the conv kernels):
//   h0  nothing                         h1  11 x 4 ds_write_b32, conflict-free        h2  11 x 4 ds_write_b32, 4-way bank conflicts
//   h3  11 global_load_dwordx4 (fresh 43.5 KB per unit and workgroup) + s_waitcnt    h4  150 VALU (address-arithmetic stand-in)
//   h5  h1 + h3 + h4 (a whole loader)   h6  8 global_store_dwordx4 (32 KB per unit and workgroup)     h7  11 ds_write_b128, conflict-free
// prints the MFMA rate of the compute waves (TFLOP/s over the launch) for every pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// c4 / c5: k_down32ws<16>'s REAL operand addresses (swizzled A tile behind the 64 KB B image), one / two taps of look-ahead
template <int DEPTH>
__device__ __forceinline__ void compute_wave_down(const float* lds, int lane, int wv, int units, float* out, int tid) {
  f32x4 acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, kq = lane >> 4;
  const float* wl = lds;
  const float* bt = lds + 16384;
  f32x4 R[DEPTH + 1][6];
  auto rd = [&](int tap, int slot) {
    const int kh = tap >> 2, kw = tap & 3;
    const int r = 2 * wv + kh;
    const int par = kw & 1, cw = i16 + (kw >> 1);
    const float* arow = bt + ((r * 2 + par) * 17 + cw) * 32;
    const int sw = (cw >> 1) & 7;
    const float* brow = wl + (tap * 8) * 128 + i16 * 4;
    R[slot][0] = *(const f32x4*)(arow + ((kq ^ sw) << 2));
    R[slot][1] = *(const f32x4*)(arow + (((4 + kq) ^ sw) << 2));
    R[slot][2] = *(const f32x4*)(brow + kq * 128);
    R[slot][3] = *(const f32x4*)(brow + kq * 128 + 64);
    R[slot][4] = *(const f32x4*)(brow + (4 + kq) * 128);
    R[slot][5] = *(const f32x4*)(brow + (4 + kq) * 128 + 64);
  };
  for (int u = 0; u < units; ++u) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) rd(d, d);
    __builtin_amdgcn_sched_group_barrier(0x100, 6 * DEPTH, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int cur = t % (DEPTH + 1);
      if (t + DEPTH < 16) rd(t + DEPTH, (t + DEPTH) % (DEPTH + 1));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][0][j], R[cur][2][j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][0][j], R[cur][3][j], acc[4 + j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][1][j], R[cur][4][j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][1][j], R[cur][5][j], acc[4 + j], 0, 0, 0);
      if (t + DEPTH < 16) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 8; ++c) for (int e = 0; e < 4; ++e) s += acc[c][e];
  if (s == 12345.f) out[tid] = s;
}

template <int CM>
__device__ __forceinline__ void compute_wave(const float* lds, int lane, int units, float* out, int tid) {
  if (CM == 4) { compute_wave_down<1>(lds, lane, tid >> 6, units, out, tid); return; }
  if (CM == 5) { compute_wave_down<2>(lds, lane, tid >> 6, units, out, tid); return; }
  if (CM == 3) {
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = lds + lane * 4;
    f32x4 R[2][6];
    for (int q = 0; q < 6; ++q) R[0][q] = *(const f32x4*)(ap + q * 256);
    for (int u = 0; u < units; ++u) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int cur = t & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q) R[cur ^ 1][q] = *(const f32x4*)(ap + (((((t + 1) & 15) * 6 + q) * 256) & 0x3F00));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][0][j], R[cur][2][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][0][j], R[cur][3][j], acc[4 + j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][1][j], R[cur][4][j], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(R[cur][1][j], R[cur][5][j], acc[4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int e = 0; e < 4; ++e) s += acc[c][e];
    if (s == 12345.f) out[tid] = s;
    return;
  }
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  const float* ap = lds + lane * 4;
  if (CM == 0) {
    f32x4 a = *(const f32x4*)ap, b = *(const f32x4*)(ap + 256);
    for (int u = 0; u < units; ++u) {
#pragma unroll
      for (int g = 0; g < 32; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[j], 0, 0, 0);
    }
  } else if (CM == 1) {
    f32x4 A[2][2];
    A[0][0] = *(const f32x4*)ap; A[0][1] = *(const f32x4*)(ap + 256);
    for (int u = 0; u < units; ++u) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {           // 8 MFMAs per group, 2 reads
        const int cur = g & 1;
        A[cur ^ 1][0] = *(const f32x4*)(ap + (((g + 1) & 15) * 2) * 256);
        A[cur ^ 1][1] = *(const f32x4*)(ap + (((g + 1) & 15) * 2 + 1) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][0][j], A[cur][1][j], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][1][j], A[cur][0][j], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
    }
  } else {
    f32x4 A[2][5];
    for (int q = 0; q < 5; ++q) A[0][q] = *(const f32x4*)(ap + q * 256);
    for (int u = 0; u < units; ++u) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {            // 16 MFMAs per group, 5 reads
        const int cur = g & 1;
#pragma unroll
        for (int q = 0; q < 5; ++q) A[cur ^ 1][q] = *(const f32x4*)(ap + (((g + 1) & 7) * 5 + q) * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][0][j], A[cur][1][j], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][0][j], A[cur][2][j], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][0][j], A[cur][3][j], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[cur][0][j], A[cur][4][j], acc[3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      }
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  if (s == 12345.f) out[tid] = s;
}

// helper work of ONE unit; `pace` = approximate matrix-pipe cycles of a unit, used to spread the work (s_sleep between
// slices) so that the helper is active all along the compute waves' unit, as a real loader wave is
template <int HM>
__device__ __forceinline__ void helper_unit(float* lds_w, const float* gsrc, float* gdst, int ht, int u, int wg, f32x4 (&hold)[11],
                                            float& vacc) {
  constexpr int TILE = 2720 * 4;                           // floats per unit and workgroup (43.5 KB)
  if (HM == 1 || HM == 5) {
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) lds_w[(k * 4 + c) * 256 + ht] = hold[k][c];        // consecutive lanes, consecutive banks
  }
  if (HM == 2) {
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c)          // lanes l, l+16, l+32, l+48 of a wave: 64 floats apart = the same bank
        lds_w[(k * 4 + c) * 256 + (ht & 15) + (ht >> 6) * 16 + ((ht >> 4) & 3) * 64] = hold[k][c];
  }
  if (HM == 7) {
#pragma unroll
    for (int k = 0; k < 11; ++k) *(f32x4*)(lds_w + (k * 256 + ht) * 4) = hold[k];
  }
  if (HM == 3 || HM == 5) {
    const float* src = gsrc + ((long)(u * 256 + wg) % 4096) * TILE;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const int s = k * 256 + ht;
      if (HM == 3) vacc += hold[k][1];                     // consume the previous unit's tile (a loader does so at its LDS store)
      hold[k] = *(const f32x4*)(src + (s < 2720 ? s : 0) * 4);
    }
  }
  if (HM == 4 || HM == 5) {
    float v = vacc;
#pragma unroll
    for (int k = 0; k < 150; ++k) v = v * 1.0001f + (float)k;
    vacc = v;
  }
  if (HM == 8) {                                           // 150 scalar ALU instructions
    int x = wg + u;                                          // uniform
#pragma unroll
    for (int k = 0; k < 150; ++k) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x));
    if (x == 123456789) vacc += 1.f;
  }
  if (HM == 9) {                                           // 150 VALU in 6 independent chains
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = vacc + (float)c;
#pragma unroll
    for (int k = 0; k < 25; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] = v[c] * 1.0001f + (float)k;
    vacc = ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
  }
  if (HM == 6) {
    float* dst = gdst + ((long)(u * 256 + wg) % 4096) * 8192;
#pragma unroll
    for (int k = 0; k < 8; ++k) *(f32x4*)(dst + (k * 256 + ht) * 4) = hold[k];
  }
}

template <int CM, int HM>
__global__ __launch_bounds__(768) void k(float* out, const float* in, const float* gsrc, float* gdst, int units, long long* tms, int prio) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 16384 + 11264; e += blockDim.x) lds[e] = in[e & 1023];
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long t0 = wall_clock64();                     // 100 MHz constant clock
  if (wv < 4) {
    if (prio == 0) __builtin_amdgcn_s_setprio(1);          // the conv kernels' setting: compute waves above the loaders
    else if (prio == 3) __builtin_amdgcn_s_setprio(0);
    compute_wave<CM>(lds, lane, units, out, tid);
    if (lane == 0) tms[blockIdx.x * 8 + wv] = wall_clock64() - t0;
  } else {
    if (prio == 1) __builtin_amdgcn_s_setprio(3);          // helpers ABOVE the compute waves
    else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    const int ht = (tid - 256) & 255;                      // (with 768 threads: two helper waves per SIMD doing the same work)
    f32x4 hold[11];
    for (int q = 0; q < 11; ++q) hold[q] = f32x4{1.f + ht, 2.f, 3.f, 4.f};
    float vacc = (float)ht;
    float* lds_w = lds + 16384;                            // the helpers write their own 44 KB: no data race with the readers
    if (HM != 0) {
      for (int u = 0; u < units; ++u) {
        helper_unit<HM>(lds_w, gsrc, gdst, ht, u, blockIdx.x, hold, vacc);
        // a loader waits for the unit barrier; stand-in: sleep about 3/4 of a unit (8192 cycles) between two units' work
        for (int z = 0; z < 12; ++z) __builtin_amdgcn_s_sleep(8);
      }
    }
    float s = vacc;
    for (int q = 0; q < 11; ++q) s += hold[q][0] + hold[q][3];
    if (s == 12345.f) out[tid] = s;
    if (lane == 0 && wv < 8) tms[blockIdx.x * 8 + wv] = wall_clock64() - t0;
  }
}

static long long* g_tms;
static int g_threads = 512;
static int g_prio = 0;   // 0: compute prio 1 / helper 0 (the conv kernels); 1: compute default 0 / helper 3; 2: nobody sets a priority; 3: compute 0 (explicit) / helper 3
struct Res { float tf_compute, helper_over_compute; };
template <int CM, int HM> static Res bench(float* out, float* in, float* gsrc, float* gdst) {
  const int units = 64;
  const int ldsb = (16384 + 11264) * 4;
  (void)hipFuncSetAttribute((const void*)k<CM, HM>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  static long long h[2048];
  double tc = 0, th = 0;
  for (int r = 0; r < 4; ++r) {
    hipLaunchKernelGGL((k<CM, HM>), dim3(256), dim3(g_threads), ldsb, 0, out, in, gsrc, gdst, units, g_tms, g_prio);
    (void)hipDeviceSynchronize();
    if (r == 0) continue;                                  // warm-up
    (void)hipMemcpy(h, g_tms, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? tc : th) += (double)h[b * 8 + w];
  }
  tc /= 3.0 * 256 * 4; th /= 3.0 * 256 * 4;               // mean ticks of a compute / helper wave (100 MHz)
  const double flops_wave = (double)units * 128 * 4096.0;  // per compute wave
  Res r;
  r.tf_compute = (float)(flops_wave * 1024 / (tc * 1e-8) / 1e12);     // 1024 compute waves on the chip
  r.helper_over_compute = (float)(th / tc);
  return r;
}

template <int CM> static void row(const char* name, float* out, float* in, float* gsrc, float* gdst) {
  Res r[10] = {bench<CM, 0>(out, in, gsrc, gdst), bench<CM, 1>(out, in, gsrc, gdst), bench<CM, 2>(out, in, gsrc, gdst),
               bench<CM, 7>(out, in, gsrc, gdst), bench<CM, 3>(out, in, gsrc, gdst), bench<CM, 4>(out, in, gsrc, gdst),
               bench<CM, 9>(out, in, gsrc, gdst), bench<CM, 8>(out, in, gsrc, gdst), bench<CM, 5>(out, in, gsrc, gdst),
               bench<CM, 6>(out, in, gsrc, gdst)};
  const char* hn[10] = {"h0", "h1", "h2", "h7", "h3", "h4", "h9", "h8", "h5", "h6"};
  printf("%-22s", name);
  for (int q = 0; q < 10; ++q) printf(" %s %5.1f (%4.2f)", hn[q], r[q].tf_compute, r[q].helper_over_compute);
  printf("\n");
  fflush(stdout);
}

int main() {
  float *out, *in, *gsrc, *gdst;
  hipMalloc(&out, 4096); hipMalloc(&in, 4096);
  hipMalloc(&gsrc, (size_t)4096 * 2720 * 16); hipMalloc(&gdst, (size_t)4096 * 8192 * 4);
  hipMemset(gsrc, 0, (size_t)4096 * 2720 * 16);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 7919) % 1000) / 1000.f - 0.5f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  (void)hipMalloc(&g_tms, 2048 * sizeof(long long));
  printf("MFMA rate of the compute waves from the device clock, TFLOP/s chip-wide (in brackets: helper-wave time / compute-wave time)\n");
  printf("helpers: h0 none, h1 44 ds_write_b32, h2 same with bank conflicts, h7 11 ds_write_b128, h3 11 global loads 16 B, h4 150 VALU, h5 h1+h3+h4, h6 8 global stores 16 B (per unit)\n");
  printf("         h9 150 VALU in 6 independent chains, h8 150 SALU\n");
  g_prio = 0; g_threads = 512;
  printf("-- 512 threads; priorities compute 1, helper 0; c4 / c5 = k_down32ws<16>'s real LDS addresses, 1 / 2 taps of look-ahead\n");
  row<3>("c3 16x16x4 6rd/16mfma", out, in, gsrc, gdst);
  row<4>("c4 down16 addresses", out, in, gsrc, gdst);
  row<5>("c5 down16, 2 taps ahead", out, in, gsrc, gdst);
  return 0;
}
