// Shared helpers for the gfx950 kernels of libdvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/dvae_hip.h"

namespace dvae {

void set_error(const char* fmt, ...);

#define DVAE_CHECK_ARG(cond)                                                            \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      dvae::set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond);        \
      return -1;                                                                        \
    }                                                                                   \
  } while (0)

#define DVAE_CHECK_LAUNCH()                                                             \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      dvae::set_error("%s:%d: launch failed: %s", __FILE__, __LINE__,                   \
                      hipGetErrorString(e__));                                          \
      return -2;                                                                        \
    }                                                                                   \
  } while (0)

// element strides of a [N,C,H,W] tensor stored as NCHW or NHWC
struct Strides {
  long n, c, h, w;
};
static inline Strides make_strides(int layout, int C, int H, int W) {
  Strides s;
  if (layout == DVAE_NHWC) {
    s.n = (long)H * W * C; s.h = (long)W * C; s.w = C; s.c = 1;
  } else {
    s.n = (long)C * H * W; s.c = (long)H * W; s.h = W; s.w = 1;
  }
  return s;
}

// "once per device" flag.  Function attributes such as MaxDynamicSharedMemorySize belong to the (function, device) pair: a
// process that drives several GPUs has to set them on each (one process per GPU, the data-parallel layout, never notices).
struct DeviceOnce {
  bool done[32] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return true;
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// per-element reconstruction likelihood (losses.py:394-449) and its gradient: p = sigmoid output,
// x = target.  Returns the un-normalised loss term; *gl = dL/dlogit, *gr = dL/dp (un-scaled).
__device__ __forceinline__ float recon_elem(float p, float x, int dist, float* gl, float* gr) {
  const float q1 = 1.f - p;
  const float qq = q1 * p;                       // sigmoid backward factor (1-y)*y
  float term;
  if (dist == DVAE_REC_BERNOULLI) {
    // ATen binary_cross_entropy: (x-1)*max(log1p(-p),-100) - x*max(log(p),-100); backward
    // (p-x)/max((1-p)*p, 1e-12).  log1p(-p) is evaluated as log(1-p): 1-p is exact for p >= 0.5 and
    // off by <= 6e-8 (absolute) below; __logf = v_log_f32 * ln2 (~1 ulp), not for denormal inputs.
    const float lp = p < 1e-30f ? logf(p) : __logf(p);
    term = (x - 1.f) * fmaxf(__logf(q1), -100.f) - x * fmaxf(lp, -100.f);
    const float d = p - x;
    *gl = qq >= 1e-12f ? d : d * 1e12f * qq;     // ((p-x)/max(qq,1e-12)) * qq
    *gr = d / fmaxf(qq, 1e-12f);
  } else if (dist == DVAE_REC_GAUSSIAN) {
    const float d = p * 255.f - x * 255.f;       // mse(255p, 255x, sum) / 255
    term = d * d / 255.f;
    *gr = 2.f * d;
    *gl = *gr * qq;
  } else {
    const float d = p - x;                        // 3 * l1(sum)
    term = 3.f * fabsf(d);
    *gr = d > 0.f ? 3.f : (d < 0.f ? -3.f : 0.f);
    *gl = *gr * qq;
  }
  return term;
}

// torch.sigmoid as ATen's fp32 kernel computes it (1 / (1 + exp(-v)): IEEE add, IEEE division) on the hardware
// transcendentals.  With e = exp(-|v|), s = fl(1 + e), r = v_rcp_f32(s) (1 ulp):
//   v >= 0:  1 / s = 1 - q / s with q = s - 1 (exact: the ROUNDED e, which is all ATen's sum keeps of it); one FMA rounds
//            1 - q r onto p's grid (spacing 2^-24 below 1): the correctly rounded quotient except for near-ties.  That is the
//            point: F.binary_cross_entropy takes log(1 - p) of THIS p, so from v ~ 8 upwards the reference's likelihood term
//            is -log(k 2^-24) for an integer k, and p == 1 (term clamped to 100) from v ~ 16.64 -- losses.py:430;
//   v <  0:  e / s (no cancellation on that side);
//   v < DVAE_SIGMOID_ZERO_BELOW:  ATen's exp(-v) overflows fp32, so its p is exactly 0.
#define DVAE_SIGMOID_ZERO_BELOW (-88.72283935546875f)
__device__ __forceinline__ float sigmoid_aten(float v) {
  const float e = __expf(-fabsf(v));
  const float s = 1.f + e;
  const float r = __builtin_amdgcn_rcpf(s);
  float p = v >= 0.f ? fmaf(1.f - s, r, 1.f) : e * r;
  if (v < -87.f)   // v_exp_f32 flushes results below 2^-126: ATen's p is a denormal down to its overflow point, then 0
    p = v < DVAE_SIGMOID_ZERO_BELOW ? 0.f : ldexpf(__expf(v + 17.328679513998633f), -25);
  return p;
}

// sigmoid + Bernoulli likelihood term + dL/dlogit from the LOGIT v, equal to what the reference computes in fp32
// (losses.py:430: F.binary_cross_entropy(sigmoid(v), x) = (x - 1) max(log(1 - p), -100) - x max(log p, -100) on the fp32 p).
// With e, s, r, p as in sigmoid_aten and ls = log(1 + e) (= log(s) + (e - q) / s, accurate also where s rounds to 1):
//   v >= 0:  -log(1 - p) is taken from ATen's p itself (1 - p is exact; 0 -> +inf -> the clamp's 100) -- NOT v + ls, which
//            is what it would be in exact arithmetic and differs from the reference by up to 4 % per element from v ~ 8 on
//            and by (100 - v) (1 - x) once p rounds to 1 (round 5's form);  -log p = ls;
//   v <  0:  -log(1 - p) = ls, -log p = ls - v (no cancellation in the reference either), 100 where ATen's p is 0.
// Four transcendentals per output (exp, rcp, 2 x log).  *gl = dL/dlogit: (p - x), scaled down where (1 - p) p < 1e-12
// (ATen's backward clamp, as recon_elem's).
__device__ __forceinline__ float sigmoid_bce_logit(float v, float x, float* p_out, float* gl) {
  const float e = __expf(-fabsf(v));
  const float s = 1.f + e;
  const float r = __builtin_amdgcn_rcpf(s);
  const float q = s - 1.f;
  const bool pos = v >= 0.f, zero = v < DVAE_SIGMOID_ZERO_BELOW;
  float p = pos ? fmaf(-q, r, 1.f) : e * r;
  if (__builtin_expect(v < -87.f, 0))                      // as sigmoid_aten: a denormal down to ATen's overflow point, then 0
    p = zero ? 0.f : ldexpf(__expf(v + 17.328679513998633f), -25);
  // v_log_f32 (log2, ~1 ulp) x ln 2 -- __logf() expands to a denormal-safe, extended-precision sequence of 12 instructions
  const float ls = fmaf(__builtin_amdgcn_logf(s), 0.69314718056f, (e - q) * r);
  const float om = 1.f - p;
  const float l1 = fminf(__builtin_amdgcn_logf(om) * -0.69314718056f, 100.f);
  const float t0 = pos ? l1 : ls;                          // the term at x = 0
  const float dx = pos ? ls - l1 : (zero ? 100.f : -v);    // + x times this
  const float qq = om * p, d = p - x;
  *gl = qq >= 1e-12f ? d : d * 1e12f * qq;
  *p_out = p;
  return fmaf(x, dx, t0);
}

// ---- launchers implemented in the individual .hip files ---------------------------------
// "down": big[N,Cb,2Hs,2Ws] -> small[N,Cs,Hs,Ws]  (Conv2d fwd, ConvTranspose2d dgrad)
// "up"  : small -> big                           (ConvTranspose2d fwd, Conv2d dgrad)
// "wg"  : dw[Cs,Cb,4,4] = sum small (*) big       (both wgrads)
// weights are always indexed w[cs][cb][kh][kw].
struct ConvArgs {
  const float* big; int big_layout;
  const float* small; int small_layout;
  const float* w; const float* bias; const float* mask;
  float* out; int out_layout;
  int N, Cb, Cs, Hs, Ws;  // Hs,Ws: SMALL spatial dims (big is 2Hs x 2Ws)
  int act;
  int w_staged;           // 1: `w` is the pre-staged LDS weight image of dvae_stage_weights (tuned 32-channel kernels only)
  // ReLU masks as bit planes (one uint32 per pixel of a 32-channel NHWC activation, bit c = [a[p][c] > 0]):
  const uint32_t* mask_bits = nullptr;   // consumed instead of `mask` by the input-gradient kernels that support it
  uint32_t* out_bits = nullptr;          // emitted next to `out` by the forward kernels that support it
};
int launch_down_generic(const ConvArgs& a, hipStream_t s);
int launch_up_generic(const ConvArgs& a, hipStream_t s);
int launch_wgrad_generic(const float* big, int big_layout, const float* small, int small_layout,
                         float* dw, float* db, int bias_from_big, int N, int Cb, int Cs, int Hs, int Ws, float* ws,
                         size_t ws_floats, hipStream_t s);
// MFMA paths (32 <-> 32 channels, NHWC, Hs == Ws in {4,8,16}); return 1 if not applicable
int launch_down_mfma32(const ConvArgs& a, hipStream_t s);
int launch_up_mfma32(const ConvArgs& a, hipStream_t s);
int launch_down_mfma32_dma(const ConvArgs& a, hipStream_t s);  // conv_down_dma.hip: Hs in {8,16}, NHWC -> NHWC
int launch_wgrad_mfma32_ws(const float* big, const float* small, float* dw, float* db, int bias_from_big, int N, int Hs,
                           float* ws, hipStream_t s);    // wave-specialised, transposed LDS tiles (conv_wgrad_ws.hip): Hs in {8,16}
int launch_wgrad32_reduce(const float* ws, float* dw, float* db, int bias_from_big, int nblk, hipStream_t s, int N);
int launch_up_mfma32_ws(const ConvArgs& a, hipStream_t s);      // wave-specialised (conv_up_ws.hip): Hs in {8,16}, NHWC
int launch_wgrad_mfma32(const float* big, const float* small, float* dw, float* db, int bias_from_big,
                        int N, int Hs, float* ws, hipStream_t s, int small_nchw = 0);
// thin paths (Cb in {1,3}, Cs == 32, big NCHW 64x64 / small NHWC 32x32)
int launch_down_thin(const ConvArgs& a, hipStream_t s);
int launch_down_thin_ws(const ConvArgs& a, hipStream_t s);   // conv_thin_ws.hip: fp32 images, wave-specialised; 1 if not covered
int launch_up_thin(const ConvArgs& a, hipStream_t s);
// fused convT3 + sigmoid + reconstruction loss (+ dL/dlogit); returns 1 if the shape is not covered
int launch_up_thin_recon(const ConvArgs& a, const float* target, float* g, int dist, const float* coef,
                         float* partials, hipStream_t s);
int launch_wgrad_thin_ws(const float* big, const float* small, float* ws, int bias_from_big, int N, int Cb, int* grid_out,
                         hipStream_t s);   // conv_thin_ws.hip: partial sums only; 1 if not covered
int launch_wgrad_thin(const float* big, const float* small, float* dw, float* db, int bias_from_big,
                      int N, int Cb, int Hs, float* ws, hipStream_t s);
// uint8 input image x[N,C,64,64] (NCHW), converted on the fly with ToTensor's float(v)/255; return 1 if C is not 1 or 3
int launch_down_thin_u8(const uint8_t* x, const float* w, const float* bias, float* out, uint32_t* out_bits, int N, int C, int act,
                        hipStream_t s);
int launch_up_thin_recon_u8(const ConvArgs& a, const uint8_t* target, float* g, int dist, const float* coef,
                            float* partials, hipStream_t s);
int launch_wgrad_thin_u8(const uint8_t* x, const float* small, float* dw, float* db, int N, int C, float* ws, hipStream_t s);
int launch_kl_normal_bwd(const float* g, const float* mu, const float* lv, float* dmu, float* dlv, int B, int D, hipStream_t s);
int launch_reduce_sum(const float* src, long n, float scale, float* dst, hipStream_t s);
int launch_u8_to_f32(const uint8_t* src, float* dst, long n, hipStream_t s);

size_t wgrad32_ws_floats();
size_t wgrad_thin_ws_floats();
int launch_relayout(const float* src, int src_layout, float* dst, int N, int C, int H, int W, hipStream_t s);

int launch_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, float* ws,
                      size_t ws_floats, hipStream_t s);
int launch_linear_dgrad(const float* dy, const float* w, const float* x_act, int act, float* dx, int M, int K, int N,
                        float* ws, size_t ws_floats, hipStream_t s);
int launch_linear_wgrad(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, float* ws,
                        size_t ws_floats, hipStream_t s);

// gemm_dma.hip: LDS-DMA GEMMs for the large (discriminator) shapes; true if the launch was taken
bool try_gdma(bool b_jfast, const float* a, long lda, const float* b, long ldb, float* c, long ldc, int M, int N, int Kc,
              const float* bias, int act, const float* mask, int mask_act, hipStream_t s);
bool try_gdma_wgrad(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, hipStream_t s);

// linear_narrow.hip: the discriminator's last layer (1000 -> 2), forward and input gradient, as streaming passes; true if taken
bool try_narrow_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, hipStream_t s);
bool try_narrow_dgrad(const float* dy, const float* w, const float* x_act, int act, float* dx, int M, int K, int N,
                      hipStream_t s);

size_t latent_entropy_ws_floats(long N, int D, int S);
int launch_latent_entropy(const float* z_ds, const float* mean, const float* logvar, long N, int D, int S, float* ws,
                          float* H, hipStream_t s);
int launch_linear_wgrad_grouped(const dvae_linear_wgrad_desc* d, int n, hipStream_t s);
int fc_chain_rows(int n);
int launch_fc_chain_fwd(const dvae_fc_chain_fwd_args* a, hipStream_t s);
int launch_fc_chain_bwd(const dvae_fc_chain_bwd_args* a, hipStream_t s);
int reparam_kl_blocks(int B);
int launch_kl_finish(float* kl_dim, int kl_blocks, const float* coef, int D, hipStream_t s);
int launch_stage_weights(const dvae_conv_image_desc* conv, int n_conv, const dvae_fc_image_desc* fc, int n_fc,
                         const dvae_thin_image_desc* thin, float* coef, const float* coef_vals, hipStream_t s);
// tap (kh * 4 + kw) and output channel of entry `idx` of a contracted channel's pair record (k_up_thin_pk, conv_thin.hip;
// the staging kernel's gather)
static __host__ __device__ __forceinline__ int thin_pair_source(int idx, int C, int* cb) {
  const int taps[16] = {5, 6, 9, 10, 13, 14, 1, 2, 7, 11, 4, 8, 0, 3, 12, 15};
  if (C == 3 && idx < 32) {
    const int a = idx >> 1, t = a >> 2, cls = a & 3;      // tap-major: 16 consecutive floats feed all four classes' chains
    const int py = cls >> 1, px = cls & 1, ty = t >> 1, tx = t & 1;
    *cb = idx & 1;
    return (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
  }
  *cb = C - 1;
  return taps[C == 3 ? idx - 32 : idx];
}
int launch_up_thin_mm(const float* small, const float* wimg, const float* bias, const void* target, int target_u8, float* out,
                      float* g, int dist, const float* coef, float* partials, int N, int act, hipStream_t s);
int launch_up_thin_staged(const float* small, const float* wrec, const float* bias, const void* target, int target_u8,
                          float* out, float* g, int dist, const float* coef, float* partials, int N, int C, int act,
                          hipStream_t s);

int launch_reparam_kl_fwd(const float* ml, const float* eps, float* mu, float* logvar, float* z, float* kl_dim,
                          const float* coef, int B, int D, hipStream_t s);
int launch_reparam_kl_bwd(const float* dz, const float* dz2, const float* dz3, const float* dmu_x, const float* dlv_x, const float* mu, const float* logvar,
                          const float* eps, const float* scal, const float* coef, float* dml, int B, int D,
                          hipStream_t s);
int launch_recon_loss(const float* recon, const float* target, long n, int dist, const float* coef, float* partials,
                      float* g, int wrt_logit, hipStream_t s);
int launch_sigmoid_bwd(const float* gy, const float* y, float* out, long n, hipStream_t s);
int launch_btcvae_fwd(const float* z, const float* mu, const float* lv, int Bg, int D, int row0, int Bl, int is_mss,
                      const float* log_w, float* tmp, float* rowstats, hipStream_t s);
int launch_btcvae_bwd(const float* z, const float* mu, const float* lv, const float* rowstats, int Bg, int D, int row0,
                      int Bl, int is_mss, const float* log_w, const float* coef, const float* tmp, float* dz, float* dmu,
                      float* dlv, hipStream_t s);
// latent dimensions above DVAE_MAX_D (latent_wide.hip)
int launch_reparam_kl_fwd_wide(const float* ml, const float* eps, float* mu, float* logvar, float* z, float* kl_dim,
                               const float* coef, int B, int D, hipStream_t s);
int launch_btcvae_fwd_wide(const float* z, const float* mu, const float* lv, int Bg, int D, int row0, int Bl, int is_mss,
                           const float* log_w, float* tmp, float* rowstats, hipStream_t s);
int launch_btcvae_bwd_wide(const float* z, const float* mu, const float* lv, const float* rowstats, int Bg, int D, int row0,
                           int Bl, int is_mss, const float* log_w, const float* coef, const float* tmp, float* dz, float* dmu,
                           float* dlv, hipStream_t s);
int launch_permute_dims(const float* z, const int64_t* perm, float* out, int B, int D, hipStream_t s);
int launch_disc_losses(const float* lg, int Bh, const float* coef, float* sums, float* g_dtc, float* g_tc,
                       hipStream_t s);
int launch_loss_pack(const float* rec_partials, const float* kl_dim, int D, const float* rowstats, int Bl,
                     const float* disc_sums, float* packed, hipStream_t s);
int launch_loss_finalize(int kind, const float* packed, int D, int Bg, const float* coef, float* scal, hipStream_t s);
int launch_loss_epilogue(int kind, const float* rec_partials, const float* kl_dim, int kl_blocks, int D, const float* rowstats,
                         int Bl, const float* disc_sums, int Bg, const float* coef, float* packed, float* scal,
                         hipStream_t s);
int launch_set_coef(float* coef, const float* v, hipStream_t s);
int launch_adam(const dvae_adam_tensor* ts, int nt, float step_new, double lr, double beta1, double beta2, double eps,
                double weight_decay, hipStream_t s);
int launch_add(const float* a, const float* b, float* out, long n, hipStream_t s);
int launch_axpby(float* out, const float* a, float alpha, const float* b, float beta, long n, hipStream_t s);
int launch_swap_outer(const float* src, float* dst, int A, int Bn, long inner, hipStream_t s);

bool use_generic_only();  // DVAE_FORCE_GENERIC=1 (on-device reference path of the parity tests, still HIP)

// A/B and timing-ablation switches (DVAE_UP_WS, DVAE_DMA_ABLATE, DVAE_FCC_VARIANT, ...) and the kernel variants they
// select exist only in a library built with -DDVAE_DEBUG_SWITCHES (python build.py --debug); the shipped
// library compiles them out.  A switch is ON only for the value "1" (integers: atoi of the value).
#ifdef DVAE_DEBUG_SWITCHES
static inline bool env_on(const char* name) { const char* e = getenv(name); return e && e[0] == '1' && e[1] == 0; }
static inline bool env_off(const char* name) { const char* e = getenv(name); return e && e[0] == '0' && e[1] == 0; }
static inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr bool env_on(const char*) { return false; }
static inline constexpr bool env_off(const char*) { return false; }
static inline constexpr int env_int(const char*, int dflt) { return dflt; }
#endif

}  // namespace dvae
