// extern "C" surface of libdvae_hip.so (see include/dvae_hip.h): argument checking and
// dispatch between the tuned gfx950 kernels and the shape-generic HIP kernels.
#include <mutex>
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"

namespace dvae {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool use_generic_only() {
  const char* e = getenv("DVAE_FORCE_GENERIC");
  return e && e[0] == '1';
}

static int check_layout(int l) { return l == DVAE_NCHW || l == DVAE_NHWC; }

// big[N,Cb,2Hs,2Ws] -> small[N,Cs,Hs,Ws]
static int run_down(const ConvArgs& a, hipStream_t s) {
  if (!use_generic_only()) {
    int r = launch_down_mfma32(a, s);
    if (r <= 0) return r;
    r = launch_down_thin(a, s);
    if (r <= 0) return r;
  }
  return launch_down_generic(a, s);
}
static int run_up(const ConvArgs& a, hipStream_t s) {
  if (!use_generic_only()) {
    int r = 1;
    static const bool no_ws = env_off("DVAE_UP_WS");        // debug builds: DVAE_UP_WS=0 -> k_up32 for every geometry (A/B)
    if (!no_ws) {
      r = launch_up_mfma32_ws(a, s);
      if (r <= 0) return r;
    }
    r = launch_up_mfma32(a, s);
    if (r <= 0) return r;
    r = launch_up_thin(a, s);
    if (r <= 0) return r;
  }
  return launch_up_generic(a, s);
}
static int run_wgrad(const float* big, int big_layout, const float* small, int small_layout, float* dw, float* db,
                     int bias_from_big, int N, int Cb, int Cs, int Hs, int Ws, float* ws, hipStream_t s) {
  // the small side of the 4x4 end of the conv stack may be NCHW (= the FC stack's (c,h,w) order)
  if (!use_generic_only() && ws != nullptr && Hs == Ws && Cs == 32 && Cb == 32 && big_layout == DVAE_NHWC && Hs == 4 &&
      small_layout == DVAE_NCHW)
    return launch_wgrad_mfma32(big, small, dw, db, bias_from_big, N, Hs, ws, s, 1);
  if (!use_generic_only() && ws != nullptr && Hs == Ws && Cs == 32 && small_layout == DVAE_NHWC) {
    if (Cb == 32 && big_layout == DVAE_NHWC && (Hs == 4 || Hs == 8 || Hs == 16))
      return launch_wgrad_mfma32(big, small, dw, db, bias_from_big, N, Hs, ws, s, 0);
    if ((Cb == 1 || Cb == 3) && Hs == 32 && big_layout == DVAE_NCHW)
      return launch_wgrad_thin(big, small, dw, db, bias_from_big, N, Cb, Hs, ws, s);
  }
  return launch_wgrad_generic(big, big_layout, small, small_layout, dw, db, bias_from_big, N, Cb, Cs, Hs, Ws, ws,
                              dvae_conv_wgrad_ws_floats(), s);
}

}  // namespace dvae

using namespace dvae;

extern "C" {

int dvae_version(void) { return DVAE_VERSION; }
const char* dvae_last_error(void) { return g_err; }

int dvae_conv4s2_fwd(const float* x, int x_layout, const float* w, const float* b, float* y, int y_layout, int N,
                     int Cin, int H, int W, int Cout, int act, void* stream) {
  DVAE_CHECK_ARG(x && w && y && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (H % 2 == 0) && (W % 2 == 0));
  DVAE_CHECK_ARG(check_layout(x_layout) && check_layout(y_layout));
  ConvArgs a{x, x_layout, nullptr, 0, w, b, nullptr, y, y_layout, N, Cin, Cout, H / 2, W / 2, act};
  return run_down(a, (hipStream_t)stream);
}

int dvae_conv4s2_dgrad(const float* dy, int dy_layout, const float* w, const float* x_act, float* dx, int dx_layout,
                       int N, int Cin, int H, int W, int Cout, void* stream) {
  DVAE_CHECK_ARG(dy && w && dx && N > 0 && Cin > 0 && Cout > 0 && (H % 2 == 0) && (W % 2 == 0));
  DVAE_CHECK_ARG(check_layout(dy_layout) && check_layout(dx_layout));
  // conv weight w[Cout,Cin,4,4] = w[cs][cb]: small = dy (Cout channels), big = dx (Cin channels)
  ConvArgs a{nullptr, 0, dy, dy_layout, w, nullptr, x_act, dx, dx_layout, N, Cin, Cout, H / 2, W / 2, DVAE_ACT_NONE};
  return run_up(a, (hipStream_t)stream);
}

int dvae_conv4s2_wgrad(const float* x, int x_layout, const float* dy, int dy_layout, float* dw, float* db, int N,
                       int Cin, int H, int W, int Cout, float* ws, void* stream) {
  DVAE_CHECK_ARG(x && dy && dw && N > 0 && (H % 2 == 0) && (W % 2 == 0));
  const int rc = run_wgrad(x, x_layout, dy, dy_layout, dw, db, /*bias_from_big=*/0, N, Cin, Cout, H / 2, W / 2, ws,
                   (hipStream_t)stream);
  return rc;
}

int dvae_convT4s2_fwd(const float* x, int x_layout, const float* w, const float* b, float* y, int y_layout, int N,
                      int Cin, int H, int W, int Cout, int act, void* stream) {
  DVAE_CHECK_ARG(x && w && y && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
  DVAE_CHECK_ARG(check_layout(x_layout) && check_layout(y_layout));
  // convT weight w[Cin,Cout,4,4] = w[cs][cb]: small = x (Cin channels), big = y (Cout channels)
  ConvArgs a{nullptr, 0, x, x_layout, w, b, nullptr, y, y_layout, N, Cout, Cin, H, W, act};
  return run_up(a, (hipStream_t)stream);
}

int dvae_convT4s2_dgrad(const float* dy, int dy_layout, const float* w, const float* x_act, float* dx, int dx_layout,
                        int N, int Cin, int H, int W, int Cout, void* stream) {
  DVAE_CHECK_ARG(dy && w && dx && N > 0 && Cin > 0 && Cout > 0);
  DVAE_CHECK_ARG(check_layout(dy_layout) && check_layout(dx_layout));
  ConvArgs a{dy, dy_layout, nullptr, 0, w, nullptr, x_act, dx, dx_layout, N, Cout, Cin, H, W, DVAE_ACT_NONE};
  return run_down(a, (hipStream_t)stream);
}

int dvae_convT4s2_wgrad(const float* x, int x_layout, const float* dy, int dy_layout, float* dw, float* db, int N,
                        int Cin, int H, int W, int Cout, float* ws, void* stream) {
  DVAE_CHECK_ARG(x && dy && dw && N > 0);
  const int rc = run_wgrad(dy, dy_layout, x, x_layout, dw, db, /*bias_from_big=*/1, N, Cout, Cin, H, W, ws,
                   (hipStream_t)stream);
  return rc;
}

int dvae_convT4s2_sigmoid_recon_fwd(const float* x, int x_layout, const float* w, const float* b, const float* target,
                                    float* recon, float* g, int dist, const float* coef, float* partials, int N,
                                    int Cin, int H, int W, int Cout, void* stream) {
  DVAE_CHECK_ARG(x && w && target && recon && g && coef && partials && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
  DVAE_CHECK_ARG(check_layout(x_layout));
  DVAE_CHECK_ARG(dist == DVAE_REC_BERNOULLI || dist == DVAE_REC_GAUSSIAN || dist == DVAE_REC_LAPLACE);
  ConvArgs a{nullptr, 0, x, x_layout, w, b, nullptr, recon, DVAE_NCHW, N, Cout, Cin, H, W, DVAE_ACT_SIGMOID};
  static const bool two_pass = env_on("DVAE_RECON_TWO_PASS");   // A/B switch (debug builds)
  if (!use_generic_only() && !two_pass) {
    int r = launch_up_thin_recon(a, target, g, dist, coef, partials, (hipStream_t)stream);
    if (r <= 0) return r;
  }
  int r = run_up(a, (hipStream_t)stream);       // shapes outside the fused kernel: two passes
  if (r) return r;
  const long n = (long)N * Cout * 4 * H * W;
  DVAE_CHECK_ARG(n % 4 == 0);
  return launch_recon_loss(recon, target, n, dist, coef, partials, g, 1, (hipStream_t)stream);
}

// ---- uint8 input pipeline (utils/datasets.py:204-213,282-291: ToTensor fused into the consumers) ----------
static bool u8_fused_shape(int C, int H, int W, int Cout) { return (C == 1 || C == 3) && H == 64 && W == 64 && Cout == 32; }

int dvae_u8_to_f32(const uint8_t* src, float* dst, long n, void* stream) {
  DVAE_CHECK_ARG(src && dst && n > 0 && (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 15) == 0));
  return launch_u8_to_f32(src, dst, n, (hipStream_t)stream);
}

int dvae_u8_fused_supported(int C, int H, int W) { return u8_fused_shape(C, H, W, 32) ? 1 : 0; }

int dvae_conv4s2_fwd_u8(const uint8_t* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W,
                        int Cout, int act, void* stream) {
  DVAE_CHECK_ARG(x && w && y && N > 0);
  DVAE_CHECK_ARG(u8_fused_shape(Cin, H, W, Cout));
  DVAE_CHECK_ARG(act == DVAE_ACT_NONE || act == DVAE_ACT_RELU);
  return launch_down_thin_u8(x, w, b, y, nullptr, N, Cin, act, (hipStream_t)stream);
}

int dvae_conv4s2_wgrad_u8(const uint8_t* x, const float* dy, float* dw, float* db, int N, int Cin, int H, int W,
                          int Cout, float* ws, void* stream) {
  DVAE_CHECK_ARG(x && dy && dw && ws && N > 0);
  DVAE_CHECK_ARG(u8_fused_shape(Cin, H, W, Cout));
  return launch_wgrad_thin_u8(x, dy, dw, db, N, Cin, ws, (hipStream_t)stream);
}

int dvae_convT4s2_sigmoid_recon_fwd_u8(const float* x, const float* w, const float* b, const uint8_t* target,
                                       float* recon, float* g, int dist, const float* coef, float* partials, int N,
                                       int Cin, int H, int W, int Cout, void* stream) {
  DVAE_CHECK_ARG(x && w && target && recon && g && coef && partials && N > 0);
  DVAE_CHECK_ARG(Cin == 32 && u8_fused_shape(Cout, 2 * H, 2 * W, 32));
  DVAE_CHECK_ARG(dist == DVAE_REC_BERNOULLI || dist == DVAE_REC_GAUSSIAN || dist == DVAE_REC_LAPLACE);
  ConvArgs a{nullptr, 0, x, DVAE_NHWC, w, b, nullptr, recon, DVAE_NCHW, N, Cout, Cin, H, W, DVAE_ACT_SIGMOID};
  return launch_up_thin_recon_u8(a, target, g, dist, coef, partials, (hipStream_t)stream);
}

size_t dvae_conv_wgrad_ws_floats(void) {
  size_t a = wgrad32_ws_floats(), b = wgrad_thin_ws_floats();
  return a > b ? a : b;
}

int dvae_relayout(const float* src, int src_layout, float* dst, int N, int C, int H, int W, void* stream) {
  DVAE_CHECK_ARG(src && dst && check_layout(src_layout) && N > 0 && C > 0 && H > 0 && W > 0);
  return launch_relayout(src, src_layout, dst, N, C, H, W, (hipStream_t)stream);
}

int dvae_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int act, float* ws,
                    void* stream) {
  DVAE_CHECK_ARG(x && w && y && M > 0 && K > 0 && N > 0);
  DVAE_CHECK_ARG(act == DVAE_ACT_NONE || act == DVAE_ACT_RELU || act == DVAE_ACT_LEAKY02);
  return launch_linear_fwd(x, w, b, y, M, K, N, act, ws, dvae_conv_wgrad_ws_floats(), (hipStream_t)stream);
}

int dvae_linear_dgrad(const float* dy, const float* w, const float* x_act, int act, float* dx, int M, int K, int N,
                      float* ws, void* stream) {
  DVAE_CHECK_ARG(dy && w && dx && M > 0 && K > 0 && N > 0);
  return launch_linear_dgrad(dy, w, x_act, act, dx, M, K, N, ws, dvae_conv_wgrad_ws_floats(), (hipStream_t)stream);
}

int dvae_linear_wgrad(const float* x, const float* dy, float* dw, float* db, int M, int K, int N, float* ws,
                      void* stream) {
  DVAE_CHECK_ARG(x && dy && dw && M > 0 && K > 0 && N > 0);
  return launch_linear_wgrad(x, dy, dw, db, M, K, N, ws, dvae_conv_wgrad_ws_floats(), (hipStream_t)stream);
}

int dvae_linear_wgrad_grouped(const dvae_linear_wgrad_desc* descs, int n, void* stream) {
  DVAE_CHECK_ARG(descs && n >= 1 && n <= DVAE_FCW_MAX);
  for (int q = 0; q < n; ++q)
    DVAE_CHECK_ARG(descs[q].x && descs[q].dy && descs[q].dw && descs[q].M > 0 && descs[q].K > 0 && descs[q].N > 0);
  return launch_linear_wgrad_grouped(descs, n, (hipStream_t)stream);
}

size_t dvae_latent_entropy_ws_floats(long N, int D, int S) {
  if (N <= 0 || D <= 0 || S <= 0) return 0;
  return latent_entropy_ws_floats(N, D, S);
}

int dvae_latent_entropy(const float* z_ds, const float* mean, const float* logvar, long N, int D, int S, float* ws,
                        float* H, void* stream) {
  DVAE_CHECK_ARG(z_ds && mean && logvar && ws && H && N > 0 && D > 0 && D <= 65535 && S > 0);
  return launch_latent_entropy(z_ds, mean, logvar, N, D, S, ws, H, (hipStream_t)stream);
}

int dvae_reparam_kl_fwd(const float* ml, const float* eps, float* mu, float* logvar, float* z, float* kl_dim,
                        const float* coef, int B, int D, void* stream) {
  DVAE_CHECK_ARG(ml && mu && logvar && z && B > 0 && D > 0);    // kl_dim without coef: partials only
  DVAE_CHECK_ARG(D <= DVAE_MAX_D || ((!kl_dim || coef) && D <= DVAE_KL_FLOATS));   // above: final values only
  return launch_reparam_kl_fwd(ml, eps, mu, logvar, z, kl_dim, coef, B, D, (hipStream_t)stream);
}

int dvae_reparam_kl_bwd(const float* dz, const float* dz2, const float* dz3, const float* dmu_x, const float* dlv_x,
                        const float* mu, const float* logvar, const float* eps, const float* scal, const float* coef,
                        float* dml, int B, int D, void* stream) {
  DVAE_CHECK_ARG(mu && logvar && scal && coef && dml && B > 0 && D > 0);
  return launch_reparam_kl_bwd(dz, dz2, dz3, dmu_x, dlv_x, mu, logvar, eps, scal, coef, dml, B, D, (hipStream_t)stream);
}

int dvae_kl_normal_bwd(const float* g_dim, const float* mu, const float* logvar, float* dmu, float* dlogvar, int B, int D,
                       void* stream) {
  DVAE_CHECK_ARG(g_dim && mu && logvar && dmu && dlogvar && B > 0 && D > 0);
  return launch_kl_normal_bwd(g_dim, mu, logvar, dmu, dlogvar, B, D, (hipStream_t)stream);
}

int dvae_reduce_sum(const float* src, long n, float scale, float* dst, void* stream) {
  DVAE_CHECK_ARG(src && dst && n > 0);
  return launch_reduce_sum(src, n, scale, dst, (hipStream_t)stream);
}

int dvae_recon_loss(const float* recon, const float* target, long n, int dist, const float* coef, float* partials,
                    float* g, int wrt_logit, void* stream) {
  DVAE_CHECK_ARG(recon && target && coef && partials && n > 0 && (n % 4 == 0));
  DVAE_CHECK_ARG(dist == DVAE_REC_BERNOULLI || dist == DVAE_REC_GAUSSIAN || dist == DVAE_REC_LAPLACE);
  return launch_recon_loss(recon, target, n, dist, coef, partials, g, wrt_logit, (hipStream_t)stream);
}

int dvae_sigmoid_bwd(const float* grad_y, const float* y, float* out, long n, void* stream) {
  DVAE_CHECK_ARG(grad_y && y && out && n > 0);
  return launch_sigmoid_bwd(grad_y, y, out, n, (hipStream_t)stream);
}

int dvae_btcvae_fwd(const float* z, const float* mu, const float* logvar, int Bg, int D, int row0, int Bl, int is_mss,
                    const float* log_w, float* tmp, float* rowstats, void* stream) {
  DVAE_CHECK_ARG(z && mu && logvar && tmp && rowstats && Bg > 1 && D >= 1 && row0 >= 0 && Bl > 0 && row0 + Bl <= Bg);
  DVAE_CHECK_ARG(D <= DVAE_WIDE_MAX_D /* one row of D log-sum-exps in LDS (latent_wide.hip: k_tcw_rowstats) */);
  DVAE_CHECK_ARG(!is_mss || log_w);
  return launch_btcvae_fwd(z, mu, logvar, Bg, D, row0, Bl, is_mss, log_w, tmp, rowstats, (hipStream_t)stream);
}

int dvae_btcvae_bwd(const float* z, const float* mu, const float* logvar, const float* rowstats, int Bg, int D,
                    int row0, int Bl, int is_mss, const float* log_w, const float* coef, const float* tmp, float* dz,
                    float* dmu_all, float* dlv_all, void* stream) {
  DVAE_CHECK_ARG(z && mu && logvar && rowstats && coef && tmp && dz && dmu_all && dlv_all && Bg > 1 && D >= 1);
  DVAE_CHECK_ARG(D <= DVAE_WIDE_MAX_D /* one row of D log-sum-exps in LDS (latent_wide.hip) */);
  DVAE_CHECK_ARG(row0 >= 0 && Bl > 0 && row0 + Bl <= Bg && (!is_mss || log_w));
  return launch_btcvae_bwd(z, mu, logvar, rowstats, Bg, D, row0, Bl, is_mss, log_w, coef, tmp, dz, dmu_all, dlv_all,
                           (hipStream_t)stream);
}

int dvae_permute_dims(const float* z, const int64_t* perm, float* out, int B, int D, void* stream) {
  DVAE_CHECK_ARG(z && perm && out && B > 0 && D > 0);
  return launch_permute_dims(z, perm, out, B, D, (hipStream_t)stream);
}

int dvae_disc_losses(const float* dlogits, int Bh, const float* coef, float* sums, float* g_dtc, float* g_tc,
                     void* stream) {
  DVAE_CHECK_ARG(dlogits && coef && sums && g_dtc && Bh > 0);
  return launch_disc_losses(dlogits, Bh, coef, sums, g_dtc, g_tc, (hipStream_t)stream);
}

int dvae_loss_pack(const float* rec_partials, const float* kl_dim, int D, const float* rowstats, int Bl,
                   const float* disc_sums, float* packed, void* stream) {
  DVAE_CHECK_ARG(rec_partials && packed && D >= 0);
  return launch_loss_pack(rec_partials, kl_dim, D, rowstats, Bl, disc_sums, packed, (hipStream_t)stream);
}

int dvae_loss_epilogue(int kind, const float* rec_partials, const float* kl_dim, int kl_blocks, int D, const float* rowstats,
                       int Bl, const float* disc_sums, int Bg, const float* coef, float* packed, float* scal,
                       void* stream) {
  DVAE_CHECK_ARG(rec_partials && packed && coef && D >= 0 && Bl >= 0 && Bg > 0);
  DVAE_CHECK_ARG(kl_blocks >= 0 && kl_blocks <= DVAE_KL_MAX_BLOCKS && (D <= DVAE_MAX_D || kl_blocks == 0));
  DVAE_CHECK_ARG(kind >= DVAE_LOSS_BETAH && kind <= DVAE_LOSS_FACTOR);
  return launch_loss_epilogue(kind, rec_partials, kl_dim, kl_blocks, D, rowstats, Bl, disc_sums, Bg, coef, packed, scal,
                              (hipStream_t)stream);
}

int dvae_reparam_kl_blocks(int B) { return B > 0 ? reparam_kl_blocks(B) : 0; }

int dvae_kl_finish(float* kl_dim, int kl_blocks, const float* coef, int D, void* stream) {
  DVAE_CHECK_ARG(kl_dim && coef && kl_blocks > 0 && kl_blocks <= DVAE_KL_MAX_BLOCKS && D > 0 && D <= DVAE_MAX_D);
  return launch_kl_finish(kl_dim, kl_blocks, coef, D, (hipStream_t)stream);
}

// ---- per-step weight staging, the tuned 32-channel kernels on pre-staged weights, the FC chain ----------------------
int dvae_stage_weights(const dvae_conv_image_desc* conv, int n_conv, const dvae_fc_image_desc* fc, int n_fc,
                       const dvae_thin_image_desc* thin, float* coef, const float* coef_vals, void* stream) {
  DVAE_CHECK_ARG(n_conv >= 0 && n_conv <= DVAE_STAGE_MAX_CONV && n_fc >= 0 && n_fc <= DVAE_STAGE_MAX_FC);
  DVAE_CHECK_ARG((n_conv == 0 || conv) && (n_fc == 0 || fc));
  for (int q = 0; q < n_conv; ++q) DVAE_CHECK_ARG(conv[q].w);
  for (int q = 0; q < n_fc; ++q) DVAE_CHECK_ARG(fc[q].w && fc[q].N > 0 && fc[q].K > 0);
  if (thin) DVAE_CHECK_ARG(thin->w && (thin->C == 1 || thin->C == 3));
  return launch_stage_weights(conv, n_conv, fc, n_fc, thin, coef, coef_vals, (hipStream_t)stream);
}

int dvae_convT3_fwd_staged(const float* x, const float* img_pairs, const float* b, const void* target, int target_is_u8,
                           float* recon, float* g, int dist, const float* coef, float* partials, int N, int C,
                           void* stream) {
  DVAE_CHECK_ARG(x && img_pairs && recon && N > 0 && (C == 1 || C == 3));
  if (target) {
    DVAE_CHECK_ARG(g && coef && partials);
    DVAE_CHECK_ARG(dist == DVAE_REC_BERNOULLI || dist == DVAE_REC_GAUSSIAN || dist == DVAE_REC_LAPLACE);
  }
  return launch_up_thin_staged(x, img_pairs, b, target, target_is_u8, recon, g, dist, coef, partials, N, C, DVAE_ACT_SIGMOID,
                               (hipStream_t)stream);
}

int dvae_conv32_down(const float* big, const float* img_down, const float* bias, const float* mask, float* out,
                     int out_layout, int N, int Hs, int act, void* stream) {
  DVAE_CHECK_ARG(big && img_down && out && N > 0 && (Hs == 4 || Hs == 8 || Hs == 16) && check_layout(out_layout));
  DVAE_CHECK_ARG(out_layout == DVAE_NHWC || Hs == 4);
  DVAE_CHECK_ARG(act == DVAE_ACT_NONE || act == DVAE_ACT_RELU);
  ConvArgs a{big, DVAE_NHWC, nullptr, 0, img_down, bias, mask, out, out_layout, N, 32, 32, Hs, Hs, act, 1};
  const int r = launch_down_mfma32(a, (hipStream_t)stream);
  if (r > 0) { set_error("dvae_conv32_down: geometry not covered"); return -1; }
  return r;
}

int dvae_conv32_up(const float* small, int small_layout, const float* img_up, const float* bias, const float* mask,
                   float* out, int N, int Hs, int act, void* stream) {
  DVAE_CHECK_ARG(small && img_up && out && N > 0 && (Hs == 4 || Hs == 8 || Hs == 16) && check_layout(small_layout));
  DVAE_CHECK_ARG(small_layout == DVAE_NHWC || Hs == 4);
  DVAE_CHECK_ARG(act == DVAE_ACT_NONE || act == DVAE_ACT_RELU);
  ConvArgs a{nullptr, 0, small, small_layout, img_up, bias, mask, out, DVAE_NHWC, N, 32, 32, Hs, Hs, act, 1};
  int r = launch_up_mfma32_ws(a, (hipStream_t)stream);
  if (r > 0) r = launch_up_mfma32(a, (hipStream_t)stream);
  if (r > 0) { set_error("dvae_conv32_up: geometry not covered"); return -1; }
  return r;
}

// ---- ReLU masks as bit planes (include/dvae_hip.h) ---------------------------------------------------------------------
int dvae_conv1_fwd_bits(const void* x, int x_is_u8, const float* w, const float* b, float* y, uint32_t* y_bits, int N,
                        int Cin, void* stream) {
  DVAE_CHECK_ARG(x && w && y && y_bits && N > 0 && (Cin == 1 || Cin == 3));
  if (x_is_u8) return launch_down_thin_u8((const uint8_t*)x, w, b, y, y_bits, N, Cin, DVAE_ACT_RELU, (hipStream_t)stream);
  ConvArgs a{(const float*)x, DVAE_NCHW, nullptr, 0, w, b, nullptr, y, DVAE_NHWC, N, Cin, 32, 32, 32, DVAE_ACT_RELU, 0};
  a.out_bits = y_bits;
  const int r = launch_down_thin(a, (hipStream_t)stream);
  if (r > 0) { set_error("dvae_conv1_fwd_bits: geometry not covered"); return -1; }
  return r;
}

int dvae_conv32_up_bits(const float* small, const float* img_up, const float* bias, const uint32_t* mask_bits, float* out,
                        uint32_t* out_bits, int N, int act, void* stream) {
  DVAE_CHECK_ARG(small && img_up && out && N > 0 && (mask_bits != nullptr) != (out_bits != nullptr));
  DVAE_CHECK_ARG(act == DVAE_ACT_NONE || act == DVAE_ACT_RELU);
  DVAE_CHECK_ARG(!out_bits || act == DVAE_ACT_RELU);
  ConvArgs a{nullptr, 0, small, DVAE_NHWC, img_up, bias, nullptr, out, DVAE_NHWC, N, 32, 32, 16, 16, act, 1};
  a.mask_bits = mask_bits;
  a.out_bits = out_bits;
  const int r = launch_up_mfma32_ws(a, (hipStream_t)stream);
  if (r > 0) { set_error("dvae_conv32_up_bits: geometry not covered"); return -1; }
  return r;
}

int dvae_convT3_dgrad_bits(const float* dy, const float* w, const uint32_t* x_act_bits, float* dx, int N, int Cout,
                           void* stream) {
  DVAE_CHECK_ARG(dy && w && x_act_bits && dx && N > 0 && (Cout == 1 || Cout == 3));
  ConvArgs a{dy, DVAE_NCHW, nullptr, 0, w, nullptr, nullptr, dx, DVAE_NHWC, N, Cout, 32, 32, 32, DVAE_ACT_NONE, 0};
  a.mask_bits = x_act_bits;
  const int r = launch_down_thin(a, (hipStream_t)stream);
  if (r > 0) { set_error("dvae_convT3_dgrad_bits: geometry not covered"); return -1; }
  return r;
}

int dvae_fc_chain_fwd(const dvae_fc_chain_fwd_args* a, void* stream) {
  DVAE_CHECK_ARG(a && a->a_flat && a->w_e1 && a->w_e2 && a->w_ml && a->b_e1 && a->b_e2 && a->b_ml);
  DVAE_CHECK_ARG(a->h1 && a->h2 && a->ml && a->mu && a->logvar && a->z);
  DVAE_CHECK_ARG(a->D >= 1 && a->D <= DVAE_MAX_D && a->n_enc > 0 && a->n_enc <= 8 * DVAE_KL_MAX_BLOCKS);
  DVAE_CHECK_ARG(a->n_kl >= 0 && a->n_kl <= a->n_enc && a->n_dec >= 0 && a->n_dec <= a->n_enc);
  if (a->n_dec > 0) DVAE_CHECK_ARG(a->w_d1 && a->w_d2 && a->w_d3 && a->b_d1 && a->b_d2 && a->b_d3 && a->d1 && a->d2 && a->d3);
  if (a->conv_in) DVAE_CHECK_ARG(a->conv_w && a->conv_b && ((((uintptr_t)a->conv_in | (uintptr_t)a->conv_w) & 15) == 0));
  if (a->convT_w) DVAE_CHECK_ARG(a->conv_in && a->convT_b && a->convT_out && a->n_dec > 0 && (((uintptr_t)a->convT_w & 15) == 0));
  return launch_fc_chain_fwd(a, (hipStream_t)stream);
}

int dvae_fc_chain_rows(int n) { return fc_chain_rows(n); }

int dvae_fc_chain_bwd(const dvae_fc_chain_bwd_args* a, void* stream) {
  DVAE_CHECK_ARG(a && a->gd3 && a->w_d3 && a->w_d2 && a->w_d1 && a->w_ml && a->w_e2 && a->w_e1);
  DVAE_CHECK_ARG(a->d2 && a->d1 && a->h2 && a->h1 && a->a_flat && a->mu && a->logvar && a->scal && a->coef);
  DVAE_CHECK_ARG(a->gd2 && a->gd1 && a->dml && a->gh2 && a->gh1 && a->ga_flat);
  DVAE_CHECK_ARG(a->D >= 1 && a->D <= DVAE_MAX_D && a->n > 0);
  if (a->convT_gout) DVAE_CHECK_ARG(a->convT_w && a->d3 && ((((uintptr_t)a->convT_gout | (uintptr_t)a->convT_w) & 15) == 0));
  if (a->conv_w) DVAE_CHECK_ARG(a->convT_gout && a->conv_act && a->conv_gin && (((uintptr_t)a->conv_w & 15) == 0));
  return launch_fc_chain_bwd(a, (hipStream_t)stream);
}

int dvae_loss_finalize(int kind, const float* packed, int D, int Bg, const float* coef, float* scal, void* stream) {
  DVAE_CHECK_ARG(packed && coef && scal && D >= 0 && Bg > 0);
  DVAE_CHECK_ARG(kind >= DVAE_LOSS_BETAH && kind <= DVAE_LOSS_FACTOR);
  return launch_loss_finalize(kind, packed, D, Bg, coef, scal, (hipStream_t)stream);
}

int dvae_set_coef(float* coef, float c0, float c1, float c2, float c3, float c4, float c5, float c6, float c7,
                  void* stream) {
  DVAE_CHECK_ARG(coef);
  const float v[8] = {c0, c1, c2, c3, c4, c5, c6, c7};
  return launch_set_coef(coef, v, (hipStream_t)stream);
}

int dvae_stream_create(void** stream) {
  DVAE_CHECK_ARG(stream);
  hipStream_t s = nullptr;
  // (lowest-priority streams for the engine -- the caller's kernels first -- measured the same or 0.5-1.5 % slower at 64 .. 1024
  // images: profiles/r06_s2_prio1.txt; round 4 measured a HIGH-priority side stream: the same)
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    set_error("dvae_stream_create: %s", hipGetErrorString(hipGetLastError()));
    return -2;
  }
  *stream = (void*)s;
  return 0;
}

int dvae_stream_order(void* earlier, void* later) {
  // events are re-used round-robin: 256 of them outlive any window of outstanding fork / join pairs of an iteration (~20);
  // per device (an event belongs to the device it was created on)
  constexpr int NEV = 256, NDEV = 32;
  static hipEvent_t pool[NDEV][NEV];
  static int made[NDEV] = {}, next[NDEV] = {};
  static std::mutex mu;                       // pool creation and the round-robin cursor: callers may come from several host threads
  std::lock_guard<std::mutex> lock(mu);
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= NDEV) { set_error("dvae_stream_order: no current device"); return -2; }
  if (!made[d]) {
    for (int k = 0; k < NEV; ++k)
      if (hipEventCreateWithFlags(&pool[d][k], hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) {
        set_error("dvae_stream_order: hipEventCreateWithFlags failed");
        return -2;
      }
    made[d] = 1;
  }
  hipEvent_t ev = pool[d][next[d]];
  next[d] = (next[d] + 1) % NEV;
  if (hipEventRecord(ev, (hipStream_t)earlier) != hipSuccess || hipStreamWaitEvent((hipStream_t)later, ev, 0) != hipSuccess) {
    set_error("dvae_stream_order: %s", hipGetErrorString(hipGetLastError()));
    return -2;
  }
  return 0;
}

static hipEvent_t* event_slot(int slot, const char* who) {
  constexpr int NDEV = 32;
  static hipEvent_t pool[NDEV][DVAE_EVENT_SLOTS];
  static int made[NDEV] = {};
  static std::mutex mu;                       // lazy pool creation: callers may come from several host threads
  std::lock_guard<std::mutex> lock(mu);
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= NDEV) { set_error("%s: no current device", who); return nullptr; }
  if (slot < 0 || slot >= DVAE_EVENT_SLOTS) { set_error("%s: slot %d outside [0, %d)", who, slot, DVAE_EVENT_SLOTS); return nullptr; }
  if (!made[d]) {
    for (int k = 0; k < DVAE_EVENT_SLOTS; ++k)
      if (hipEventCreateWithFlags(&pool[d][k], hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) {
        set_error("%s: hipEventCreateWithFlags failed", who);
        return nullptr;
      }
    made[d] = 1;
  }
  return &pool[d][slot];
}

int dvae_event_record(int slot, void* stream) {
  hipEvent_t* ev = event_slot(slot, "dvae_event_record");
  if (!ev) return -2;
  if (hipEventRecord(*ev, (hipStream_t)stream) != hipSuccess) { set_error("dvae_event_record: %s", hipGetErrorString(hipGetLastError())); return -2; }
  return 0;
}

int dvae_event_wait(int slot, void* stream) {
  hipEvent_t* ev = event_slot(slot, "dvae_event_wait");
  if (!ev) return -2;
  if (hipStreamWaitEvent((hipStream_t)stream, *ev, 0) != hipSuccess) { set_error("dvae_event_wait: %s", hipGetErrorString(hipGetLastError())); return -2; }
  return 0;
}

int dvae_add(const float* a, const float* b, float* out, long n, void* stream) {
  DVAE_CHECK_ARG(a && b && out && n > 0);
  return launch_add(a, b, out, n, (hipStream_t)stream);
}

int dvae_adam_step(const dvae_adam_tensor* tensors, int nt, float step_new, double lr, double beta1, double beta2,
                   double eps, double weight_decay, void* stream) {
  DVAE_CHECK_ARG(tensors && nt > 0 && step_new >= 1.f && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1.);
  DVAE_CHECK_ARG(eps >= 0. && weight_decay >= 0.);
  for (int i = 0; i < nt; ++i) DVAE_CHECK_ARG(tensors[i].p && tensors[i].g && tensors[i].m && tensors[i].v && tensors[i].n >= 0);
  return launch_adam(tensors, nt, step_new, lr, beta1, beta2, eps, weight_decay, (hipStream_t)stream);
}

int dvae_axpby(float* out, const float* a, float alpha, const float* b, float beta, long n, void* stream) {
  DVAE_CHECK_ARG(out && a && n > 0);
  return launch_axpby(out, a, alpha, b, beta, n, (hipStream_t)stream);
}

int dvae_swap_outer(const float* src, float* dst, int A, int Bn, long inner, void* stream) {
  DVAE_CHECK_ARG(src && dst && src != dst && A > 0 && Bn > 0 && inner > 0 && inner < (1L << 31) && (long)A * Bn < (1L << 31));
  return launch_swap_outer(src, dst, A, Bn, inner, (hipStream_t)stream);
}

}  // extern "C"
