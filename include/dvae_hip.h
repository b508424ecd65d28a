/*
 * libdvae_hip.so -- C-ABI of the MI355X (gfx950) kernels behind the disvae training step.
 *
 * The reference (YannDubs/disentangling-vae) is pure Python on torch.nn and has NO native
 * FFI for this path: every entry point below replaces the ATen call(s) that the cited
 * reference line reaches through torch.nn / torch.nn.functional / autograd.  The host side
 * (disentangling-vae_amd/disvae_amd) binds them with ctypes; see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to fp32 (int64 where stated); sizes are element counts;
 *   - `stream` is a hipStream_t passed as void*; calls only ENQUEUE work (no allocation, no
 *     synchronisation, no ownership transfer); workspace is caller-provided;
 *   - return 0 on success, <0 on invalid argument / launch error (text via dvae_last_error());
 *   - re-entrant per stream; safe to capture into a hipGraph.
 *   - layouts: DVAE_NCHW (reference / API boundary) or DVAE_NHWC (engine-internal, 32-channel
 *     activations); weights always keep the reference's state_dict shapes.
 */
#ifndef DVAE_HIP_H
#define DVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVAE_VERSION 109

/* Latent dimensions.  The reference's --latent-dim is any integer (main.py:81); every experiment of hyperparam.ini uses 10.
 * The FUSED kernels (the FC chain, the register-resident beta-TCVAE estimator, the 16-wide KL / scalar records) cover
 * 1 <= D <= DVAE_MAX_D, D = 10 fully unrolled.  Above that the same entry points take any D (the estimator: up to
 * DVAE_WIDE_MAX_D) and run run-time-D kernels
 * (csrc/latent_wide.hip; the FC layers go one launch each through dvae_linear_*): only dvae_fc_chain_* and the partial-block
 * form of the KL (dvae_reparam_kl_fwd without coef, dvae_kl_finish, kl_blocks > 0) stay limited to DVAE_MAX_D.  Buffers whose
 * size depends on D beyond that point are sized by the macros below ("wide" layouts):                              */
#define DVAE_MAX_D 16
#define DVAE_WIDE_MAX_D 12288   /* the run-time-D estimator keeps one row of D log-sum-exps in LDS: 48 KB of the 64 KB a launch
                                 * gets without an opt-in; dvae_btcvae_fwd / _bwd refuse more with a message */
#define DVAE_BTCVAE_MAX_D DVAE_MAX_D   /* (kept for callers of version <= 107: the estimator's fused kernels) */
#define DVAE_ROWSTATS 32        /* floats per row of the estimator's row statistics: 4 + D used */
/* row stride of `rowstats` */
#define DVAE_ROWSTATS_STRIDE(D) ((D) <= DVAE_MAX_D ? DVAE_ROWSTATS : (((D) + 4 + 3) & ~3))
/* floats of the estimator's `tmp`: [3][D][Bg] column constants; above DVAE_MAX_D followed by the [Bl][Bg] joint log-densities */
#define DVAE_BTCVAE_TMP_FLOATS(Bg, Bl, D) \
  ((size_t)3 * (D) * (Bg) + ((D) <= DVAE_MAX_D ? (size_t)0 : (size_t)(Bl) * (Bg)))
/* floats of `packed` / `scal`; above DVAE_MAX_D the per-dimension KL values follow the 32 fixed slots: packed[32 + d], scal[32 + d]
 * (the 16 narrow KL slots stay zero) */
#define DVAE_WIDE_KL0 32
#define DVAE_NPACK_D(D) ((D) <= DVAE_MAX_D ? 32 : 32 + (D))
#define DVAE_NSCAL_D(D) ((D) <= DVAE_MAX_D ? 32 : 32 + (D))

enum { DVAE_NCHW = 0, DVAE_NHWC = 1 };
enum { DVAE_ACT_NONE = 0, DVAE_ACT_RELU = 1, DVAE_ACT_LEAKY02 = 2, DVAE_ACT_SIGMOID = 3 };
enum { DVAE_REC_BERNOULLI = 0, DVAE_REC_GAUSSIAN = 1, DVAE_REC_LAPLACE = 2 };
enum { DVAE_LOSS_BETAH = 0, DVAE_LOSS_BETAB = 1, DVAE_LOSS_BTCVAE = 2, DVAE_LOSS_FACTOR = 3 };

/* scalar slots written by dvae_loss_finalize (float[DVAE_NSCAL]) */
enum {
  DVAE_S_LOSS = 0, DVAE_S_REC = 1, DVAE_S_KL = 2, DVAE_S_KL0 = 3 /* ..3+D-1 for D <= DVAE_MAX_D; above: DVAE_WIDE_KL0 + d */,
  DVAE_S_MI = 19, DVAE_S_TC = 20, DVAE_S_DWKL = 21, DVAE_S_KLW = 22, DVAE_S_DTC = 23,
  DVAE_NSCAL = 32
};
/* coefficient slots read by the loss kernels (float[DVAE_NCOEF], host-written every step) */
enum {
  DVAE_C_INV_B = 0,   /* 1 / (reconstruction+KL batch denominator) */
  DVAE_C_ANNEAL = 1,  /* linear_annealing value of this step */
  DVAE_C_BETA = 2,    /* betaH beta | betaB gamma | btcvae beta | factor gamma */
  DVAE_C_ALPHA = 3,   /* btcvae alpha */
  DVAE_C_GAMMA = 4,   /* btcvae gamma */
  DVAE_C_CAP = 5,     /* betaB capacity C of this step */
  DVAE_NCOEF = 8
};

int dvae_version(void);
const char* dvae_last_error(void);

/* ---- Conv2d(k=4,s=2,p=1): encoders.py:54-60,73-77 (nn.Conv2d + torch.relu) -------------
 * x[N,Cin,H,W] -> y[N,Cout,H/2,W/2], w[Cout,Cin,4,4], y = act(conv(x,w)+b).               */
int dvae_conv4s2_fwd(const float* x, int x_layout, const float* w, const float* b, float* y,
                     int y_layout, int N, int Cin, int H, int W, int Cout, int act, void* stream);
/* autograd of the above (training.py:157 loss.backward): dy is the gradient w.r.t. the
 * PRE-activation output; dx = conv_dgrad(dy,w) (* [x_act > 0] when x_act != NULL, i.e. the
 * ReLU backward of the producing layer fused in).  H,W are the dims of x.                   */
int dvae_conv4s2_dgrad(const float* dy, int dy_layout, const float* w, const float* x_act, float* dx,
                       int dx_layout, int N, int Cin, int H, int W, int Cout, void* stream);
/* dw[Cout,Cin,4,4] = sum x (*) dy ; db[Cout] = sum dy (db may be NULL).
 * ws: workspace of at least dvae_conv_wgrad_ws_floats() floats.                             */
int dvae_conv4s2_wgrad(const float* x, int x_layout, const float* dy, int dy_layout, float* dw,
                       float* db, int N, int Cin, int H, int W, int Cout, float* ws, void* stream);

/* ---- ConvTranspose2d(k=4,s=2,p=1): decoders.py:57-65,77-82 ------------------------------
 * x[N,Cin,H,W] -> y[N,Cout,2H,2W], w[Cin,Cout,4,4]; act in {none, relu, sigmoid}.          */
int dvae_convT4s2_fwd(const float* x, int x_layout, const float* w, const float* b, float* y,
                      int y_layout, int N, int Cin, int H, int W, int Cout, int act, void* stream);
/* dx[N,Cin,H,W] from dy[N,Cout,2H,2W] (pre-activation grad), fused ReLU mask like above.   */
int dvae_convT4s2_dgrad(const float* dy, int dy_layout, const float* w, const float* x_act, float* dx,
                        int dx_layout, int N, int Cin, int H, int W, int Cout, void* stream);
int dvae_convT4s2_wgrad(const float* x, int x_layout, const float* dy, int dy_layout, float* dw,
                        float* db, int N, int Cin, int H, int W, int Cout, float* ws, void* stream);
size_t dvae_conv_wgrad_ws_floats(void);

/* ---- per-step weight staging + the tuned 32 <-> 32 channel kernels on pre-staged weights ------------------------------
 * The six hidden conv / convT layers of the 64x64 Burgess stack (encoders.py:55-60, decoders.py:57-64; on 32x32 images the
 * four that exist) are 18 launches per training iteration that each keep a re-laid 64 KB image of the layer's weights in
 * LDS, and the six fully-connected layers (encoders.py:63-67, decoders.py:53-55) are streamed by the FC-chain kernels below
 * in k-chunked order.  dvae_stage_weights writes all of those images in ONE launch at the head of a forward pass (the
 * parameters only change in optimizer.step(), training.py:158), optionally together with the step's loss coefficients
 * (= dvae_set_coef).  Images: conv img_down / img_up = 16384 floats each (wl[tap][kc/4][n][kc%4], kc = cb / cs);
 * fc img_fwd = [ceil(K/4)][N][4] floats, img_bwd = [ceil(N/4)][K][4] floats (zero padded).  Any image pointer may be NULL
 * (skipped).  `conv`, `fc`, `thin`, `coef_vals` are HOST arrays / structs read during the call.                         */
#define DVAE_STAGE_MAX_CONV 6
#define DVAE_STAGE_MAX_FC 8
#define DVAE_THIN_PAIR_FLOATS(C) ((C) == 3 ? 112 : 16)  /* per contracted channel: the pair record of k_up_thin_pk (48 / 16 floats,
                                                          conv_thin.hip) and, C = 3, 64 floats of the matrix-core forward kernel's
                                                          operand image (k_up_thin_mm: 2048 floats behind the 32 records) */
typedef struct {
  const float* w;       /* Conv2d [32,32,4,4] or ConvTranspose2d [32,32,4,4] weight = w[cs][cb][kh][kw] */
  float* img_down;      /* consumed by dvae_conv32_down */
  float* img_up;        /* consumed by dvae_conv32_up   */
} dvae_conv_image_desc;
typedef struct {
  const float* w;       /* nn.Linear weight [N,K] */
  float* img_fwd;       /* forward operand stream of dvae_fc_chain_fwd */
  float* img_bwd;       /* input-gradient operand stream of dvae_fc_chain_bwd */
  int N, K;
} dvae_fc_image_desc;
typedef struct {
  const float* w;       /* the last decoder layer's ConvTranspose2d weight [32,C,4,4] (decoders.py:65), C in {1, 3} */
  float* img_pairs;     /* 32 * DVAE_THIN_PAIR_FLOATS(C) floats: operand pairs of the packed-FMA forward kernel (+ the matrix-core image) */
  int C;
} dvae_thin_image_desc;
int dvae_stage_weights(const dvae_conv_image_desc* conv, int n_conv, const dvae_fc_image_desc* fc, int n_fc,
                       const dvae_thin_image_desc* thin /* may be NULL */,
                       float* coef /* device, may be NULL */, const float* coef_vals /* host float[8], may be NULL */,
                       void* stream);
/* The last decoder layer on its staged pair records: recon[N,C,64,64] (NCHW) = sigmoid(convT(x[N,32,32,32] NHWC) + b)
 * (decoders.py:82); with target != NULL (fp32 [N,C,64,64], or uint8 pixels when target_is_u8) also the reconstruction
 * likelihood partial sums and g = coef[INV_B] * dLoss/dlogit, exactly as dvae_convT4s2_sigmoid_recon_fwd[_u8]
 * (losses.py:394-449) -- the same kernel structure with the multiply-adds issued as packed fp32 FMAs.             */
int dvae_convT3_fwd_staged(const float* x, const float* img_pairs, const float* b, const void* target, int target_is_u8,
                           float* recon, float* g, int dist, const float* coef, float* partials, int N, int C,
                           void* stream);
/* "down" = big[N,32,2Hs,2Hs] -> small[N,32,Hs,Hs]: Conv2d forward (encoders.py:75-77: bias + ReLU, mask NULL) and
 * ConvTranspose2d input gradient (decoders.py:77-80 under training.py:157: bias NULL, act none, mask = the producing
 * layer's post-ReLU output, gradient zeroed where it is 0).  "up" = small -> big: ConvTranspose2d forward / Conv2d
 * input gradient.  Hs in {4, 8, 16}; NHWC on both sides, except that the 4x4 side of Hs = 4 may be NCHW (= the (c,h,w)
 * flatten order of encoders.py:80 / decoders.py:74).  Same kernels, same results as dvae_conv4s2_* / dvae_convT4s2_* on the
 * raw weights: only the prologue differs.                                                                         */
int dvae_conv32_down(const float* big, const float* img_down, const float* bias, const float* mask, float* out,
                     int out_layout, int N, int Hs, int act, void* stream);
int dvae_conv32_up(const float* small, int small_layout, const float* img_up, const float* bias, const float* mask,
                   float* out, int N, int Hs, int act, void* stream);

/* ---- ReLU masks as bit planes (new; the reference's autograd keeps the whole post-ReLU activation for relu's backward:
 * encoders.py:73-77, decoders.py:77-80 under training.py:157).  For a 32-channel NHWC activation a[P pixels][32] the bit plane
 * is bits[P] (uint32), bit c = [a[p][c] > 0]: 4 bytes per pixel instead of 128.  At the 32x32x32 geometry (the outputs of
 * conv1 and convT2 on 64x64 images: 134 MB each at B = 1024) the forward kernels EMIT the plane next to the activation and the
 * input-gradient kernels of the following layer CONSUME it instead of re-reading the activation for its sign.  Results are
 * bit-identical to the fp32-mask entry points.
 *   dvae_conv1_fwd_bits    : y[N,32,32,32] (NHWC) = relu(conv(x[N,Cin,64,64] NCHW fp32 or uint8 pixels) + b) and y_bits[N*1024]
 *                            (= dvae_conv4s2_fwd / _u8 at that geometry, encoders.py:54,73)
 *   dvae_conv32_up_bits    : the 16x16 -> 32x32 member of dvae_conv32_up; exactly one of mask_bits (input-gradient form: the
 *                            result is zeroed where the bit is clear; conv2's input gradient, mask = conv1's output) and
 *                            out_bits (forward form with ReLU: convT2's forward, decoders.py:80) is given
 *   dvae_convT3_dgrad_bits : dx[N,32,32,32] (NHWC) = convT3 input gradient of dy[N,Cout,64,64] (NCHW), zeroed where the bit of
 *                            x_act_bits[N*1024] (convT2's output) is clear (= dvae_convT4s2_dgrad at that geometry)          */
int dvae_conv1_fwd_bits(const void* x, int x_is_u8, const float* w, const float* b, float* y, uint32_t* y_bits, int N,
                        int Cin, void* stream);
int dvae_conv32_up_bits(const float* small, const float* img_up, const float* bias, const uint32_t* mask_bits, float* out,
                        uint32_t* out_bits, int N, int act, void* stream);
int dvae_convT3_dgrad_bits(const float* dy, const float* w, const uint32_t* x_act_bits, float* dx, int N, int Cout,
                           void* stream);

/* ---- uint8 input pipeline: utils/datasets.py:204-213 (dSprites: imgs * 255 -> ToTensor), :282-291 (CelebA:
 * imread -> ToTensor).  The batch stays uint8 [N,C,H,W] in HBM (NCHW = ToTensor's output order, 1 byte per pixel);
 * ToTensor's arithmetic, float(v) / 255 with IEEE division, is applied by the three consumers of the input image
 * while they stage it: conv1 forward, conv1 weight gradient and the reconstruction-likelihood target of the fused last
 * decoder layer.  Results are bit-identical to running the fp32 entry points on dvae_u8_to_f32's output.
 * Fused shapes: C in {1,3}, 64x64 images, 32 output channels (dvae_u8_fused_supported); other geometries convert once
 * with dvae_u8_to_f32 and use the fp32 entry points.                                                             */
int dvae_u8_to_f32(const uint8_t* src, float* dst, long n, void* stream);       /* both 16-byte aligned */
int dvae_u8_fused_supported(int C, int H, int W);
int dvae_conv4s2_fwd_u8(const uint8_t* x, const float* w, const float* b, float* y /* NHWC */, int N, int Cin,
                        int H, int W, int Cout, int act, void* stream);
int dvae_conv4s2_wgrad_u8(const uint8_t* x, const float* dy /* NHWC */, float* dw, float* db, int N, int Cin,
                          int H, int W, int Cout, float* ws, void* stream);
int dvae_convT4s2_sigmoid_recon_fwd_u8(const float* x /* NHWC */, const float* w, const float* b,
                                       const uint8_t* target, float* recon, float* g, int dist,
                                       const float* coef, float* partials, int N, int Cin, int H, int W,
                                       int Cout, void* stream);

/* NCHW <-> NHWC re-layout of a [N,C,H,W] tensor (the flatten of encoders.py:80 /
 * the view of decoders.py:74 are in c,h,w order; the engine's conv activations are NHWC).  */
int dvae_relayout(const float* src, int src_layout, float* dst, int N, int C, int H, int W, void* stream);

/* ---- nn.Linear: encoders.py:63-67,81-86; decoders.py:53-55,71-73; discriminator.py:51-68 --
 * y[M,N] = act(x[M,K] w[N,K]^T + b[N]);  act in {none, relu, leaky 0.2}.
 * ws (all three, may be NULL): the dvae_conv_wgrad_ws_floats() workspace; enables the
 * split-contraction schedule for layers with few output tiles (fixed-order, deterministic). */
int dvae_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int K, int N,
                    int act, float* ws, void* stream);
/* dx[M,K] = dy[M,N] w[N,K], times act'(x_act) when x_act != NULL (x_act = the post-activation
 * input of this layer, i.e. the ReLU / LeakyReLU backward of the previous layer fused in). */
int dvae_linear_dgrad(const float* dy, const float* w, const float* x_act, int act, float* dx,
                      int M, int K, int N, float* ws, void* stream);
/* dw[N,K] = dy^T x ; db[N] = column sums of dy (db may be NULL).  ws (may be NULL): the
 * dvae_conv_wgrad_ws_floats() workspace, enables the split-batch (split-K) schedule for layers
 * with few output tiles (fixed-order reduction, deterministic).                              */
int dvae_linear_wgrad(const float* x, const float* dy, float* dw, float* db, int M, int K, int N,
                      float* ws, void* stream);

/* Grouped form of dvae_linear_wgrad: n <= DVAE_FCW_MAX independent problems (the weight gradients of the six
 * fully-connected layers of the VAE, encoders.py:63-67 / decoders.py:53-55 under training.py:157) in ONE launch of
 * ~400 short-lived workgroups instead of six chip-starving ones.  `descs` is a HOST array read during the call (not
 * retained).  Fixed summation order (deterministic), no workspace; rows need no alignment (16-byte loads are used
 * where a problem's rows allow them).                                                                               */
#define DVAE_FCW_MAX 8
typedef struct {
  const float* x;   /* [M,K] layer input                    */
  const float* dy;  /* [M,N] gradient w.r.t. the pre-activation output */
  float* dw;        /* [N,K] */
  float* db;        /* [N] or NULL */
  int M, K, N;
} dvae_linear_wgrad_desc;
int dvae_linear_wgrad_grouped(const dvae_linear_wgrad_desc* descs, int n, void* stream);

/* ---- the fully-connected core as one launch per direction (fc_chain.hip) ---------------------------------------------
 * forward : encoder lin1 -> lin2 -> mu_logvar_gen (encoders.py:81-87) -> reparameterise + per-dim KL partials
 *           (vae.py:52-71, losses.py:452-480) -> decoder lin1 -> lin2 -> lin3 (decoders.py:71-73), for rows [0, n_enc);
 *           KL terms only from rows < n_kl, decoder only for rows < n_dec (FactorVAE: both halves of the batch are encoded,
 *           the first half enters the KL and is decoded, losses.py:254-259,286).  Hidden sizes are the Burgess ones
 *           (512 -> 256 -> 256 -> 2D, D -> 256 -> 256 -> 512), 1 <= D <= DVAE_MAX_D, n_enc <= 8 * DVAE_KL_MAX_BLOCKS.
 * backward: the input-gradient chain of the same six layers with the ReLU masks of the saved activations and
 *           dvae_reparam_kl_bwd's arithmetic in the middle, rows [0, n) (weight gradients: dvae_linear_wgrad_grouped).
 * Weight operands are the img_fwd / img_bwd images of dvae_stage_weights.  Same results as the per-layer entry points
 * up to fp32 summation order (documented tolerance: rtol 1e-5 of the layer scale).  Structs are HOST memory.        */
typedef struct {
  const float* a_flat;                                      /* [n_enc,512] conv-stack output, (c,h,w) order, post-ReLU */
  const float *w_e1, *w_e2, *w_ml, *w_d1, *w_d2, *w_d3;     /* img_fwd of encoder.lin1, lin2, mu_logvar_gen, decoder.lin1..3 */
  const float *b_e1, *b_e2, *b_ml, *b_d1, *b_d2, *b_d3;     /* their biases */
  const float* eps;                                         /* [n_enc,D] N(0,1) draws; NULL: z = mu (eval mode) */
  float *h1, *h2, *ml, *mu, *logvar, *z;                    /* [n_enc,256] x2, [n_enc,2D] (interleaved), [n_enc,D] x3 */
  float* kl_part;                                           /* [ceil(n_enc/dvae_fc_chain_rows(n_enc)),16] partial blocks (= kl_dim + 16) or NULL */
  float *d1, *d2, *d3;                                      /* [n_dec,256] x2, [n_dec,512] */
  int n_enc, n_kl, n_dec, D;
  /* The 4x4 end of the conv stacks in the same launch (version >= 109; 64x64 geometry; all NULL: the plain chain).
   * conv_in != NULL: the launch FIRST computes a_flat = ReLU(conv_64(conv_in) + conv_b) (encoders.py:76-80) -- a_flat is then an
   * OUTPUT -- bit-identical to dvae_conv32_down(conv_in, conv_w, conv_b, NULL, a_flat, DVAE_NCHW, n_enc, 4, DVAE_ACT_RELU).
   * convT_w != NULL (needs conv_in): it ENDS with convT_out = ReLU(convT_64(d3) + convT_b) (decoders.py:74-76) for the rows
   * < n_dec, bit-identical to dvae_conv32_up(d3, DVAE_NCHW, convT_w, convT_b, NULL, convT_out, n_dec, 4, DVAE_ACT_RELU).      */
  const float* conv_in;                                     /* [n_enc,8,8,32] NHWC: conv3's post-ReLU output */
  const float *conv_w, *conv_b;                             /* img_down of encoder.conv_64 (dvae_stage_weights), its bias */
  const float *convT_w, *convT_b;                           /* img_up of decoder.convT_64, its bias */
  float* convT_out;                                         /* [n_dec,8,8,32] NHWC */
} dvae_fc_chain_fwd_args;
typedef struct {
  const float* gd3;                                         /* [n,512] gradient w.r.t. decoder.lin3's pre-activation output */
  const float *w_d3, *w_d2, *w_d1, *w_ml, *w_e2, *w_e1;     /* img_bwd of the six layers */
  const float *d2, *d1, *h2, *h1, *a_flat;                  /* saved post-ReLU activations (masks) */
  const float *mu, *logvar, *eps;                           /* [n,D]; eps NULL: z = mu */
  const float *dz2, *dz3, *dmu_x, *dlv_x;                   /* as in dvae_reparam_kl_bwd; each may be NULL */
  const float *scal, *coef;
  float *gd2, *gd1, *dz /* may be NULL */, *dml, *gh2, *gh1, *ga_flat;   /* [n,256] x2, [n,D], [n,2D], [n,256] x2, [n,512] */
  int n, D;
  /* The 4x4 end in the same launch (version >= 109; all NULL: the plain chain).  convT_gout != NULL: the launch FIRST computes
   * gd3 = [d3 > 0] * (convT_64's input gradient of convT_gout) -- gd3 is then an OUTPUT -- bit-identical to
   * dvae_conv32_down(convT_gout, convT_w, NULL, d3, gd3, DVAE_NCHW, n, 4, DVAE_ACT_NONE).  conv_w != NULL (needs convT_gout): it
   * ENDS with conv_gin = [conv_act > 0] * (conv_64's input gradient of ga_flat), bit-identical to
   * dvae_conv32_up(ga_flat, DVAE_NCHW, conv_w, NULL, conv_act, conv_gin, n, 4, DVAE_ACT_NONE).                                */
  const float* convT_gout;                                  /* [n,8,8,32] NHWC: gradient w.r.t. convT_64's output (already masked) */
  const float *convT_w, *d3;                                /* img_down of decoder.convT_64; [n,512] lin3's post-ReLU output */
  const float *conv_w, *conv_act;                           /* img_up of encoder.conv_64; [n,8,8,32] conv3's post-ReLU output */
  float* conv_gin;                                          /* [n,8,8,32] NHWC: gradient w.r.t. conv3's output */
} dvae_fc_chain_bwd_args;
int dvae_fc_chain_fwd(const dvae_fc_chain_fwd_args* args, void* stream);
int dvae_fc_chain_bwd(const dvae_fc_chain_bwd_args* args, void* stream);
/* Batch rows one workgroup of the chain kernels owns in a launch over n rows (4 up to 1024 rows, 8 above: small batches get
 * twice the workgroups, each with half the matrix-core work) = the granularity of the forward's KL partial blocks:
 * dvae_fc_chain_fwd over n_enc rows leaves ceil(n_enc / dvae_fc_chain_rows(n_enc)) of them at kl_part.               */
int dvae_fc_chain_rows(int n);

/* ---- reparameterisation + per-dim Gaussian KL: vae.py:52-71, losses.py:452-480 -----------
 * ml[B,2D] is the interleaved output of mu_logvar_gen (encoders.py:87: mu = ml[:,0::2],
 * logvar = ml[:,1::2]).  z = mu + exp(.5 logvar) eps (eps == NULL: z = mu, eval mode).
 * kl_dim (may be NULL): float[DVAE_KL_FLOATS]; [0,D) = coef[INV_B] * sum_b 0.5(-1 - lv + mu^2 + e^lv),
 * the rest (kl_dim + 16) holds per-workgroup partial sums, blocks of 16 floats (D <= DVAE_MAX_D; above, kl_dim is just the
 * D final values, kl_dim != NULL needs coef, and D must fit the buffer).  With kl_dim != NULL and coef == NULL
 * only the dvae_reparam_kl_blocks(B) partial blocks are written (no finishing launch): dvae_loss_epilogue(kl_blocks = that
 * count) or dvae_kl_finish completes them -- every consumer adds the blocks in the same fixed order.            */
#define DVAE_KL_MAX_BLOCKS 8192
#define DVAE_KL_FLOATS (16 + DVAE_KL_MAX_BLOCKS * 16)
int dvae_reparam_kl_fwd(const float* ml, const float* eps, float* mu, float* logvar, float* z,
                        float* kl_dim, const float* coef, int B, int D, void* stream);
int dvae_reparam_kl_blocks(int B);
/* kl_dim[0,D) <- coef[INV_B] * (sum of the kl_blocks partial blocks at kl_dim + 16)  (sharded batches: before dvae_loss_pack) */
int dvae_kl_finish(float* kl_dim, int kl_blocks, const float* coef, int D, void* stream);
/* dml[B,2D] (interleaved) from dz + dz2 + dz3 [B,D] (each may be NULL: gradients that reach z
 * through the decoder, the TC estimator, the discriminator -- quirk Q1) and optional direct grads
 * dmu_x/dlv_x[B,D]; the KL term enters with weight scal[DVAE_S_KLW] * coef[INV_B].             */
int dvae_reparam_kl_bwd(const float* dz, const float* dz2, const float* dz3, const float* dmu_x,
                        const float* dlv_x, const float* mu,
                        const float* logvar, const float* eps, const float* scal, const float* coef,
                        float* dml, int B, int D, void* stream);

/* Backward of _kl_normal_loss's per-dimension KL (losses.py:470-473) for an upstream gradient g_dim[D]:
 * dmu[b,d] = g_dim[d] mu[b,d] / B, dlogvar[b,d] = g_dim[d] 0.5 (e^logvar - 1) / B  (the autograd-compatible path). */
int dvae_kl_normal_bwd(const float* g_dim, const float* mu, const float* logvar, float* dmu, float* dlogvar,
                       int B, int D, void* stream);
/* dst[0] = scale * sum(src[0..n)), fixed summation order (finishes per-workgroup partial sums of a loss).  */
int dvae_reduce_sum(const float* src, long n, float scale, float* dst, void* stream);

/* ---- reconstruction likelihood: losses.py:394-449 (F.binary_cross_entropy / mse / l1) ----
 * recon, target: [n] elements.  partials[DVAE_REC_NPART] receives per-block partial sums of
 * the un-normalised loss; g[n] (may be NULL) = coef[INV_B] * dLoss/d(pre-sigmoid logit)
 * when wrt_logit != 0 (the logit gradient is what the fused backward consumes).            */
#define DVAE_REC_NPART 2048
int dvae_recon_loss(const float* recon, const float* target, long n, int dist, const float* coef,
                    float* partials, float* g, int wrt_logit, void* stream);
/* wrt_logit == 0: g is the gradient w.r.t. `recon` itself (autograd-compatible path).      */
/* out[n] = grad_y * (1 - y) * y : backward of the final torch.sigmoid (decoders.py:82).     */
int dvae_sigmoid_bwd(const float* grad_y, const float* y, float* out, long n, void* stream);

/* Final decoder layer fused with the likelihood: recon = sigmoid(convT(x,w)+b) (decoders.py:82),
 * partials[DVAE_REC_NPART] = per-workgroup sums of the loss vs `target`, g = coef[INV_B] *
 * dLoss/d(pre-sigmoid logit).  x[N,Cin,H,W]; target, recon, g: [N,Cout,2H,2W] NCHW.  One pass over
 * the reconstruction instead of three (convT3 store, loss read, gradient write).              */
int dvae_convT4s2_sigmoid_recon_fwd(const float* x, int x_layout, const float* w, const float* b,
                                    const float* target, float* recon, float* g, int dist,
                                    const float* coef, float* partials, int N, int Cin, int H, int W,
                                    int Cout, void* stream);

/* ---- beta-TCVAE estimator: losses.py:523-544, utils/math.py:8-73 --------------------------
 * z,mu,logvar: [Bg,D] (the whole -- global -- batch); this call evaluates rows
 * [row0,row0+Bl).  log_w = {log(1/N), log(strat), log(1/M)} in fp32 as the reference
 * computes them (math.py:66-73), ignored when is_mss == 0.
 * rowstats[Bl,DVAE_ROWSTATS_STRIDE(D)]: log_pz, log_qz, log_prod_qzi, log_q_zCx, lse_d[0..D-1]  (1 <= D <= DVAE_WIDE_MAX_D).
 * tmp[DVAE_BTCVAE_TMP_FLOATS(Bg,Bl,D)]: scratch filled by fwd with the transposed per-column constants (mu, -0.5(log
 * 2pi + logvar), exp(-logvar)) and re-read by bwd: pass the same buffer to both.             */
int dvae_btcvae_fwd(const float* z, const float* mu, const float* logvar, int Bg, int D, int row0,
                    int Bl, int is_mss, const float* log_w, float* tmp, float* rowstats, void* stream);
/* gradient of alpha*mi + beta*tc + anneal*gamma*dw_kl (means over Bg rows) restricted to
 * rows [row0,row0+Bl): dz[Bl,D] (local rows), dmu_all/dlv_all[Bg,D] (column sums over the
 * local rows; reduce over ranks when the batch is sharded).                                 */
int dvae_btcvae_bwd(const float* z, const float* mu, const float* logvar, const float* rowstats,
                    int Bg, int D, int row0, int Bl, int is_mss, const float* log_w,
                    const float* coef, const float* tmp, float* dz, float* dmu_all, float* dlv_all,
                    void* stream);

/* ---- FactorVAE pieces: losses.py:261-265,293-295,483-508 ----------------------------------*/
/* out[b,d] = z[perm[d*B+b], d]; perm int64 [D,B] (torch.randperm per latent dim).          */
int dvae_permute_dims(const float* z, const int64_t* perm, float* out, int B, int D, void* stream);
/* dlogits[2*Bh,2]: rows [0,Bh) = D(z1), rows [Bh,2Bh) = D(z_perm).  sums[4] receives
 * {sum(d_z[:,0]-d_z[:,1]), sum CE(d_z,0), sum CE(d_z_perm,1), 0};
 * g_dtc[2Bh,2] = d(0.5(CE+CE))/dlogits; g_tc[Bh,2] = coef[ANNEAL]*coef[BETA]/Bh * [+1,-1]. */
int dvae_disc_losses(const float* dlogits, int Bh, const float* coef, float* sums, float* g_dtc,
                     float* g_tc, void* stream);

/* ---- entropy estimator of the MIG / AAM metrics: evaluate.py:233-297 (_estimate_latent_entropies) -------
 * H[d] = 1/S sum_s [ log N - logsumexp_n log N(z_ds[d,s]; mean[n,d], exp(logvar[n,d])) ],  d < D (any D).
 * z_ds[D,S]: the S sampled latents exactly as the reference lays them out -- the [S,D] gather of sampled rows
 * re-viewed as [D,S] (evaluate.py:262, a reshape, not a transpose); mean, logvar: [N,D] (the whole data set or a
 * conditional slice of it).  ws: dvae_latent_entropy_ws_floats(N, D, S) floats.                          */
size_t dvae_latent_entropy_ws_floats(long N, int D, int S);
int dvae_latent_entropy(const float* z_ds, const float* mean, const float* logvar, long N, int D, int S,
                        float* ws, float* H, void* stream);

/* ---- scalar epilogue of the loss plugins: losses.py:139-153,186-202,268-274,369-389 -------
 * dvae_loss_pack reduces this rank's partial sums into packed[DVAE_NPACK_D(D)]:
 *   [0] sum of rec_partials, [1..16] kl_dim, [17..20] column sums of rowstats[:,0..3]
 *   (log_pz, log_qz, log_prod_qzi, log_q_zCx), [21..23] discriminator sums; D > DVAE_MAX_D: kl_dim at [32..32+D), [1..16] zero,
 *   rowstats with stride DVAE_ROWSTATS_STRIDE(D).
 * When the batch is sharded over ranks, sum-all-reduce `packed` before dvae_loss_finalize.
 * dvae_loss_finalize(kind = DVAE_LOSS_*) writes scal[DVAE_NSCAL_D(D)]; Bg = global batch
 * (btcvae) or global half batch (factor).                                                   */
#define DVAE_NPACK 32
int dvae_loss_pack(const float* rec_partials, const float* kl_dim, int D, const float* rowstats, int Bl,
                   const float* disc_sums, float* packed, void* stream);
int dvae_loss_finalize(int kind, const float* packed, int D, int Bg, const float* coef, float* scal,
                       void* stream);
/* Un-sharded batches: dvae_loss_pack and dvae_loss_finalize in ONE launch (scal == NULL: pack only).
 * kl_blocks > 0: kl_dim + 16 holds that many un-finished partial blocks (dvae_reparam_kl_fwd(coef = NULL):
 * dvae_reparam_kl_blocks(B) of them; dvae_fc_chain_fwd: ceil(n_enc / dvae_fc_chain_rows(n_enc))); they are summed (same order as
 * dvae_kl_finish) and scaled by coef[INV_B].  kl_blocks == 0: kl_dim[0,D) is final.                 */
int dvae_loss_epilogue(int kind, const float* rec_partials, const float* kl_dim, int kl_blocks, int D,
                       const float* rowstats, int Bl, const float* disc_sums, int Bg,
                       const float* coef, float* packed, float* scal, void* stream);

/* coef[0..7] <- the eight values (passed as kernel arguments: in-order with the stream, no
 * host buffer whose lifetime must outlive the launch).                                      */
int dvae_set_coef(float* coef, float c0, float c1, float c2, float c3, float c4, float c5, float c6,
                  float c7, void* stream);

/* out[i] = a[i] + b[i] (n elements), helper for merging latent gradients (quirk Q1).       */
int dvae_add(const float* a, const float* b, float* out, long n, void* stream);

/* ---- torch.optim.Adam's update as one launch (main.py:208, losses.py:238: the optimizers; training.py:158,
 * losses.py:307-308: their step()) ------------------------------------------------------------------------------------
 * The optimizer object, its hyper-parameters and its state tensors stay the caller's (torch's): this runs the element-wise
 * arithmetic of Adam.step() -- amsgrad off, maximize off, L2 weight decay added to the gradient -- over `nt` tensors:
 *   g' = g + wd p;  m += (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps),   t = step_new
 * and writes t into every tensor's `step` slot (fp32 scalar on the device, as torch's fused Adam keeps it; may be NULL).
 * `tensors` is HOST memory (copied into the launch).  Within 1 ulp of the parameter of torch's implementations.          */
typedef struct {
  float* p;            /* parameter [n] */
  const float* g;      /* gradient [n] */
  float* m;            /* exp_avg [n] */
  float* v;            /* exp_avg_sq [n] */
  float* step;         /* state["step"]: one float on the device, or NULL */
  long n;
} dvae_adam_tensor;
int dvae_adam_step(const dvae_adam_tensor* tensors, int nt, float step_new, double lr, double beta1, double beta2,
                   double eps, double weight_decay, void* stream);

/* ---- glue of the data-parallel step (new: the reference is single-process; disvae_amd/parallel.py) -------------------
 * out[i] = alpha * a[i] + beta * b[i] (b may be NULL: out = alpha * a; out may alias a or b): the 1 / world factors of means
 * over the global batch.                                                                                              */
int dvae_axpby(float* out, const float* a, float alpha, const float* b, float beta, long n, void* stream);
/* dst[j][i][:] = src[i][j][:] for src [A][Bn][inner] floats: the packed exchanges of the sharded beta-TCVAE step -- an
 * all-gather of every rank's (z, mu, logvar) delivers [world][3][B*D] where the estimator (losses.py:523-544 over the
 * GLOBAL batch) reads three [world*B, D] tensors; its column gradients (dmu, dlogvar) [2][world][B*D] leave through a
 * reduce-scatter as [world][2][B*D].                                                                                  */
int dvae_swap_outer(const float* src, float* dst, int A, int Bn, long inner, void* stream);

/* ---- stream ordering (new; the reference is one stream of ATen ops) -----------------------------------------------
 * Work enqueued on `later` after this call runs after everything enqueued on `earlier` before it (both hipStream_t of the
 * current device).  The native training step issues its weight-gradient kernels on a second stream beside the chain of
 * input gradients (training.py:157 is ONE autograd pass: only the data dependencies between its kernels matter); this is
 * its fork / join primitive: an event recorded with a DEVICE-scope release (hipEventReleaseToDevice: no system-scope cache
 * flush, the two streams share the device) from a small internal pool, then hipStreamWaitEvent.  Capturable.            */
int dvae_stream_order(void* earlier, void* later);
/* A new non-blocking stream on the current device (never destroyed: the host side keeps ONE side stream, one exchange stream
 * and one communication stream per device and process).  Why not the framework's stream pool: HIP multiplexes streams onto
 * GPU_MAX_HW_QUEUES hardware queues, a new stream joining the least-loaded queue; a pool creates dozens of streams at once, so
 * which pool entry a caller is handed decides whether its stream shares a queue with the default stream -- and two streams of
 * one iteration on one queue serialise (0.35 -> 0.79-1.3 ms per 128-image iteration, profiles/r05_final1_bench.json,
 * r05_v28).  A stream created here lands on a queue the default stream is not on.                                       */
int dvae_stream_create(void** stream /* out: hipStream_t */);
/* The same in two halves, for a consumer that is enqueued much later than the producer: dvae_event_record marks the work
 * enqueued on `stream` so far in slot `slot` (0 <= slot < DVAE_EVENT_SLOTS, per device); dvae_event_wait makes the work enqueued
 * on `stream` afterwards wait for the slot's LAST mark.  (The btcvae step: the estimator's backward kernels and the scalar
 * loss epilogue stay on the side stream until the FC chain's input gradients need them, a whole convT backward later.)   */
#define DVAE_EVENT_SLOTS 16
int dvae_event_record(int slot, void* stream);
int dvae_event_wait(int slot, void* stream);

/* ---- recorded launch lists (new; host-side overhead only) ------------------------------------------------------------
 * A training iteration at a fixed batch size repeats the same entry-point calls with the same arguments; the host records
 * them once and replays the list with ONE call per step segment instead of one foreign call per launch.  An entry names its
 * entry point by dvae_plan_op("dvae_...") (>= 0; -1: not replayable) and carries every argument in 64 bits: pointers and
 * integers by value (sign-extended), floats as their fp32 bit pattern in the low word.  HOST structs passed by pointer
 * (dvae_stage_weights tables, dvae_fc_chain_*_args, wgrad descriptors) must stay alive and are re-read at every replay.
 * dvae_plan_run executes the entries in order and stops at the first failure (returns its status).               */
#define DVAE_PLAN_MAX_ARGS 20
typedef struct {
  int op, nargs;
  uint64_t args[DVAE_PLAN_MAX_ARGS];
} dvae_plan_entry;
int dvae_plan_op(const char* name);
int dvae_plan_run(const dvae_plan_entry* entries, int n);

/* ---- RCCL collectives over xGMI (data parallelism over the GPUs of one node; new, the reference is single-process) ----
 * One communicator per process (= per GPU).  Every collective is ENQUEUED on `stream` (ordered with the kernels around
 * it, no host synchronisation) in fp32 with sum reduction.  librccl is dlopen()ed on first use ($DVAE_RCCL_LIB, the
 * path given to dvae_comm_load, "librccl.so", /opt/rocm/lib/librccl.so); single-GPU processes never load it.
 * Rendezvous: rank 0 calls dvae_comm_unique_id and ships the 128 bytes to the other ranks out of band (the host
 * side broadcasts them through its torch.distributed store); then every rank calls dvae_comm_init with the current HIP
 * device selected.  Uses per step: all-reduce of the flat gradient arena(s) and of the packed loss
 * sums, all-gather of (z, mu, logvar) / z2 for the global B x B estimator / permute_dims, reduce-scatter of the
 * estimator's column gradients.                                                                                  */
typedef struct dvae_comm dvae_comm;
int dvae_comm_load(const char* librccl_path /* may be NULL */);
int dvae_comm_unique_id(void* id128 /* out: 128 bytes */);
int dvae_comm_init(dvae_comm** comm, const void* id128, int world, int rank);
int dvae_comm_destroy(dvae_comm* comm);
int dvae_comm_world(const dvae_comm* comm);
int dvae_comm_rank(const dvae_comm* comm);
int dvae_comm_allreduce(dvae_comm* comm, float* buf, long n, void* stream);                          /* in place */
int dvae_comm_allgather(dvae_comm* comm, const float* send, float* recv, long n_per_rank, void* stream);
int dvae_comm_reducescatter(dvae_comm* comm, const float* send, float* recv, long n_per_rank, void* stream);
int dvae_comm_broadcast(dvae_comm* comm, float* buf, long n, int root, void* stream);                /* in place */
int dvae_comm_group_start(void);      /* ncclGroupStart / ncclGroupEnd: fuse the collectives in between */
int dvae_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* DVAE_HIP_H */
