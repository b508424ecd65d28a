// Per-step weight staging: ONE launch, issued at the head of every forward pass, re-lays the parameters that the tuned
// kernels consume in a kernel-specific order, so that no kernel of the step re-derives that order on its own critical path:
//   * the six 32 <-> 32 channel conv / convT layers (encoders.py:55-60, decoders.py:57-64): the two 64 KB LDS weight images
//     wl[tap][kc/4][n][kc%4] of conv_mfma_common.h (kc = contracted channel; "down" contracts over cb, "up" over cs).  The
//     18 conv launches of an iteration copy their image with 16-byte LDS-DMA transfers (copy_weight_image) instead of
//     re-laying 64 KB with scalar, bank-conflicting LDS stores: 7.4 us -> ~1.5 us of prologue per launch
//     (profiles/r02_run14_15_downws_wgws_ablation.txt);
//   * the six fully-connected layers (encoders.py:63-67, decoders.py:53-55): k-chunked images [K/4][N][4] (forward, contract
//     over K) and [N/4][K][4] (input gradient, contract over N) -- the operand streams of the fused FC-chain kernels
//     (fc_chain.hip): one coalesced 16-byte load per lane = four contraction steps of the lane's output column;
//   * the last decoder layer (decoders.py:65): its weights as per-channel records of operand PAIRS for the packed-FMA
//     forward kernel (k_up_thin_pk, conv_thin.hip);
//   * optionally the eight loss coefficients of the step (dvae_set_coef folded in: one launch less).
// Output-driven: every thread produces ONE 16-byte chunk of one image (coalesced stores, gathered 4-byte reads of
// parameters that sit in L2); 1.0 M floats per step, ~3 us.
#include "common.h"

namespace dvae {

#define STG_MAX_SEG 29        // 2 images x (DVAE_STAGE_MAX_CONV + DVAE_STAGE_MAX_FC) + the thin layer's pair records
enum { SEG_CONV_DOWN = 0, SEG_CONV_UP = 1, SEG_FC_FWD = 2, SEG_FC_BWD = 3, SEG_THIN_PAIRS = 4 };

struct StageSeg {
  const float* w;
  float* img;
  int kind, N, K;
  int wg0;         // first workgroup of this segment
};
struct StageTable {
  StageSeg seg[STG_MAX_SEG];
  int n;
  float coef[8];
  float* coef_dst;
};

__global__ __launch_bounds__(256) void k_stage_weights(StageTable t) {
  const int tid = threadIdx.x;
  const int wg = blockIdx.x;
  if (wg == 0 && t.coef_dst && tid < 8) t.coef_dst[tid] = t.coef[tid];
  int s = 0;
  for (int q = 1; q < t.n; ++q)
    if (wg >= t.seg[q].wg0) s = q;                       // workgroup-uniform (scalar) scan
  const StageSeg g = t.seg[s];
  if (!g.img) return;                                    // coefficients-only launch
  const long c = (long)(wg - g.wg0) * 256 + tid;         // 16-byte chunk index inside the image
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (g.kind == SEG_CONV_DOWN || g.kind == SEG_CONV_UP) {
    if (c >= 4096) return;
    // image float o = tap*1024 + ((kc>>2)*32 + n)*4 + (kc&3);  w[cs][cb][tap]: down kc = cb, n = cs; up kc = cs, n = cb
    const int tap = (int)(c >> 8), kc4 = (int)(c >> 5) & 7, n = (int)c & 31;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kc = kc4 * 4 + r;
      const int cs = g.kind == SEG_CONV_DOWN ? n : kc, cb = g.kind == SEG_CONV_DOWN ? kc : n;
      v[r] = g.w[(cs * 32 + cb) * 16 + tap];
    }
    reinterpret_cast<f32x4*>(g.img)[c] = v;
  } else if (g.kind == SEG_FC_FWD) {
    // [ceil(K/4)][N][4]: chunk (k4, n) = w[n][4*k4 .. 4*k4+3] (zero beyond K)
    const long nch = (long)((g.K + 3) >> 2) * g.N;
    if (c >= nch) return;
    const int k4 = (int)(c / g.N), n = (int)(c % g.N);
    if ((g.K & 3) == 0) {                                // rows are 16-byte aligned: one load per chunk
      v = *reinterpret_cast<const f32x4*>(g.w + (long)n * g.K + 4 * k4);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * k4 + r;
        v[r] = k < g.K ? g.w[(long)n * g.K + k] : 0.f;
      }
    }
    reinterpret_cast<f32x4*>(g.img)[c] = v;
  } else if (g.kind == SEG_THIN_PAIRS) {
    // per contracted channel cs a record of REC floats in the pair order of k_up_thin_pk (conv_thin.hip); g.N = C
    // C = 3: behind the 32 records, the A-operand image of the matrix-core forward kernel (k_up_thin_mm, conv_up_thin_mm.hip):
    // float (tap * 8 + i) * 64 + lane = W'[m = lane % 16][(a, b, cs = 8 (lane / 16) + i)], tap = 2 a + b, output row
    // m = 4 c + 2 dy + dx (rows 12..15 zero) = w[cs][c][kh = 2 - 2a + dy][kw = 2 - 2b + dx]
    const int REC = g.N == 3 ? 48 : 16;
    if (c >= 32 * DVAE_THIN_PAIR_FLOATS(g.N) / 4) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (int)c * 4 + r;
      if (o < 32 * REC) {
        const int cs = o / REC, idx = o % REC;
        int cb;
        const int tap = thin_pair_source(idx, g.N, &cb);
        v[r] = g.w[(cs * g.N + cb) * 16 + tap];
      } else {
        const int q = o - 32 * REC, mf = q >> 6, ln = q & 63;
        const int m = ln & 15, cs = 8 * (ln >> 4) + (mf & 7), a = mf >> 4, b = (mf >> 3) & 1;
        const int cch = m >> 2, dy = (m >> 1) & 1, dx = m & 1;
        v[r] = m < 12 ? g.w[(cs * 3 + cch) * 16 + (2 - 2 * a + dy) * 4 + (2 - 2 * b + dx)] : 0.f;
      }
    }
    reinterpret_cast<f32x4*>(g.img)[c] = v;
  } else {
    // [ceil(N/4)][K][4]: chunk (n4, k) = w[4*n4 .. 4*n4+3][k] (zero beyond N)
    const long nch = (long)((g.N + 3) >> 2) * g.K;
    if (c >= nch) return;
    const int n4 = (int)(c / g.K), k = (int)(c % g.K);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * n4 + r;
      v[r] = n < g.N ? g.w[(long)n * g.K + k] : 0.f;
    }
    reinterpret_cast<f32x4*>(g.img)[c] = v;
  }
}

int launch_stage_weights(const dvae_conv_image_desc* conv, int n_conv, const dvae_fc_image_desc* fc, int n_fc,
                         const dvae_thin_image_desc* thin, float* coef, const float* coef_vals, hipStream_t s) {
  StageTable t;
  memset(&t, 0, sizeof(t));
  int n = 0, wg = 0;
  auto add = [&](const float* w, float* img, int kind, int N, int K, long chunks) {
    if (!img) return;
    StageSeg& g = t.seg[n++];
    g.w = w; g.img = img; g.kind = kind; g.N = N; g.K = K; g.wg0 = wg;
    wg += (int)((chunks + 255) / 256);
  };
  for (int q = 0; q < n_conv; ++q) {
    add(conv[q].w, conv[q].img_down, SEG_CONV_DOWN, 32, 32, 4096);
    add(conv[q].w, conv[q].img_up, SEG_CONV_UP, 32, 32, 4096);
  }
  for (int q = 0; q < n_fc; ++q) {
    add(fc[q].w, fc[q].img_fwd, SEG_FC_FWD, fc[q].N, fc[q].K, (long)((fc[q].K + 3) / 4) * fc[q].N);
    add(fc[q].w, fc[q].img_bwd, SEG_FC_BWD, fc[q].N, fc[q].K, (long)((fc[q].N + 3) / 4) * fc[q].K);
  }
  if (thin) add(thin->w, thin->img_pairs, SEG_THIN_PAIRS, thin->C, 32, 32 * DVAE_THIN_PAIR_FLOATS(thin->C) / 4);
  t.n = n;
  t.coef_dst = (coef && coef_vals) ? coef : nullptr;
  if (t.coef_dst) for (int i = 0; i < 8; ++i) t.coef[i] = coef_vals[i];
  if (wg == 0) {
    if (!t.coef_dst) return 0;
    wg = 1; t.n = 1; t.seg[0].kind = SEG_CONV_DOWN; t.seg[0].img = nullptr;   // coefficients only
  }
  hipLaunchKernelGGL(k_stage_weights, dim3(wg), dim3(256), 0, s, t);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
