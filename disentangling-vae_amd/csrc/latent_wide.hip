// Latent-space kernels for latent dimensions ABOVE DVAE_MAX_D (16): the reference's --latent-dim is any integer
// (main.py:81) and its losses are dimension-agnostic (losses.py:452-480, 523-544; utils/math.py:8-73).  Every experiment of
// hyperparam.ini uses 10, which is what the fused kernels (loss.hip, fc_chain.hip) are built around: they keep their
// per-dimension state in registers and their scalar slots in fixed 16-wide records.  This file is the same arithmetic with
// the dimension as a run-time bound and NO upper limit: per-dimension state lives in memory, one wave (or one thread) owns one
// (row, dimension) or (column, dimension) pair, and the B x B matrix of JOINT log-densities -- the one quantity that couples
// all dimensions -- is materialised once ([Bl][Bg] floats behind the transposed column constants in `tmp`; the reference
// materialises B x B x D).  The launchers of loss.hip dispatch here when D > DVAE_MAX_D; layouts: include/dvae_hip.h
// (DVAE_ROWSTATS_STRIDE, DVAE_BTCVAE_TMP_FLOATS).  A capability path, not a tuned one: 7 launches of plain kernels.
// All reductions are fixed-order (deterministic).
#include "common.h"

namespace dvae {

#define LOG2PI_W 1.8378770664093453f

// math.py:66-72 with M+1 == B: column 0 <- 1/N, column 1 <- strat, then W[M-1,0] <- strat   (= log_w_ij of loss.hip)
__device__ __forceinline__ float log_w_ij_w(int i, int j, int Bg, float lN, float lS, float lM) {
  if (j == 0) return (i == Bg - 2) ? lS : lN;
  if (j == 1) return lS;
  return lM;
}

// fixed-order sum / max over the 256 threads of a workgroup (4 waves); every thread gets the result.  `red`: 4 floats of LDS.
__device__ __forceinline__ float block_sum_w(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();                              // `red` may still be read from a previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_w(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- reparameterisation (vae.py:52-71), elementwise; ml[B,2D] interleaved (encoders.py:87) ----------------------------
__global__ void k_reparam_wide(const float* __restrict__ ml, const float* __restrict__ eps, float* __restrict__ mu,
                               float* __restrict__ logvar, float* __restrict__ z, long n, int D) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const long b = idx / D; const int d = (int)(idx % D);
  const float m = ml[b * 2 * D + 2 * d], lv = ml[b * 2 * D + 2 * d + 1];
  mu[idx] = m;
  logvar[idx] = lv;
  z[idx] = eps ? m + expf(0.5f * lv) * eps[idx] : m;
}

// ---- per-dimension KL (losses.py:452-480): one workgroup per dimension, kl[d] = coef[INV_B] sum_b 0.5(-1 - lv + mu^2 + e^lv)
__global__ __launch_bounds__(256) void k_kl_cols_wide(const float* __restrict__ mu, const float* __restrict__ logvar, int B,
                                                      int D, const float* __restrict__ coef, float* __restrict__ kl) {
  __shared__ float red[4];
  const int d = blockIdx.x;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float m = mu[(long)b * D + d], lv = logvar[(long)b * D + d];
    acc += 0.5f * (-1.f - lv + m * m + expf(lv));
  }
  const float t = block_sum_w(acc, red);
  if (threadIdx.x == 0) kl[d] = t * coef[DVAE_C_INV_B];
}

// ---- beta-TCVAE estimator (losses.py:523-544, math.py:8-73) -------------------------------------------------------------
// `tmp` = [3][D][Bg] transposed column constants (k_btcvae_prep of loss.hip: mu, -0.5(log 2pi + logvar), exp(-logvar))
// followed by S[Bl][Bg], the joint log-densities of the local rows against all columns.

// S[il][j] = sum_d ( log N(z_i[d]; mu_j[d], var_j[d]) + log W[i][j] )      (quirk Q3: the weight is counted D times)
__global__ __launch_bounds__(256) void k_tcw_joint(const float* __restrict__ z, const float* __restrict__ tmp, int Bg, int D,
                                                   int row0, int is_mss, const float* __restrict__ log_w,
                                                   float* __restrict__ S) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int il = blockIdx.y, i = row0 + il;
  if (j >= Bg) return;
  const float lN = is_mss ? log_w[0] : 0.f, lS = is_mss ? log_w[1] : 0.f, lM = is_mss ? log_w[2] : 0.f;
  const float lw = log_w_ij_w(i, j, Bg, lN, lS, lM);
  const float* muT = tmp; const float* cT = tmp + (long)D * Bg; const float* ivT = tmp + (long)2 * D * Bg;
  float acc = 0.f;
  for (int d = 0; d < D; ++d) {
    const float diff = z[(long)i * D + d] - muT[(long)d * Bg + j];
    const float ld = (cT[(long)d * Bg + j] - 0.5f * (diff * diff * ivT[(long)d * Bg + j])) + lw;
    acc += ld;
  }
  S[(long)il * Bg + j] = acc;
}

// row statistics: one workgroup per local row.  rowstats[il] = {log_pz, log_qz, log_prod_qzi, log_q_zCx, lse_d[0..D-1]}
__global__ __launch_bounds__(256) void k_tcw_rowstats(const float* __restrict__ z, const float* __restrict__ mu,
                                                      const float* __restrict__ lv, const float* __restrict__ tmp,
                                                      const float* __restrict__ S, int Bg, int D, int row0, int is_mss,
                                                      const float* __restrict__ log_w, float* __restrict__ rowstats,
                                                      int rstride) {
  extern __shared__ float lse_d[];              // [D]
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int il = blockIdx.x, i = row0 + il;
  const float lN = is_mss ? log_w[0] : 0.f, lS = is_mss ? log_w[1] : 0.f, lM = is_mss ? log_w[2] : 0.f;
  const float* muT = tmp; const float* cT = tmp + (long)D * Bg; const float* ivT = tmp + (long)2 * D * Bg;
  const float* Srow = S + (long)il * Bg;
  // joint: logsumexp_j S[il][j] -- maximum first, then one exponential per column
  float m = -INFINITY;
  for (int j = tid; j < Bg; j += 256) m = fmaxf(m, Srow[j]);
  const float M = block_max_w(m, red);
  const float sh = M > -INFINITY ? M : 0.f;     // a row of -inf: exp(-inf - 0) = 0 instead of exp(-inf + inf) = NaN
  float s = 0.f;
  for (int j = tid; j < Bg; j += 256) s += __expf(Srow[j] - sh);
  const float lqz = sh + logf(block_sum_w(s, red));
  // marginals: wave wv owns the dimensions wv, wv + 4, ...
  for (int d = wv; d < D; d += 4) {
    const float zi = z[(long)i * D + d];
    float md = -INFINITY;
    for (int j = lane; j < Bg; j += 64) {
      const float diff = zi - muT[(long)d * Bg + j];
      const float ld = (cT[(long)d * Bg + j] - 0.5f * (diff * diff * ivT[(long)d * Bg + j])) + log_w_ij_w(i, j, Bg, lN, lS, lM);
      md = fmaxf(md, ld);
    }
    md = wave_max(md);
    const float shd = md > -INFINITY ? md : 0.f;
    float sd = 0.f;
    for (int j = lane; j < Bg; j += 64) {
      const float diff = zi - muT[(long)d * Bg + j];
      const float ld = (cT[(long)d * Bg + j] - 0.5f * (diff * diff * ivT[(long)d * Bg + j])) + log_w_ij_w(i, j, Bg, lN, lS, lM);
      sd += __expf(ld - shd);
    }
    sd = wave_sum(sd);
    if (lane == 0) lse_d[d] = shd + logf(sd);
  }
  __syncthreads();
  if (tid == 0) {
    float* rs = rowstats + (long)il * rstride;
    float log_pz = 0.f, log_qzCx = 0.f, log_prod = 0.f;
    for (int d = 0; d < D; ++d) {
      const float zi = z[(long)i * D + d], mm = mu[(long)i * D + d], l = lv[(long)i * D + d];
      const float diff = zi - mm;
      log_qzCx += -0.5f * (LOG2PI_W + l) - 0.5f * (diff * diff * expf(-l));
      log_pz += -0.5f * LOG2PI_W - 0.5f * (zi * zi);
      const float lse = lse_d[d];
      rs[4 + d] = lse;
      log_prod += lse;
    }
    rs[0] = log_pz;
    rs[1] = lqz;
    rs[2] = log_prod;
    rs[3] = log_qzCx;
  }
}

// row pass of the gradient: dz[il][d], one wave per (local row, dimension); lanes over the columns
__global__ __launch_bounds__(256) void k_tcw_bwd_rows(const float* __restrict__ z, const float* __restrict__ mu,
                                                      const float* __restrict__ lv, const float* __restrict__ tmp,
                                                      const float* __restrict__ S, const float* __restrict__ rowstats,
                                                      int rstride, int Bg, int D, int row0, int is_mss,
                                                      const float* __restrict__ log_w, const float* __restrict__ coef,
                                                      float* __restrict__ dz) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int il = blockIdx.x, d = blockIdx.y * 4 + wv;
  if (d >= D) return;                           // (no workgroup barrier below)
  const int i = row0 + il;
  const float lN = is_mss ? log_w[0] : 0.f, lS = is_mss ? log_w[1] : 0.f, lM = is_mss ? log_w[2] : 0.f;
  const float alpha = coef[DVAE_C_ALPHA], beta = coef[DVAE_C_BETA], gam = coef[DVAE_C_GAMMA] * coef[DVAE_C_ANNEAL];
  const float invB = 1.f / (float)Bg;
  const float cP = (beta - alpha) * invB, cQ = (gam - beta) * invB;
  const float* muT = tmp + (long)d * Bg; const float* cT = tmp + (long)(D + d) * Bg; const float* ivT = tmp + (long)(2 * D + d) * Bg;
  const float* Srow = S + (long)il * Bg;
  const float* rs = rowstats + (long)il * rstride;
  const float lqz = rs[1], lse = rs[4 + d];
  const float zi = z[(long)i * D + d];
  float g = 0.f;
  for (int j = lane; j < Bg; j += 64) {
    const float iv = ivT[j];
    const float diff = zi - muT[j];
    const float r = diff * iv;
    const float ld = (cT[j] - 0.5f * (diff * diff * iv)) + log_w_ij_w(i, j, Bg, lN, lS, lM);
    const float G = cP * __expf(Srow[j] - lqz) + cQ * __expf(ld - lse);
    g -= G * r;
  }
  g = wave_sum(g);
  if (lane == 0) {
    // diagonal terms: alpha * log q(z_i|x_i) / B  and  -gamma' * log p(z_i) / B
    const float r = (zi - mu[(long)i * D + d]) * expf(-lv[(long)i * D + d]);
    dz[(long)il * D + d] = g - alpha * invB * r + gam * invB * zi;
  }
}

// column pass: dmu[j][d], dlv[j][d] summed over the local rows; thread = column j, blockIdx.y = dimension, rows in order
__global__ __launch_bounds__(256) void k_tcw_bwd_cols(const float* __restrict__ z, const float* __restrict__ mu,
                                                      const float* __restrict__ lv, const float* __restrict__ S,
                                                      const float* __restrict__ rowstats, int rstride, int Bg, int D,
                                                      int row0, int Bl, int is_mss, const float* __restrict__ log_w,
                                                      const float* __restrict__ coef, float* __restrict__ dmu,
                                                      float* __restrict__ dlv) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int d = blockIdx.y;
  if (j >= Bg) return;
  const float lN = is_mss ? log_w[0] : 0.f, lS = is_mss ? log_w[1] : 0.f, lM = is_mss ? log_w[2] : 0.f;
  const float alpha = coef[DVAE_C_ALPHA], beta = coef[DVAE_C_BETA], gam = coef[DVAE_C_GAMMA] * coef[DVAE_C_ANNEAL];
  const float invB = 1.f / (float)Bg;
  const float cP = (beta - alpha) * invB, cQ = (gam - beta) * invB;
  const float mj = mu[(long)j * D + d], lj = lv[(long)j * D + d];
  const float ivj = expf(-lj), cj = -0.5f * (LOG2PI_W + lj);
  float gm = 0.f, gl = 0.f;
  for (int il = 0; il < Bl; ++il) {
    const int i = row0 + il;
    const float* rs = rowstats + (long)il * rstride;
    const float diff = z[(long)i * D + d] - mj;
    const float r = diff * ivj;
    const float ld = (cj - 0.5f * (diff * diff * ivj)) + log_w_ij_w(i, j, Bg, lN, lS, lM);
    const float G = cP * __expf(S[(long)il * Bg + j] - rs[1]) + cQ * __expf(ld - rs[4 + d]);
    gm += G * r;
    gl += G * (-0.5f + 0.5f * r * diff);
  }
  if (j >= row0 && j < row0 + Bl) {             // diagonal term alpha * log q(z_j|x_j) / B
    const float diff = z[(long)j * D + d] - mj;
    const float r = diff * ivj;
    gm += alpha * invB * r;
    gl += alpha * invB * (-0.5f + 0.5f * r * diff);
  }
  dmu[(long)j * D + d] = gm;
  dlv[(long)j * D + d] = gl;
}

// ---- launchers ------------------------------------------------------------------------------------------------------
int launch_reparam_kl_fwd_wide(const float* ml, const float* eps, float* mu, float* logvar, float* z, float* kl_dim,
                               const float* coef, int B, int D, hipStream_t s) {
  const long n = (long)B * D;
  hipLaunchKernelGGL(k_reparam_wide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ml, eps, mu, logvar, z, n, D);
  DVAE_CHECK_LAUNCH();
  if (kl_dim) {                                 // (coef != NULL checked by the entry point: no partial blocks above DVAE_MAX_D)
    hipLaunchKernelGGL(k_kl_cols_wide, dim3(D), dim3(256), 0, s, mu, logvar, B, D, coef, kl_dim);
    DVAE_CHECK_LAUNCH();
  }
  return 0;
}

int launch_btcvae_fwd_wide(const float* z, const float* mu, const float* lv, int Bg, int D, int row0, int Bl, int is_mss,
                           const float* log_w, float* tmp, float* rowstats, hipStream_t s) {
  float* S = tmp + (size_t)3 * D * Bg;          // (the column constants were written by k_btcvae_prep: loss.hip)
  hipLaunchKernelGGL(k_tcw_joint, dim3((Bg + 255) / 256, Bl), dim3(256), 0, s, z, tmp, Bg, D, row0, is_mss, log_w, S);
  DVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_tcw_rowstats, dim3(Bl), dim3(256), (size_t)D * sizeof(float), s, z, mu, lv, tmp, S, Bg, D, row0, is_mss,
                     log_w, rowstats, DVAE_ROWSTATS_STRIDE(D));
  DVAE_CHECK_LAUNCH();
  return 0;
}

int launch_btcvae_bwd_wide(const float* z, const float* mu, const float* lv, const float* rowstats, int Bg, int D, int row0,
                           int Bl, int is_mss, const float* log_w, const float* coef, const float* tmp, float* dz, float* dmu,
                           float* dlv, hipStream_t s) {
  const float* S = tmp + (size_t)3 * D * Bg;
  const int rstride = DVAE_ROWSTATS_STRIDE(D);
  hipLaunchKernelGGL(k_tcw_bwd_rows, dim3(Bl, (D + 3) / 4), dim3(256), 0, s, z, mu, lv, tmp, S, rowstats, rstride, Bg, D, row0,
                     is_mss, log_w, coef, dz);
  DVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_tcw_bwd_cols, dim3((Bg + 255) / 256, D), dim3(256), 0, s, z, mu, lv, S, rowstats, rstride, Bg, D, row0, Bl,
                     is_mss, log_w, coef, dmu, dlv);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
