// Entropy estimator behind the MIG / AAM disentanglement metrics (disvae/evaluate.py:233-297,
// Evaluator._estimate_latent_entropies):
//     H(z_d) = 1/S sum_s [ log N - logsumexp_{n=1..N} log N(z[d,s]; mean[n,d], exp(logvar[n,d])) ]
// for S = 10 000 sampled latents and the whole data set (N = 737 280 for dSprites, or one of its
// conditional slices): N x D x S Gaussian log-densities -- 7.4e13 for one marginal entropy, six times that
// for the metric -- and the same logsumexp kernel family as the beta-TCVAE estimator (loss.hip).
//
// Decomposition: thread = one (d, s) pair, workgroup = 256 consecutive s of one d, gridDim.z = chunks of the data set.
// The per-n constants (mean, -0.5 (log 2pi + logvar), exp(-logvar): utils/math.py:48-50) are wave-uniform, prepared
// once in a transposed [3][D][N] image and fetched with scalar loads; the online logsumexp handles 8 data points
// per rescale (9 v_exp_f32 per 8 densities).  Chunk partials (max, sum) are merged in a fixed order.
#include "common.h"

namespace dvae {

#define LOG2PI_M 1.8378770664093453f
#define ENT_CHUNK 16384          // data points per workgroup
#define ENT_MAX_CHUNKS 64

__global__ void k_entropy_prep(const float* __restrict__ mu, const float* __restrict__ lv, long N, int D,
                               float* __restrict__ tmp) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= N * D) return;
  const long n = idx / D; const int d = (int)(idx % D);
  const float l = lv[idx];
  tmp[(long)d * N + n] = mu[idx];
  tmp[(long)(D + d) * N + n] = -0.5f * (LOG2PI_M + l);
  tmp[(long)(2 * D + d) * N + n] = expf(-l);
}

__global__ __launch_bounds__(256) void k_entropy_lse(const float* __restrict__ z_ds, const float* __restrict__ tmp, long N,
                                                     int D, int S, int chunk, float* __restrict__ part) {
  const int d = blockIdx.y;
  const int s = blockIdx.x * 256 + threadIdx.x;
  const long n0 = (long)blockIdx.z * chunk;
  const long n1 = n0 + chunk < N ? n0 + chunk : N;
  const float* muT = tmp + (long)d * N;
  const float* cT = tmp + (long)(D + d) * N;
  const float* ivT = tmp + (long)(2 * D + d) * N;
  const float z = s < S ? z_ds[(long)d * S + s] : 0.f;
  float m = -INFINITY, acc = 0.f;
  long n = n0;
  for (; n + 8 <= n1; n += 8) {
    float v[8];
    float mx = m;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float diff = z - muT[n + u];
      v[u] = cT[n + u] - 0.5f * (diff * diff * ivT[n + u]);        // log_density_gaussian, utils/math.py:48-50
      mx = fmaxf(mx, v[u]);
    }
    // every density of the block (and everything before it) can be -inf (exp(-logvar) overflowing): shift by 0 then, so
    // that exp(-inf - shift) = 0 instead of exp(-inf + inf) = NaN; torch.logsumexp returns -inf for such a column too
    const float sh = mx > -INFINITY ? mx : 0.f;
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += __expf(v[u] - sh);
    acc = acc * __expf(m - sh) + t;
    m = mx;
  }
  for (; n < n1; ++n) {
    const float diff = z - muT[n];
    const float v = cT[n] - 0.5f * (diff * diff * ivT[n]);
    const float mx = fmaxf(m, v);
    const float sh = mx > -INFINITY ? mx : 0.f;
    acc = acc * __expf(m - sh) + __expf(v - sh);
    m = mx;
  }
  if (s < S) {
    float* p = part + (((long)blockIdx.z * D + d) * S + s) * 2;
    p[0] = m; p[1] = acc;
  }
}

// merge the chunk partials of every (d, s) in chunk order, then H[d] = 1/S sum_s (log N - lse[d,s]) (fixed order)
__global__ __launch_bounds__(256) void k_entropy_finish(const float* __restrict__ part, int chunks, long N, int D, int S,
                                                        float* __restrict__ lse, float* __restrict__ H) {
  __shared__ float red[4];
  const int d = blockIdx.x;
  const float logN = logf((float)N);
  float hs = 0.f;
  for (int s = threadIdx.x; s < S; s += 256) {
    float m = -INFINITY, acc = 0.f;
    for (int c = 0; c < chunks; ++c) {
      const float* p = part + (((long)c * D + d) * S + s) * 2;
      const float m2 = p[0], a2 = p[1];
      if (m2 > m) { acc = acc * __expf(m - m2) + a2; m = m2; }
      else if (m2 > -INFINITY) { acc += a2 * __expf(m2 - m); }
    }
    const float l = m + logf(acc);
    lse[(long)d * S + s] = l;
    hs += logN - l;                       // -log q(z_d) = log N - logsumexp_n log q(z_d | x_n)   (evaluate.py:286-289)
  }
  float v = wave_sum(hs);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) H[d] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)S;
}

static int entropy_chunks(long N) {
  long c = (N + ENT_CHUNK - 1) / ENT_CHUNK;
  return (int)(c < 1 ? 1 : (c > ENT_MAX_CHUNKS ? ENT_MAX_CHUNKS : c));
}

size_t latent_entropy_ws_floats(long N, int D, int S) {
  return (size_t)3 * D * N + (size_t)entropy_chunks(N) * D * S * 2 + (size_t)D * S;
}

int launch_latent_entropy(const float* z_ds, const float* mean, const float* logvar, long N, int D, int S, float* ws,
                          float* H, hipStream_t s) {
  float* tmp = ws;
  float* part = ws + (size_t)3 * D * N;
  const int chunks = entropy_chunks(N);
  float* lse = part + (size_t)chunks * D * S * 2;
  const int chunk = (int)(((N + chunks - 1) / chunks + 7) / 8 * 8);
  const long nd = N * D;
  hipLaunchKernelGGL(k_entropy_prep, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, mean, logvar, N, D, tmp);
  DVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_entropy_lse, dim3((S + 255) / 256, D, chunks), dim3(256), 0, s, z_ds, tmp, N, D, S, chunk, part);
  DVAE_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_entropy_finish, dim3(D), dim3(256), 0, s, part, chunks, N, D, S, lse, H);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
