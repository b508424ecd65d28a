// torch.optim.Adam's element-wise update of a list of tensors as ONE launch (main.py:208 / losses.py:238 build the optimizers;
// training.py:158 and losses.py:307-308 step them).
//
// The optimizer object, its param groups, hyper-parameters and state tensors (step, exp_avg, exp_avg_sq: what
// optimizer.state_dict() saves) stay torch's; only the arithmetic of step() runs here.  Why: through torch, the step of the
// 0.5 M parameters of the VAE costs the host 100-150 us of Python per iteration (per-parameter state gathering, grouping by
// device and dtype, two multi-tensor launches: profiles/r05_v10_host_profile.txt) and the GPU 17-22 us -- at the 128 images
// per GPU of the 8-GPU headline configuration the iteration is 0.38 ms and the host needs 0.32 ms to issue it.  Here: one
// foreign call, one launch of ~4 us.
//
// Arithmetic = adam.py's single-tensor path in fp32 (amsgrad off, maximize off, L2 weight decay added to the gradient):
//   g' = g + wd p;  m = m + (1 - b1)(g' - m);  v = b2 v + (1 - b2) g' g';  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - b1^t, bc2 = 1 - b2^t evaluated in double on the host.  Within 1 ulp of the parameter of torch's own CPU and
// fused implementations on the same gradients (tests/test_gpu_timed_config.py).
#include "common.h"

namespace dvae {

#define ADAM_MAX_T 64             // tensors per launch: the by-value table is 64 x 48 B + 65 x 4 B + 4 B = 3336 B (+ 32 B of scalars),
                                  // inside the classic 4 KB kernel-argument block (asserted below)
#define ADAM_CHUNK 4096           // elements per workgroup

struct AdamTable {
  dvae_adam_tensor t[ADAM_MAX_T];
  int blk0[ADAM_MAX_T + 1];       // first workgroup of tensor i; blk0[nt] = grid size
  int nt;
};

static_assert(sizeof(AdamTable) + 64 <= 4096, "k_adam's by-value table must fit a 4 KB kernel-argument block");

__global__ __launch_bounds__(256) void k_adam(const AdamTable T, float step_new, float lr_over_bc1, float rsqrt_bc2_inv,
                                              float b1c, float b2, float b2c, float eps, float wd) {
  // which tensor: wave-uniform search over <= ADAM_MAX_T entries
  int ti = 0;
  const int b = blockIdx.x;
  while (ti + 1 < T.nt && T.blk0[ti + 1] <= b) ++ti;
  const dvae_adam_tensor e = T.t[ti];
  const long off = (long)(b - T.blk0[ti]) * ADAM_CHUNK;
  if (off == 0 && threadIdx.x == 0 && e.step) *e.step = step_new;
  const long n = e.n - off < ADAM_CHUNK ? e.n - off : ADAM_CHUNK;
  float* __restrict__ p = e.p + off;
  const float* __restrict__ g = e.g + off;
  float* __restrict__ m = e.m + off;
  float* __restrict__ v = e.v + off;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    if (wd != 0.f) gg += wd * pp;
    mm = mm + b1c * (gg - mm);
    vv = b2 * vv + b2c * gg * gg;
    const float denom = sqrtf(vv) * rsqrt_bc2_inv + eps;
    pp = pp - lr_over_bc1 * (mm / denom);
  };
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec) {
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += blockDim.x) {
      f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
      const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = pp[j], c = mm[j], d = vv[j];
        upd(a, gg[j], c, d);
        pp[j] = a; mm[j] = c; vv[j] = d;
      }
      reinterpret_cast<f32x4*>(p)[i] = pp;
      reinterpret_cast<f32x4*>(m)[i] = mm;
      reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) upd(p[i], g[i], m[i], v[i]);
  } else {
    for (long i = threadIdx.x; i < n; i += blockDim.x) upd(p[i], g[i], m[i], v[i]);
  }
}

int launch_adam(const dvae_adam_tensor* ts, int nt, float step_new, double lr, double beta1, double beta2, double eps,
                double weight_decay, hipStream_t s) {
  // bias corrections as adam.py computes them (python floats = doubles), then the per-element arithmetic in fp32
  const double bc1 = 1.0 - pow(beta1, (double)step_new), bc2 = 1.0 - pow(beta2, (double)step_new);
  const float lr_over_bc1 = (float)(lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  for (int t0 = 0; t0 < nt; t0 += ADAM_MAX_T) {
    AdamTable T;
    T.nt = nt - t0 < ADAM_MAX_T ? nt - t0 : ADAM_MAX_T;
    int blk = 0;
    for (int i = 0; i < T.nt; ++i) {
      T.t[i] = ts[t0 + i];
      T.blk0[i] = blk;
      blk += (int)((ts[t0 + i].n + ADAM_CHUNK - 1) / ADAM_CHUNK);
    }
    T.blk0[T.nt] = blk;
    if (blk == 0) continue;
    hipLaunchKernelGGL(k_adam, dim3(blk), dim3(256), 0, s, T, step_new, lr_over_bc1, inv_sqrt_bc2, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay);
    DVAE_CHECK_LAUNCH();
  }
  return 0;
}

}  // namespace dvae
