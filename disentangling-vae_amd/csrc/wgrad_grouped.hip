// Grouped finishing pass of the conv / convT weight gradients of a training step (training.py:157): the eight
// weight-gradient launches (dvae_conv*_wgrad_partial) leave per-workgroup partial sums in their own workspaces; ONE launch
// reduces all of them (fixed order per output element, identical to the per-layer reduce kernels) instead of eight
// latency-bound ~10 us launches queued between the big kernels of the weight-gradient stream.
#include "common.h"
#include "wgrad_reduce.h"

namespace dvae {

struct WgrProb { const float* ws; float* dw; float* db; int kind, bias_from_big, nblk, blk0; };   // kind: 32 = 32<->32 channels, 1 / 3 = thin C
struct WgrTable { WgrProb p[DVAE_WGR_MAX]; int n; };

__global__ __launch_bounds__(256) void k_wgrad_reduce_grouped(const WgrTable t) {
  const int bid = blockIdx.x;
  const float* ws = t.p[0].ws; float* dw = t.p[0].dw; float* db = t.p[0].db;
  int kind = t.p[0].kind, bfb = t.p[0].bias_from_big, nblk = t.p[0].nblk, blk0 = 0;
#pragma unroll
  for (int q = 1; q < DVAE_WGR_MAX; ++q) {
    if (q < t.n && bid >= t.p[q].blk0) {
      ws = t.p[q].ws; dw = t.p[q].dw; db = t.p[q].db;
      kind = t.p[q].kind; bfb = t.p[q].bias_from_big; nblk = t.p[q].nblk; blk0 = t.p[q].blk0;
    }
  }
  const int blk = bid - blk0;
  if (kind == 32) wgrad32_reduce_body(blk, ws, dw, db, bfb, nblk);
  else if (kind == 3) wgrad_thin_reduce_body<3>(blk, ws, dw, db, bfb, nblk);
  else wgrad_thin_reduce_body<1>(blk, ws, dw, db, bfb, nblk);
}

// number of partial blocks the partial launchers produce (must mirror launch_wgrad_t / launch_wgrad_ws_t / launch_wgrad_thin)
int wgrad_partial_blocks(int kind, int N, int Hs) {
  if (kind == 32) {
    const long n_units = ((long)N * Hs * Hs + 63) / 64;
    return (int)(n_units < WG_MAX_BLOCKS ? n_units : WG_MAX_BLOCKS);
  }
  const long n_units = (long)N * 8;
  return (int)(n_units < WT_MAX_BLOCKS ? n_units : WT_MAX_BLOCKS);
}

int launch_wgrad_reduce_grouped(const dvae_conv_wgrad_desc* d, int n, hipStream_t s) {
  WgrTable t;
  memset(&t, 0, sizeof(t));
  t.n = n;
  int blocks = 0;
  for (int q = 0; q < n; ++q) {
    // geometry in "small / big" terms: conv: big = x (Cin, H), small = dy (Cout, H/2); convT: big = dy (Cout, 2H), small = x (Cin, H)
    const int Cb = d[q].transposed ? d[q].Cout : d[q].Cin;
    const int Cs = d[q].transposed ? d[q].Cin : d[q].Cout;
    const int Hs = d[q].transposed ? d[q].H : d[q].H / 2;
    int kind;
    if (Cs == 32 && Cb == 32 && (Hs == 4 || Hs == 8 || Hs == 16)) kind = 32;
    else if (Cs == 32 && (Cb == 1 || Cb == 3) && Hs == 32) kind = Cb;
    else { set_error("dvae_conv_wgrad_reduce_grouped: problem %d is not a tuned geometry", q); return -1; }
    WgrProb& p = t.p[q];
    p.ws = d[q].ws; p.dw = d[q].dw; p.db = d[q].db;
    p.kind = kind; p.bias_from_big = d[q].transposed ? 1 : 0;
    p.nblk = wgrad_partial_blocks(kind, d[q].N, Hs);
    p.blk0 = blocks;
    blocks += kind == 32 ? WG_REDUCE_BLOCKS : WT_REDUCE_BLOCKS(kind);
  }
  hipLaunchKernelGGL(k_wgrad_reduce_grouped, dim3(blocks), dim3(256), 0, s, t);
  DVAE_CHECK_LAUNCH();
  return 0;
}

}  // namespace dvae
