#!/bin/bash
# round 2, GPU visit 15 (last minutes): timing ablations of k_wgrad32ws<16> (debug build, conv2 weight gradient, B = 1024, partial sums only).
# DVAE_WGWS_ABLATE bits: 1 loaders do not write LDS, 2 no tile loads, 8 no MFMAs, 16 no LDS operand reads, 32 no per-unit barrier.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for a in 0 1 3 16 19 32 51 8 11 0; do DVAE_WGWS_ABLATE=$a timeout 60 python tools/wgws_one.py 2>&1 | tail -n 1; done | tee gpurun_out/wgws_ablation.txt
# and the masked down kernel (convT2 dgrad) with the same switches as visit 14
