#!/bin/bash
# round 2, GPU visit 14 (last minutes of the budget): timing ablations of k_down32ws<16> (debug build, conv2 forward, B = 1024).
# DVAE_ABLATE bits: 1 loaders do not write LDS, 2 loaders do not load tiles, 4 no output stores, 8 no MFMAs,
#                   16 no LDS operand reads in the MFMA loop, 32 no per-unit barrier.  Results of ablated launches are invalid.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for a in 0 4 1 3 7 16 23 32 55 8 15 0; do DVAE_ABLATE=$a timeout 60 python tools/kone.py 2>&1 | tail -n 1; done | tee gpurun_out/downws_ablation.txt
