#!/bin/bash
# round 2, GPU visit 7: where does k_up32ws<16> spend its time (timing ablations, debug build)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for a in 0 1 2 4 8 16 32 33 12 24 40 56 63 0; do DVAE_UPWS_ABLATE=$a timeout 120 python tools/upws_one.py 1024 2>&1 | tail -n 1; done | tee gpurun_out/upws_ablate.txt
echo "== other batch sizes (no ablation)"
for b in 256 512 2048; do timeout 120 python tools/upws_one.py $b 2>&1 | tail -n 1; done | tee -a gpurun_out/upws_ablate.txt
