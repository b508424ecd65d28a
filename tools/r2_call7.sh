#!/bin/bash
# round 2, GPU visit 7: where does k_up32ws<16> spend its time (timing ablations, debug build)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for a in 0 33 97 289 353 161 417 481 33 0; do DVAE_UPWS_ABLATE=$a timeout 120 python tools/upws_one.py 1024 2>&1 | tail -n 1; done | tee gpurun_out/upws_ablate.txt
echo "== other batch sizes (no ablation)"
for b in 4096; do timeout 120 python tools/upws_one.py $b 2>&1 | tail -n 1; done | tee -a gpurun_out/upws_ablate.txt
