#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
for m in 1 2 3 1; do echo "DVAE_UPWS_OUTBITS=$m"; DVAE_UPWS_OUTBITS=$m timeout 200 python tools/kbench.py 1024 2>&1 | grep -E "bits: conv"; done | tee gpurun_out/r04_v20_outbits_abl.txt
