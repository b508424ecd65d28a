#!/bin/bash
# round 2, GPU visit 9: do LDS-light FC kernels (k_gemm32 / k_gemm instead of k_fc32 / k_gemm_big) escape the starvation behind the
# persistent weight-gradient kernels?  (debug build switches DVAE_GEMM_FC=0, DVAE_GEMM_BIG=0)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
for cfg in "1 1" "0 1" "0 0" "1 1" "0 1"; do set -- $cfg; echo -n "DVAE_GEMM_FC=$1 DVAE_GEMM_BIG=$2: "; DVAE_GEMM_FC=$1 DVAE_GEMM_BIG=$2 bench; done | tee gpurun_out/fc_light.txt
echo "== timeline default"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 > gpurun_out/prof_summary.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 2 gpurun_out/timeline.txt
echo "== timeline DVAE_GEMM_FC=0"
rm -rf gpurun_out/prof_fc0
(cd /tmp && DVAE_GEMM_FC=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_fc0" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof_fc0.log" 2>&1)
python tools/timeline.py gpurun_out/prof_fc0/prof_results.db > gpurun_out/timeline_fc0.txt 2>&1; tail -n 2 gpurun_out/timeline_fc0.txt
