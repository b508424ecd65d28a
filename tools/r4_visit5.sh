#!/bin/bash
# round 4, last profile refresh: factor_celeba kernel stats / timeline / PMC, the B = 128 timeline, kbench -- with the wave-specialised
# thin kernels and the late joins
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
TAG=${TAG:-r04_final2}
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
echo "== rocprofv3 kernel stats: factor_celeba"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --config factor_celeba --steps 10 --warmup 3 $BA > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_factor_celeba_kernel_stats.md; head -n 14 gpurun_out/${TAG}_factor_celeba_kernel_stats.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_factor_celeba_timeline.md 2>&1; tail -n 2 gpurun_out/${TAG}_factor_celeba_timeline.md
rm -rf gpurun_out/prof
TAG=$TAG STEPS="timeline kbench" bash tools/visit.sh
PMC_BENCH_ARGS="--config factor_celeba" PMC_OUT=${TAG}_factor_pmc_summary.md bash tools/pmc_collect.sh > gpurun_out/${TAG}_factor_pmc.log 2>&1
grep -E "k_up32ws<16|k_gdma|thin" gpurun_out/${TAG}_factor_pmc_summary.md | cut -c1-60 | head; rm -rf gpurun_out/pmc
echo "== done"
