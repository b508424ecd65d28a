set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --no-header -x -k "fc_chain" 2>&1 | tail -2
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for cfg in factor_celeba factor_dsprites; do
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --config $cfg --steps 40 --warmup 10 $BA > "$REPO/gpurun_out/prof.log" 2>&1)
  python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/r06_s2_${cfg}_kernel_stats.md
  python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/r06_s2_${cfg}_timeline.md 2>&1
  rm -rf gpurun_out/prof
done
cat gpurun_out/r06_s2_factor_celeba_timeline.md
