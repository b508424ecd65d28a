# btcvae_dsprites (64x64x1, B = 256) and factor_dsprites as timelines + kernel statistics (rocprofv3 kernel trace)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
for cfg in btcvae_dsprites factor_dsprites; do
  rm -rf gpurun_out/prof_$cfg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_$cfg" -o prof -- python "$REPO/bench.py" --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in > "$REPO/gpurun_out/prof_$cfg.log" 2>&1)
  tail -n 1 gpurun_out/prof_$cfg.log | cut -c1-200
  python tools/prof_summary.py gpurun_out/prof_$cfg/prof_results.db > gpurun_out/${cfg}_kernel_stats.md
  python tools/timeline.py gpurun_out/prof_$cfg/prof_results.db > gpurun_out/${cfg}_timeline.md 2>&1
  tail -n 2 gpurun_out/${cfg}_timeline.md
  rm -rf gpurun_out/prof_$cfg
done
