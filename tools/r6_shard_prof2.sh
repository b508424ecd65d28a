set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
rm -rf gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --config btcvae_celeba --shard-world 2 --steps 40 --warmup 10 --no-parity-check --no-roofline --shard-legs --shard-which rccl > "$REPO/gpurun_out/prof.log" 2>&1)
tail -n 1 gpurun_out/prof.log | cut -c1-600
python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/r06_s2_shard2_rccl_kernel_stats.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/r06_s2_shard2_rccl_timeline.md 2>&1
rm -rf gpurun_out/prof
cat gpurun_out/r06_s2_shard2_rccl_timeline.md
