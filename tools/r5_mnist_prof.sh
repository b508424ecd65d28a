# kernel statistics of the 32x32x1 plumbing config (BASELINE configs[0]: VAE loss, B = 64) under rocprofv3
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out; rm -rf gpurun_out/prof_mnist
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_mnist" -o prof -- python "$REPO/bench.py" --config vae_mnist --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in > "$REPO/gpurun_out/prof_mnist.log" 2>&1)
tail -n 2 gpurun_out/prof_mnist.log | cut -c1-300
python tools/prof_summary.py gpurun_out/prof_mnist/prof_results.db 60 | tee gpurun_out/mnist_kernel_stats.md | head -60
rm -rf gpurun_out/prof_mnist
