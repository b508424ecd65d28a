# thin ends at any size (k_down_thin_px / k_up_thin_px): parity, then the 32x32 config against the plain generic kernels' build
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -m gpu -q --no-header -x 2>&1 | tail -3
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
L=disentangling-vae_amd/lib
run() { if [ "$1" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$1.so; fi; python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 lib=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do for t in default plain; do run $t "--config vae_mnist"; done; done
unset DVAE_HIP_LIB
export TMPDIR=/tmp; REPO=$(pwd)
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --config vae_mnist --steps 60 --warmup 10 $BA > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/r06_s2_mnist_kernel_stats.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/r06_s2_mnist_timeline.md 2>&1
rm -rf gpurun_out/prof
cat gpurun_out/r06_s2_mnist_kernel_stats.md | head -36; tail -3 gpurun_out/r06_s2_mnist_timeline.md
