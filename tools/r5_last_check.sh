# last sanity visit on the final tree: smoke, the bench line as the driver runs it, a quick subset of the GPU suite
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -n 2 | cut -c1-300 | tee gpurun_out/r05_final4_smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -n 1 > gpurun_out/r05_final4_bench_driver_like.json; cut -c1-260 gpurun_out/r05_final4_bench_driver_like.json
timeout 200 python -m pytest tests/test_gpu_wide_latent.py tests/test_gpu_fused_core.py tests/test_gpu_adam.py -q --timeout=120 --no-header 2>&1 | tail -n 2 | tee gpurun_out/r05_final4_pytest_subset.txt
