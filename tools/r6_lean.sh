# k_wgrad32_reduce with ~42 VGPRs (WGR_LEAN variant: fits beside k_up32ws<8> / k_down32dma on a CU) against the shipped 74
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -x -k "wgrad or conv_fwd_dgrad" 2>&1 | tail -2
L=disentangling-vae_amd/lib
DVAE_HIP_LIB=$L/libdvae_hip_lean.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_sizes.py -m gpu -q --no-header -x -k "wgrad or conv_fwd_dgrad or step" 2>&1 | tail -2
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
run() { if [ "$1" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$1.so; fi; python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 lib=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do for t in default lean; do for w in "--batch 1024" "--batch 512" "--batch 128" "--config factor_celeba" "--config btcvae_dsprites"; do run $t "$w"; done; done; done
