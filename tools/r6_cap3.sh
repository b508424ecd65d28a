# the batch-sized weight-gradient grid (shipped) against one workgroup per CU at every batch (variant build): parity of the wgrad kernels, then steps
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_sizes.py -m gpu -q --no-header -x -k "wgrad or step or conv32" 2>&1 | tail -3
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
L=disentangling-vae_amd/lib
run() { if [ "$1" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$1.so; fi; python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 lib=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in default nocap; do
  for w in "--batch 128" "--batch 192" "--batch 256" "--batch 384" "--batch 512" "--batch 1024" "--config btcvae_dsprites" "--config factor_dsprites" "--config factor_celeba" "--config vae_mnist"; do run $t "$w"; done
 done
done
unset DVAE_HIP_LIB
for v in 2048 1000000; do for rep in 1 2 3; do DVAE_DEBUG=1 DVAE_THREE_STREAM_MIN_ROWS=$v python bench.py --config factor_celeba $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('factor_celeba three_min_rows=$v lib=x', d['ms_per_step'])"; done; done
