# estimator backward, both passes in one launch (shipped) against two launches (libdvae_hip_lossold.so)
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_ddp.py -m gpu -q --no-header -x -k "btcvae or sharded or mirrored or rccl" 2>&1 | tail -3
L=disentangling-vae_amd/lib
SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do for t in default lossold; do
  if [ "$t" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$t.so; fi
  for w in 8 4; do python bench.py --config btcvae_celeba --shard-world $w $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('one rank of $w lib=$t single', d['single_process']['ms_per_step']); print('one rank of $w lib=$t rccl', d['transports']['rccl']['ms_per_step'])"; done
  python bench.py --config btcvae_dsprites $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae_dsprites lib=$t x', d['ms_per_step'])"
done; done
