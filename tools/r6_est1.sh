# beta-TCVAE estimator backward: a workgroup per row / column (shipped) against a wave per row / column (round 5: libdvae_hip_lossold.so)
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_ddp.py -m gpu -q --no-header -x -k "btcvae or sharded or mirrored or rccl" 2>&1 | tail -3
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
L=disentangling-vae_amd/lib
run() { if [ "$1" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$1.so; fi; python bench.py $2 $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 lib=$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
 for t in default lossold; do
  for w in "--batch 1024" "--batch 128" "--config btcvae_dsprites" "--batch 2048"; do run $t "$w"; done
 done
done
SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
for rep in 1 2 3; do for t in default lossold; do
  if [ "$t" = "default" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$t.so; fi
  python bench.py --config btcvae_celeba $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('shard-rccl lib=$t', d['transports']['rccl']['ms_per_step'])"; done; done
