# shard legs (one rank of eight, mirrored world) with the round-6 host-side changes switched off one at a time
BA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
run() { DVAE_DEBUG=1 DVAE_FUSE_ENDS=$1 DVAE_EARLY_THIN=$2 python bench.py --config $3 $BA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$3 fuse_ends=$1 early_thin=$2 single', d['single_process']['ms_per_step']); print('$3 fuse_ends=$1 early_thin=$2 rccl', d['transports']['rccl']['ms_per_step'])"; }
for rep in 1 2 3; do
 for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v
  run $1 $2 factor_celeba; run $1 $2 btcvae_celeba
 done
done
