# the sharded btcvae step as one rank of 4 / 2 (256 / 512 images per rank): the small-step policies with the FC-gradient placement switched off / on
SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
for rep in 1 2 3; do for w in 8 4 2; do for f in 1 0; do DVAE_DEBUG=1 DVAE_FCW_MAIN=$f python bench.py --config btcvae_celeba --shard-world $w $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('world=$w fcw_main=$f single', d['single_process']['ms_per_step']); print('world=$w fcw_main=$f rccl', d['transports']['rccl']['ms_per_step'])"; done; done; done
