SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl,torch"
for w in 8 4 2; do for c in btcvae_celeba factor_celeba; do python bench.py --config $c --shard-world $w $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$c one rank of $w: single', d['single_process']['ms_per_step'], {k: v['ms_per_step'] for k, v in d['transports'].items()})"; done; done
