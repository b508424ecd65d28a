SA="--steps 100 --warmup 20 --no-parity-check --no-roofline --shard-legs --shard-which single,rccl"
for rep in 1 2; do for w in 2 4; do for r in auto plan; do DVAE_DEBUG=1 DVAE_REPLAY=$r python bench.py --config btcvae_celeba --shard-world $w $SA 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('world=$w replay=$r single', d['single_process']['ms_per_step'], 'host', d['single_process'].get('host_issue_ms_per_step')); print('world=$w replay=$r rccl', d['transports']['rccl']['ms_per_step'], 'host', d['transports']['rccl'].get('host_issue_ms_per_step'))"; done; done; done
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2; do for r in auto plan; do for b in 512 1024; do DVAE_DEBUG=1 DVAE_REPLAY=$r python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single-process B=$b replay=$r', d['ms_per_step'], 'host', d.get('host_issue_ms_per_step'))"; done; done; done
