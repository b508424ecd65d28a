#!/bin/bash
# round 2, GPU visit 3: full parity suite (uint8 pipeline, MIG/AAM metrics, RCCL C-ABI, latent dims, mixed paths), bench line,
# PMC passes.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== pytest -m gpu"
DVAE_PARITY_STATS=gpurun_out/parity_stats.json timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --no-header --durations=12 -s > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|dvae_latent_entropy N=" gpurun_out/pytest.log | head -40
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -40
grep -A14 "slowest" gpurun_out/pytest.log | head -16
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 3 | cut -c1-300
echo "== bench (default, full line)"
timeout 900 python bench.py --steps 100 --warmup 20 2>&1 | tail -n 1 > gpurun_out/bench_default.json; cut -c1-900 gpurun_out/bench_default.json
echo "== bench --force-ddp (1 rank): torch vs rccl transport"
for t in torch rccl; do timeout 300 python bench.py --steps 60 --warmup 15 --force-ddp --transport $t --no-cpu-baseline --no-roofline --no-parity-check 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'])"; done
echo "== uint8 input: bench with a uint8 batch"
timeout 300 python tools/bench_u8.py 2>&1 | tail -n 4
echo "== PMC passes"
bash tools/pmc_collect.sh > gpurun_out/pmc.log 2>&1; tail -n 3 gpurun_out/pmc.log | cut -c1-200
