#!/bin/bash
# round 2, GPU visit 13: k_down32ws2 (weights in registers, output through LDS) parity + A/B (debug build: DVAE_DOWN_WS2=0);
# gate-matched smoke() and bench parity_check
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 2 | cut -c1-300
echo "== pytest (conv kernels, bench sizes, steps)"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_sizes.py tests/test_gpu_step.py -m gpu -q --timeout=900 --no-header -x -k "conv or step or persistent or 4x4" > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -20
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -20
echo "== kbench: k_down32ws2 (default) vs k_down32ws (DVAE_DOWN_WS2=0)"
timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "conv fwd|convT dgrad" | tee gpurun_out/kbench_dws2.txt
DVAE_DOWN_WS2=0 timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "conv fwd|convT dgrad" | tee -a gpurun_out/kbench_dws2.txt
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
for v in 1 0 1 0; do echo -n "DVAE_DOWN_WS2=$v: "; DVAE_DOWN_WS2=$v bench; done | tee -a gpurun_out/kbench_dws2.txt
echo "== parity_check of the factor configs (gate-matched)"
for c in factor_dsprites factor_celeba; do timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], json.dumps(d['parity_check']))"; done
