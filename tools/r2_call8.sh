#!/bin/bash
# round 2, GPU visit 8: k_up32ws with register-resident weights + epilogue inside the next unit's MFMA stream
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== pytest (conv kernels + bench sizes + steps)"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_sizes.py tests/test_gpu_step.py -m gpu -q --timeout=900 --no-header -x -k "conv or step or persistent or 4x4" > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -20
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -20
echo "== k_up32ws<16> alone"
for a in 0 1 3 0; do DVAE_UPWS_ABLATE=$a timeout 120 python tools/upws_one.py 1024 2>&1 | tail -n 1; done | tee gpurun_out/upws_v2.txt
timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "convT fwd|conv dgrad" | tee -a gpurun_out/upws_v2.txt
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
echo "== bench"; bench; bench
