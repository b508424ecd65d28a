#!/bin/bash
# round 2, GPU visit 5: k_wgrad32ws (transposed-LDS, wave-specialised wgrad) parity + A/B (debug build: DVAE_WGRAD_WS=0), uint8 conversion at LDS-store time
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== pytest -m gpu"
DVAE_PARITY_STATS=gpurun_out/parity_stats.json timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --no-header -x > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -40
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -30
echo "== kbench: k_up32ws (default) vs k_up32 (DVAE_WGRAD_WS=0)"
timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "convT fwd|conv dgrad|conv fwd|convT dgrad|conv wgrad" | tee gpurun_out/kbench_wgws.log
DVAE_WGRAD_WS=0 timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "conv wgrad" | tee gpurun_out/kbench_nowgws.log
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
echo "== bench A/B"
for v in 1 0 1 0; do echo -n "DVAE_WGRAD_WS=$v: "; DVAE_WGRAD_WS=$v bench; done
echo -n "factor_celeba: "; bench --config factor_celeba
echo -n "btcvae_dsprites: "; bench --config btcvae_dsprites
echo "== uint8 batch A/B"
timeout 300 python tools/bench_u8.py 2>&1 | tail -n 4
echo "== bench (full line, roofline of the new kernel)"
timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/bench_wgws.json; python -c "import json; d=json.load(open('gpurun_out/bench_ws.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'])[:500]); [print(json.dumps(r)[:300]) for r in d['roofline_kernels']]"
echo "== rocprofv3 kernel stats + timeline"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 > gpurun_out/prof_summary.md; head -24 gpurun_out/prof_summary.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 3 gpurun_out/timeline.txt
