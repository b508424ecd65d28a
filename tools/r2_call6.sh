#!/bin/bash
# round 2, GPU visit 6: grouped conv-wgrad reduction, k_up32ws store/load reordering, k_wgrad32ws two tiles in flight; PMC
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== pytest -m gpu"
DVAE_PARITY_STATS=gpurun_out/parity_stats.json timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --no-header -x > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -40
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -30
echo "== kbench"
timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "convT fwd|conv dgrad|conv fwd|convT dgrad|conv wgrad|thin" | tee gpurun_out/kbench.log
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
echo "== bench A/B: deferred (grouped) conv reductions"
for v in 1 0 1 0; do echo -n "DVAE_DEFER_REDUCE=$v: "; DVAE_DEFER_REDUCE=$v bench; done
echo -n "factor_celeba: "; bench --config factor_celeba
echo -n "btcvae_dsprites: "; bench --config btcvae_dsprites
echo -n "factor_dsprites: "; bench --config factor_dsprites
for b in 128 256; do echo -n "btcvae 3ch B=$b: "; bench --batch $b --steps 200 --warmup 30; done
echo "== rocprofv3 kernel stats + timeline"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 > gpurun_out/prof_summary.md; head -30 gpurun_out/prof_summary.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 3 gpurun_out/timeline.txt
echo "== PMC passes"
bash tools/pmc_collect.sh > gpurun_out/pmc.log 2>&1; grep -E "k_up32ws|k_wgrad32ws|k_down32ws<16" gpurun_out/pmc_summary.md | cut -c1-40,330-420
