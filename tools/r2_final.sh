#!/bin/bash
# round 2, final visit: the SHIPPED (non-debug) library -- full parity suite, smoke, bench lines of all four BASELINE configs,
# rocprofv3 kernel stats + timeline, PMC passes.  Everything goes to gpurun_out/ (copied to profiles/r02_final_*).
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== weight-gradient kernels first (fail fast: nothing below is worth measuring if these are wrong)"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 --no-header -x -k "wgrad or reduce" > gpurun_out/pytest_wgrad.log 2>&1
rc=$?; tail -n 3 gpurun_out/pytest_wgrad.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/pytest_wgrad.log | cut -c1-300 | head -20; echo "ABORT: weight-gradient parity failed"; exit 1; fi
echo "== pytest -m gpu"
DVAE_PARITY_STATS=gpurun_out/parity_stats.json timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --no-header -x > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -20
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -20
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 2 | cut -c1-300
echo "== bench (default)"
timeout 900 python bench.py 2>&1 | tail -n 1 > gpurun_out/bench_default.json; cut -c1-400 gpurun_out/bench_default.json; echo
for c in factor_celeba btcvae_dsprites factor_dsprites; do
  timeout 600 python bench.py --config $c --no-roofline --no-cpu-baseline 2>&1 | tail -n 1 > gpurun_out/bench_$c.json; python -c "import json; d=json.load(open('gpurun_out/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['parity_check']['ok'])"
done
echo "== batch sweep"
for b in 64 128 256 512; do timeout 300 python bench.py --batch $b --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae 3ch B=$b', d['value'], d['ms_per_step'])"; done | tee gpurun_out/batch_sweep.txt
echo "== rocprofv3 kernel stats + timeline"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 > gpurun_out/prof_summary.md; head -20 gpurun_out/prof_summary.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 2 gpurun_out/timeline.txt
echo "== PMC passes"
bash tools/pmc_collect.sh > gpurun_out/pmc.log 2>&1; grep -E "k_up32ws<16|k_wgrad32ws<16|k_down32ws<16" gpurun_out/pmc_summary.md | awk -F'|' '{print $2, $(NF-3), $(NF-2), $(NF-1)}'
echo "== extras (records for the next round): every kernel alone at B = 1024 and B = 128; the B = 128 step as a timeline; factor kernel stats"
timeout 200 python tools/kbench.py 1024 > gpurun_out/kbench_final.txt 2>&1; grep -E "thin|reduction|partial|likelihood" gpurun_out/kbench_final.txt
timeout 120 python tools/kbench.py 128 > gpurun_out/kbench_b128.txt 2>&1; tail -n 3 gpurun_out/kbench_b128.txt
rm -rf gpurun_out/prof128
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof128" -o prof -- python "$REPO/bench.py" --batch 128 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof128.log" 2>&1)
python tools/timeline.py gpurun_out/prof128/prof_results.db > gpurun_out/timeline_b128.txt 2>&1; tail -n 1 gpurun_out/timeline_b128.txt
python tools/prof_summary.py gpurun_out/prof128/prof_results.db 35 > gpurun_out/prof128_summary.md 2>&1
rm -rf gpurun_out/prof_factor
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_factor" -o prof -- python "$REPO/bench.py" --config factor_celeba --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof_factor.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof_factor/prof_results.db 13 > gpurun_out/prof_factor_summary.md 2>&1; head -n 8 gpurun_out/prof_factor_summary.md
rm -rf gpurun_out/prof128/*.db.tmp
