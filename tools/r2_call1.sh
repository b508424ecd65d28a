#!/bin/bash
# round 2, GPU visit 1: full parity suite with error statistics, the experimental k_up32r2 (debug build), the new bench
# line for all four BASELINE configs, a kernel profile of the default bench.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== host: $(nproc) cpus"
echo "== pytest -m gpu (all, with parity stats)"
DVAE_PARITY_STATS=gpurun_out/parity_stats.json timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --no-header --durations=15 > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -60
grep -E "^E  " gpurun_out/pytest.log | cut -c1-260 | head -40
echo "== k_up32r2 correctness (DVAE_UP_R2=1)"
DVAE_UP_R2=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_sizes.py -m gpu -q --timeout=600 --no-header -k "convT_fwd or conv_fwd or persistent_loops or 4x4" > gpurun_out/pytest_r2.log 2>&1
tail -n 3 gpurun_out/pytest_r2.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r2.log | head
echo "== bench (default config)"
timeout 900 python bench.py --steps 100 --warmup 20 2>&1 | tail -n 1 > gpurun_out/bench_default.json; cut -c1-1500 gpurun_out/bench_default.json
echo "== A/B DVAE_UP_R2"
bash tools/ab.sh DVAE_UP_R2 2>&1 | tee gpurun_out/ab_r2.log
for c in factor_celeba btcvae_dsprites factor_dsprites; do
  echo "== bench --config $c"
  timeout 600 python bench.py --config $c --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 > gpurun_out/bench_$c.json; cut -c1-700 gpurun_out/bench_$c.json
done
echo "== small batches (strong-scaling shards of configs[3]/[4])"
for b in 128 256; do timeout 300 python bench.py --batch $b --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-parity-check 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae 3ch B=$b', d['value'], d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; done 2>&1 | tee gpurun_out/small_batch.log
echo "== rocprofv3 kernel stats"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
tail -n 2 gpurun_out/prof.log | cut -c1-300
python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 > gpurun_out/prof_summary.md; head -45 gpurun_out/prof_summary.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 12 gpurun_out/timeline.txt
