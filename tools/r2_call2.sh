#!/bin/bash
# round 2, GPU visit 2: grouped FC wgrad + gate-matched step parity; per-kernel A/B of k_up32r2; persistent-grid caps of the
# side-stream weight-gradient kernels (debug build); profile + timeline.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== pytest -m gpu (all, with parity stats)"
DVAE_PARITY_STATS=gpurun_out/parity_stats.json timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --no-header > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -40
grep -E "^E  " gpurun_out/pytest.log | cut -c1-260 | head -30
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
echo "== bench default (grouped FC wgrad)"; bench; bench
echo "== persistent-grid caps of the side-stream wgrad kernels"
for cfg in "256 512" "240 480" "224 448" "192 384" "224 512" "256 448"; do
  set -- $cfg
  echo -n "DVAE_WGRAD_GRID=$1 DVAE_WGRAD_THIN_GRID=$2: "; DVAE_WGRAD_GRID=$1 DVAE_WGRAD_THIN_GRID=$2 bench
done
echo "== factor / dsprites / small batches"
for c in factor_celeba btcvae_dsprites factor_dsprites; do echo -n "$c: "; bench --config $c; done
for b in 128 256; do echo -n "btcvae 3ch B=$b: "; bench --batch $b --steps 200 --warmup 30; done
echo "== kbench (k_up32 vs k_up32r2)"
timeout 300 python tools/kbench.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/kbench.log; cat gpurun_out/kbench.log
DVAE_UP_R2=1 timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "convT fwd|conv dgrad" | tee gpurun_out/kbench_r2.log
echo "== rocprofv3 kernel stats (default)"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 > gpurun_out/prof_summary.md; head -30 gpurun_out/prof_summary.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 3 gpurun_out/timeline.txt
echo "== rocprofv3 timeline with caps 224/448"
rm -rf gpurun_out/prof_cap
(cd /tmp && DVAE_WGRAD_GRID=224 DVAE_WGRAD_THIN_GRID=448 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof_cap" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof_cap.log" 2>&1)
python tools/timeline.py gpurun_out/prof_cap/prof_results.db > gpurun_out/timeline_cap.txt 2>&1; tail -n 3 gpurun_out/timeline_cap.txt
