#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Run via
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh'
# Everything worth keeping goes to gpurun_out/.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== host: $(nproc) cpus"
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q --timeout=300 --no-header ${PYTEST_ARGS:-} > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -60
echo "---- first failure detail"; grep -n -m1 -A12 "^E  " gpurun_out/pytest.log | cut -c1-300
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 12 | cut -c1-400 | tee gpurun_out/smoke.log
if [ "${DO_DEBUG:-0}" = "1" ]; then echo "== debug_step"; timeout 300 python tools/debug_step.py 8 2>&1 | tail -n 80 | tee gpurun_out/debug_step.log; fi
if [ "${DO_KBENCH:-0}" = "1" ]; then echo "== kbench"; timeout 300 python tools/kbench.py 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kbench.log; echo "== kbench (DVAE_DOWN_V1)"; DVAE_DOWN_V1=1 timeout 300 python tools/kbench.py 1024 2>&1 | grep -E "conv fwd|convT dgrad" | tee gpurun_out/kbench_v1.log; fi
echo "== bench"
timeout 600 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 2>&1 | tail -n 3 | tee gpurun_out/bench.log
if [ "${DO_PROF:-1}" = "1" ]; then
  echo "== rocprofv3 kernel stats"
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-parity-check > "$REPO/gpurun_out/prof.log" 2>&1)
  tail -n 3 gpurun_out/prof.log
  python tools/prof_summary.py gpurun_out/prof/prof_results.db 13 | head -40
fi
