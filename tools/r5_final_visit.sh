#!/bin/bash
# Round-5 final visit: the whole GPU suite, smoke, the bench line (default = 200 steps, and as the driver runs it: 20 / 5),
# kernel statistics + timeline of the default workload.     gpurun --timeout 1500 -- 'bash tools/r5_final_visit.sh'
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
TAG=${TAG:-r05_final3}
echo "== pytest -m gpu"
timeout 1100 python -m pytest tests -m gpu -q --timeout=300 --no-header > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/${TAG}_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | head -40
echo "---- first failure detail"; grep -n -m1 -A14 "^E  " gpurun_out/${TAG}_pytest.log | cut -c1-300
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 6 | cut -c1-400 | tee gpurun_out/${TAG}_smoke.log
echo "== bench (default)"
timeout 600 python bench.py 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench.json; cut -c1-400 gpurun_out/${TAG}_bench.json
echo "== bench (driver's flags)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -n 1 > gpurun_out/${TAG}_bench_driver_like.json; cut -c1-300 gpurun_out/${TAG}_bench_driver_like.json
echo "== rocprofv3 kernel stats + timeline"
rm -rf gpurun_out/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_kernel_stats.md; head -12 gpurun_out/${TAG}_kernel_stats.md
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_timeline.md 2>&1; tail -n 2 gpurun_out/${TAG}_timeline.md
rm -rf gpurun_out/prof
