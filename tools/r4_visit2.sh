#!/bin/bash
# round-4 visit: the full bench line + profiles (kernel stats, timelines, PMC) of the default and the factor_celeba workloads
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
TAG=${TAG:-r04_v13}
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit: $?"
tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "hip-event", d["hip_event_ms_per_step"]["segments"], "parity", d.get("parity_check", {}).get("ok"), "settle", d.get("settle"))
    r = d.get("roofline")
    if r: print("roofline", r["kernel"], r["us_per_launch"], r["frac"], r.get("in_step_us"), r.get("frac_in_step"), r["traffic"])
    for r in d.get("roofline_kernels", []): print("  ", r["kernel"], r.get("launch", ""), r["us_per_launch"], r["bound"], r["frac"], r.get("in_step_us"), r.get("traffic"))
    if "drop_in" in d: print("drop_in", d["drop_in"]["ms_per_step"])
    if "cpu_baseline" in d: print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"])
    for c in d.get("configs", []):
        print("  cfg", c["name"], c["value"], c["ms_per_step"], c["step_frac_of_fp32_peak"], c.get("parity_check", {}).get("ok"), c.get("cpu_baseline", {}).get("value"))
        for r in c.get("roofline_kernels", []): print("      ", r["launch"], r["us_per_launch"], r["frac"])
    print("timing", d.get("timing_s"), d.get("bench_wall_s"))
except Exception as e:
    print("bench line unreadable:", e); print(open("gpurun_out/${TAG}_bench.log").read()[-3000:])
PY
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for c in btcvae_celeba factor_celeba; do
  echo "== rocprofv3 kernel stats: $c"
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --config $c --steps 10 --warmup 3 $BA > "$REPO/gpurun_out/prof.log" 2>&1)
  python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_${c}_kernel_stats.md; head -n 14 gpurun_out/${TAG}_${c}_kernel_stats.md
  python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_${c}_timeline.md 2>&1; tail -n 2 gpurun_out/${TAG}_${c}_timeline.md
  rm -rf gpurun_out/prof
done
echo "== PMC passes"
PMC_OUT=${TAG}_pmc_summary.md bash tools/pmc_collect.sh > gpurun_out/${TAG}_pmc.log 2>&1
PMC_BENCH_ARGS="--config factor_celeba" PMC_OUT=${TAG}_factor_pmc_summary.md bash tools/pmc_collect.sh > gpurun_out/${TAG}_factor_pmc.log 2>&1
grep -E "k_up32ws<16|k_gdma|thin" gpurun_out/${TAG}_factor_pmc_summary.md | cut -c1-60 | head; rm -rf gpurun_out/pmc
echo "== done"
