#!/bin/bash
# Round-6 final visit: the whole GPU suite, smoke, the bench line as the driver runs it (20 / 5), kernel statistics + timelines of the
# default workload, of 128 images and of btcvae_dsprites, PMC passes.      gpurun --timeout 2400 -- 'TAG=r06_final2 bash tools/r6_final_visit.sh'
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
TAG=${TAG:-r06_final2}
echo "== host: $(nproc) cpus; $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Series')"
echo "== pytest -m gpu"
timeout 1300 python -m pytest tests -m gpu -q --timeout=300 --no-header > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/${TAG}_pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | head -40
echo "---- first failure detail"; grep -n -m1 -A14 "^E  " gpurun_out/${TAG}_pytest.log | cut -c1-300
[ -f gpurun_out/parity_stats.json ] && cp gpurun_out/parity_stats.json gpurun_out/${TAG}_parity_stats.json
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 6 | cut -c1-400 | tee gpurun_out/${TAG}_smoke.log
echo "== bench (driver's flags)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit: $?"
tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read())
print("value", d["value"], "ms", d["ms_per_step"], "hip-event median", d["hip_event_ms_per_step"]["median"], "parity", d["parity_check"]["ok"])
r = d["roofline"]; print("roofline", r["kernel"], r.get("achieved"), r["frac"])
for k in d.get("roofline_kernels", []): print("  ", k.get("kernel"), k.get("launch", ""), k.get("us"), k.get("bound"), k.get("frac"))
print("drop_in", d["drop_in"]["ms_per_step"], d["drop_in"]["over_timed_configuration"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"])
for c in d["configs"]:
    if "single_process" in c:
        print("  cfg", c["name"], "single", c["single_process"]["ms_per_step"], {k: v["ms_per_step"] for k, v in c["transports"].items()}, c.get("parity_check", {}).get("ok"))
    else:
        print("  cfg", c["name"], c["value"], c["ms_per_step"], c["step_frac_of_fp32_peak"], c["parity_check"]["ok"], c["cpu_baseline"]["value"])
PY
BA="--no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
prof() {  # name, bench arguments
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" $2 --steps 40 --warmup 10 $BA > "$REPO/gpurun_out/prof.log" 2>&1)
  python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_$1_kernel_stats.md
  python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_$1_timeline.md 2>&1
  echo "-- $1"; head -14 gpurun_out/${TAG}_$1_kernel_stats.md; tail -n 2 gpurun_out/${TAG}_$1_timeline.md
  rm -rf gpurun_out/prof
}
echo "== rocprofv3 kernel stats + timelines"
prof b1024 ""
prof b128 "--batch 128"
prof dsprites "--config btcvae_dsprites"
echo "== PMC passes (1024 images)"
PMC_OUT=${TAG}_pmc_summary.md bash tools/pmc_collect.sh > gpurun_out/${TAG}_pmc.log 2>&1; tail -n 40 gpurun_out/${TAG}_pmc.log | cut -c1-260
rm -rf gpurun_out/pmc
echo "== done"
