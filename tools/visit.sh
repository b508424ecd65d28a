#!/bin/bash
# One GPU-box visit, parameterised (replaces the per-visit scripts of rounds 1-2).  Run via
#   gpurun --timeout 1500 -- 'STEPS="probe fused tests smoke bench sweep kbench timeline" bash tools/visit.sh'
# STEPS (any subset, in this order):
#   probe     MFMA 4x4x1 lane-layout probe (tools/ubench/mfma4x4_probe)
#   fccab     FC-chain instantiations side by side (debug builds)
#   fused     the round-3 kernel tests only (tests/test_gpu_fused_core.py + FUSED_EXTRA)
#   tests     the whole -m gpu suite (PYTEST_ARGS to narrow it)
#   smoke     __graft_entry__.smoke()
#   bench     python bench.py (BENCH_ARGS; default run = headline + drop-in + the other three configs + CPU baseline)
#   sweep     short bench lines at B = 64 / 128 / 256 / 1024 (btcvae 3ch) and the dsprites / factor configs
#   absweep   A/B of the host-side schedule knobs (DVAE_DEBUG=1) over the batch size
#   ab2       alternating A/B runs (needs a --debug build for the grid caps)
#   kbench    every kernel alone at B = 1024 and B = 128
#   prof      rocprofv3 --kernel-trace --stats of the default workload + summary + timeline
#   timeline  the B = 128 step as a timeline (rocprofv3 kernel trace)
#   pmc       PMC passes (tools/pmc_collect.sh)
# Everything worth keeping goes to gpurun_out/ (merged back into the repo's gpurun_out/ by gpurun).
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
TAG=${TAG:-visit}
STEPS=${STEPS:-"tests smoke bench"}
has() { case " $STEPS " in *" $1 "*) return 0;; *) return 1;; esac; }
echo "== host: $(nproc) cpus; $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' || true)"
if has probe; then echo "== probe"; timeout 120 tools/ubench/mfma4x4_probe 2>&1 | tail -n 14 | tee gpurun_out/${TAG}_probe.txt; fi
if has fccab; then
  echo "== FC chain variants (debug build: DVAE_FCC_VARIANT = 10 * ring depth + contraction split)"
  for v in 81 161 82 162; do DVAE_FCC_VARIANT=$v timeout 120 python tools/fcc_ab.py 128 1024 2>&1 | grep variant; done | tee gpurun_out/${TAG}_fcc_ab.txt
fi
if has fcccold; then
  echo "== FC chain variants with cold caches (debug build)"
  for v in 82 162 82 162; do FCC_COLD=1 DVAE_FCC_VARIANT=$v timeout 120 python tools/fcc_ab.py 128 1024 2>&1 | grep variant; done | tee gpurun_out/${TAG}_fcc_cold.txt
fi
if has fused; then
  echo "== pytest (round-3 kernels)"
  timeout 900 python -m pytest tests/test_gpu_fused_core.py ${FUSED_EXTRA:-} -m gpu -q --timeout=300 --no-header > gpurun_out/${TAG}_pytest_fused.log 2>&1
  FUSED_RC=$?
  echo "pytest exit: $FUSED_RC" | tee -a gpurun_out/${TAG}_pytest_fused.log
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest_fused.log | cut -c1-260 | head -40
  grep -n -m3 -A6 "^E  " gpurun_out/${TAG}_pytest_fused.log | cut -c1-300 | head -40
  if [ "$FUSED_RC" != "0" ] && [ -n "${FCC_FALLBACK:-}" ]; then
    echo "== round-3 kernel tests failed with the default FC-chain variant: continuing with DVAE_FCC_VARIANT=$FCC_FALLBACK"
    export DVAE_FCC_VARIANT=$FCC_FALLBACK
    timeout 900 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --timeout=300 --no-header 2>&1 | tail -n 3
  elif [ "$FUSED_RC" != "0" ] && [ "${STOP_ON_FUSED_FAIL:-0}" = "1" ]; then echo "== round-3 kernel tests failed: skipping the remaining steps"; STEPS="${FAIL_STEPS:-}"; fi
fi
if has tests; then
  echo "== pytest -m gpu"
  DVAE_PARITY_STATS=gpurun_out/${TAG}_parity_stats.json timeout 1500 python -m pytest tests -m gpu -q --timeout=300 --no-header ${PYTEST_ARGS:-} > gpurun_out/${TAG}_pytest.log 2>&1
  echo "pytest exit: $?" | tee -a gpurun_out/${TAG}_pytest.log
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_pytest.log | cut -c1-260 | head -60
  echo "---- first failure detail"; grep -n -m2 -A10 "^E  " gpurun_out/${TAG}_pytest.log | cut -c1-300
fi
if has smoke; then echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -n 6 | cut -c1-400 | tee gpurun_out/${TAG}_smoke.log; fi
if has bench; then
  echo "== bench"
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/${TAG}_bench.log 2>&1; echo "bench exit: $?"
  tail -n 1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench.json"))
    print("value", d["value"], "ms", d["ms_per_step"], "hip-event median", d["hip_event_ms_per_step"]["median"], "parity", d.get("parity_check", {}).get("ok"))
    r = d.get("roofline")
    if r: print("roofline", r["kernel"], r["us_per_launch"], r["frac"])
    for r in d.get("roofline_kernels", []): print("  ", r["kernel"], r.get("launch", ""), r["us_per_launch"], r["bound"], r["frac"])
    if "drop_in" in d: print("drop_in", d["drop_in"]["ms_per_step"], d["drop_in"]["over_timed_configuration"])
    if "cpu_baseline" in d: print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"])
    for c in d.get("configs", []): print("  cfg", c["name"], c["value"], c["ms_per_step"], c["step_frac_of_fp32_peak"], c.get("parity_check", {}).get("ok"), c.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench line unreadable:", e); print(open("gpurun_out/${TAG}_bench.log").read()[-1500:])
PY
fi
if has sweep; then
  echo "== sweep"
  : > gpurun_out/${TAG}_sweep.txt
  for b in 64 128 256 512 1024; do
    timeout 200 python bench.py --batch $b --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('btcvae 3ch B=$b', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_sweep.txt
  done
  for c in factor_celeba btcvae_dsprites factor_dsprites; do
    timeout 200 python bench.py --config $c --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check --no-drop-in 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_sweep.txt
  done
fi
if has absweep; then
  echo "== schedule A/B (DVAE_DEBUG=1 knobs): weight-gradient schedule (eager = dependency-driven, batch = batch-sized), streams"
  line() { python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t); print('$1', d['value'], d['ms_per_step'])
except Exception:
    print('$1 FAILED:', t[-300:])"; }
  BA="--steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
  for b in 128 256 512 1024; do
    DVAE_DEBUG=1 DVAE_EAGER_WGRAD_ELEMS=0 timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "B=$b batch-sized"
    DVAE_DEBUG=1 DVAE_EAGER_WGRAD_ELEMS=999999999 timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "B=$b eager"
  done | tee gpurun_out/${TAG}_absweep.txt
  for b in 64 128; do
    DVAE_DEBUG=1 DVAE_STREAMS=1 timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "B=$b one stream"
    DVAE_DEBUG=1 DVAE_STREAMS=2 timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "B=$b two streams"
  done | tee -a gpurun_out/${TAG}_absweep.txt
  for c in btcvae_dsprites factor_dsprites; do
    DVAE_DEBUG=1 DVAE_EAGER_WGRAD_ELEMS=0 timeout 200 python bench.py --config $c --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check --no-drop-in 2>&1 | tail -n 1 | line "$c batch-sized"
    DVAE_DEBUG=1 DVAE_EAGER_WGRAD_ELEMS=999999999 timeout 200 python bench.py --config $c --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check --no-drop-in 2>&1 | tail -n 1 | line "$c eager"
  done | tee -a gpurun_out/${TAG}_absweep.txt
fi
if has ab2; then
  echo "== alternating A/B (debug build + DVAE_DEBUG=1): schedule, persistent-grid caps, host floor, DDP path"
  line() { python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t); print('$1', d['value'], d['ms_per_step'])
except Exception:
    print('$1 FAILED:', t[-300:])"; }
  BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
  {
  for rep in 1 2 3; do
    for b in 128 1024; do
      DVAE_DEBUG=1 DVAE_EAGER_WGRAD_ELEMS=0 timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "rep$rep B=$b batch-sized"
      DVAE_DEBUG=1 DVAE_EAGER_WGRAD_ELEMS=999999999 timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "rep$rep B=$b eager"
    done
    DVAE_WGRAD_GRID=192 DVAE_WGRAD_THIN_GRID=192 timeout 200 python bench.py $BA 2>&1 | tail -n 1 | line "rep$rep B=1024 wgrad grids capped at 192"
    DVAE_WGRAD_GRID=224 DVAE_WGRAD_THIN_GRID=224 timeout 200 python bench.py $BA 2>&1 | tail -n 1 | line "rep$rep B=1024 wgrad grids capped at 224"
  done
  timeout 200 python bench.py --batch 8 $BA 2>&1 | tail -n 1 | line "B=8 (host / launch-latency floor)"
  timeout 200 python bench.py --batch 128 --force-ddp $BA 2>&1 | tail -n 1 | line "B=128 --force-ddp (torch transport)"
  timeout 200 python bench.py --batch 128 --force-ddp --transport rccl $BA 2>&1 | tail -n 1 | line "B=128 --force-ddp --transport rccl"
  timeout 200 python bench.py --batch 128 $BA 2>&1 | tail -n 1 | line "B=128 single process"
  timeout 200 python bench.py --force-ddp $BA 2>&1 | tail -n 1 | line "B=1024 --force-ddp (torch transport)"
  timeout 200 python bench.py $BA 2>&1 | tail -n 1 | line "B=1024 single process"
  } | tee gpurun_out/${TAG}_ab2.txt
fi
if has dmaabl; then
  echo "== k_down32dma timing ablations (debug build)"
  ( for i in 1 2 3 4 5 6; do sleep 4; rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk" | head -2; done ) > gpurun_out/${TAG}_clocks.txt 2>&1 &
  timeout 300 python tools/down_abl.py 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_down_abl.txt
  wait
  cat gpurun_out/${TAG}_clocks.txt | head -12
fi
if has ddp; then
  echo "== data-parallel code path on one rank (--force-ddp), both transports"
  line() { python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t); print('$1', d['value'], d['ms_per_step'])
except Exception:
    print('$1 FAILED:', t[-300:])"; }
  BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
  {
  for rep in 1 2; do
    for b in 128 1024; do
      timeout 200 python bench.py --batch $b $BA 2>&1 | tail -n 1 | line "rep$rep B=$b single process"
      timeout 200 python bench.py --batch $b --force-ddp $BA 2>&1 | tail -n 1 | line "rep$rep B=$b --force-ddp (torch.distributed)"
      timeout 200 python bench.py --batch $b --force-ddp --transport rccl $BA 2>&1 | tail -n 1 | line "rep$rep B=$b --force-ddp --transport rccl"
    done
  done
  } | tee gpurun_out/${TAG}_ddp.txt
fi
if has kbench; then
  echo "== kbench"
  timeout 300 python tools/kbench.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kbench.txt; grep -E "staged|fc_chain|stage_w|thin|likelihood" gpurun_out/${TAG}_kbench.txt
  timeout 200 python tools/kbench.py 128 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_kbench_b128.txt; grep -E "staged|fc_chain|stage_w" gpurun_out/${TAG}_kbench_b128.txt
fi
if has prof; then
  echo "== rocprofv3 kernel stats"
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in > "$REPO/gpurun_out/prof.log" 2>&1)
  python tools/prof_summary.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_kernel_stats.md; head -n 30 gpurun_out/${TAG}_kernel_stats.md
  python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/${TAG}_timeline.md 2>&1; tail -n 2 gpurun_out/${TAG}_timeline.md
  rm -rf gpurun_out/prof
fi
if has timeline; then
  echo "== B = 128 timeline"
  rm -rf gpurun_out/prof128
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof128" -o prof -- python "$REPO/bench.py" --batch 128 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in > "$REPO/gpurun_out/prof128.log" 2>&1)
  python tools/timeline.py gpurun_out/prof128/prof_results.db > gpurun_out/${TAG}_timeline_b128.md 2>&1; tail -n 70 gpurun_out/${TAG}_timeline_b128.md
  python tools/prof_summary.py gpurun_out/prof128/prof_results.db 35 > gpurun_out/${TAG}_b128_kernel_stats.md 2>&1
  rm -rf gpurun_out/prof128
fi
if has pmc; then
  echo "== PMC passes"
  bash tools/pmc_collect.sh > gpurun_out/${TAG}_pmc.log 2>&1; cp gpurun_out/pmc_summary.md gpurun_out/${TAG}_pmc_summary.md 2>/dev/null
  grep -E "k_up32ws<16|k_wgrad32ws<16|k_down32dma<16|thin" gpurun_out/${TAG}_pmc_summary.md | cut -c1-200
fi
echo "== done"
