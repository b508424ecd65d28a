#!/bin/bash
# rocprofv3 PMC passes over a short bench run (separate passes: TCC has 4 slots, FETCH_SIZE takes 3,
# WRITE_SIZE 2 -- MI355X_MICROARCH.md).  --pmc is only combined with --kernel-trace.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$REPO/gpurun_out/pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
# PMC_BENCH_ARGS: extra bench.py arguments (e.g. "--batch 128"); PMC_OUT: name of the summary (default pmc_summary.md);
# PMC_IMAGES: images per conv launch of that run (default 1024 = the default workload), written into the summary's header
CMD="python $REPO/bench.py ${PMC_BENCH_ARGS:-} --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
pass() {  # name counters...
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.log" 2>&1)
  echo "pass $name: rc=$? $(find "$OUT/$name" -name '*counter_collection.csv' | head -1)"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python "$REPO/tools/pmc_summary.py" "$OUT" "${PMC_IMAGES:-1024}" | tee "$REPO/gpurun_out/${PMC_OUT:-pmc_summary.md}"
