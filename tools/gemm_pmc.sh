#!/bin/bash
# PMC passes over the large-GEMM kernel alone (tools/gemm_one.py M K N reps [fwd|dgrad|wgrad]); environment switches of a debug
# build (DVAE_GDMA_VAR, DVAE_GDMA_ABLATE, DVAE_GEMM_DMA) pass through.  Prints per counter the mean over the launches, the
# kernel's duration from the kernel trace and the effective shader clock = GRBM_GUI_ACTIVE / duration.
#   bash tools/gemm_pmc.sh [M] [form] [label]
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
M=${1:-2048}; FORM=${2:-fwd}; LABEL=${3:-gemm}
OUT="$REPO/gpurun_out/gemm_pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $REPO/tools/gemm_one.py $M 1000 1000 8 $FORM"
pass() { local name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.log" 2>&1); echo "pass $name rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD
LABEL="$LABEL" python - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/gemm_pmc"
d = collections.defaultdict(list)
dur = collections.defaultdict(list)
name = None
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gdma" in k or "k_gemm_big" in k or "k_fcw32" in k:
            name = k.split("(")[0]
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gdma" in k or "k_gemm_big" in k or "k_fcw32" in k:
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
ns = [x for v in dur.values() for x in v]
ns = sorted(ns)[len(ns) // 2] if ns else 0
cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
print("%s: %s; %d samples; GRBM cycles %.0f; duration under the profiler %.1f us -> effective clock %.2f GHz" % (
    os.environ["LABEL"], name, len(d.get("GRBM_GUI_ACTIVE", [])), cyc, ns / 1e3, cyc / ns if ns else 0))
for k in sorted(m):
    print("  %-32s %.4g   per SIMD-cycle %.3f" % (k, m[k], m[k] / (cyc * 1024) if cyc else 0))
PY
