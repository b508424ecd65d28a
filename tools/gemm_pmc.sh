#!/bin/bash
# PMC passes over the large-GEMM kernel alone (tools/gemm_one.py)
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$REPO/gpurun_out/gemm_pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $REPO/tools/gemm_one.py 2048 1000 1000 6"
pass() { local name=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.log" 2>&1); echo "pass $name rc=$?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass sq3 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA
python - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/gemm_pmc"
d = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm_big" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in d.items()}
cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
print("k_gemm_big: %d samples; kernel cycles %.0f (%.1f us @2.4GHz)" % (len(d.get("GRBM_GUI_ACTIVE", [])), cyc, cyc / 2400))
for k in sorted(m):
    print("  %-32s %.4g   per SIMD-cycle %.3f" % (k, m[k], m[k] / (cyc * 1024) if cyc else 0))
PY
