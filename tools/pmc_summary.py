"""Summarise rocprofv3 --pmc csv passes: per kernel (mean over dispatches) counters and derived
HBM traffic.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced
streaming reads by 2x (MI355X_MICROARCH.md section HBM) -> reported both raw and x2-corrected."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("dvae::", "")
    return name[:44]


def load(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def main(root, images=None):
    if images:      # bench.py scales traffic by (its images per launch) / this
        print("images_per_launch: %d  (every conv / thin kernel row below is one launch over that many images)\n" % int(images))
    data = defaultdict(dict)
    for p in sorted(os.listdir(root)):
        d = os.path.join(root, p)
        if os.path.isdir(d):
            for k, cs in load(d).items():
                for c, vals in cs.items():
                    data[k][c] = sum(vals) / len(vals)
    cols = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS",
            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
            "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F32", "FETCH_SIZE", "WRITE_SIZE"]
    keys = [k for k in data if k.startswith("k_")]
    keys.sort(key=lambda k: -data[k].get("GRBM_GUI_ACTIVE", 0))
    print("| kernel | " + " | ".join(cols) + " | fetch MB (x2 corr) | write MB | mfma_busy/gui |")
    print("|" + "---|" * (len(cols) + 4))
    for k in keys:
        v = data[k]
        f, w = v.get("FETCH_SIZE", float("nan")), v.get("WRITE_SIZE", float("nan"))
        gui = v.get("GRBM_GUI_ACTIVE", float("nan"))
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (check:
        # a 32-channel conv kernel at 92.7 us x 2.4 GHz = 2.2e5 cycles vs GUI 1.76e6): utilisation = busy / (gui / 8 * 1024)
        print("| %s | " % k + " | ".join("%.3g" % v.get(c, float("nan")) for c in cols) +
              " | %.1f | %.1f | %.3f |" % (2 * f * 1024 / 1e6, w * 1024 / 1e6, mf / (gui / 8 * 1024) if gui else float("nan")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.environ.get("PMC_IMAGES"))
