"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite db) as a short table.
    python tools/prof_summary.py prof_results.db [steps]      (steps omitted: the number of k_stage_weights launches)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("dvae::", "")
    if "multi_tensor_apply" in name:
        return "torch::multi_tensor_apply (Adam)"
    if "distribution_elementwise" in name:
        return "torch::randn/rand"
    if "vectorized_elementwise" in name or "elementwise_kernel" in name:
        return "torch::elementwise"
    return name[:60]


def main(db, steps):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average from top_kernels"))
    agg = {}
    for name, calls, total, avg in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += calls
        a[1] += total
    tot = sum(v[1] for v in agg.values())
    if steps <= 0:                 # one k_stage_weights launch per training iteration (settle + warm-up + timed steps)
        steps = max(1, agg.get("k_stage_weights", [1])[0])
    print("| kernel | calls | total us | avg us | %% | us/step (%d steps) |" % steps)
    print("|---|---|---|---|---|---|")
    for k, (calls, total) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.0f | %.1f | %.1f | %.1f |" % (k, calls, total, total / calls, 100 * total / tot, total / steps))
    print("| TOTAL | | %.0f | | 100 | %.1f |" % (tot, tot / steps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)     # 0: infer the step count
