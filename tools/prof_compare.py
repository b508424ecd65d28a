"""Side-by-side per-kernel totals of two rocprofv3 --kernel-trace --stats runs (rocpd sqlite)."""
import sqlite3
import sys
from prof_summary import short


def load(db):
    agg = {}
    for name, calls, total, avg in sqlite3.connect(db).execute("select name, total_calls, total_duration, average from top_kernels"):
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += calls
        a[1] += total
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
print("| kernel | calls A | us A | calls B | us B | B-A us |")
print("|---|---|---|---|---|---|")
for k in sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, [0, 0])[1] - a.get(k, [0, 0])[1])):
    ca, ta = a.get(k, [0, 0.0]); cb, tb = b.get(k, [0, 0.0])
    print("| %s | %d | %.0f | %d | %.0f | %+.0f |" % (k, ca, ta, cb, tb, tb - ta))
print("| TOTAL | | %.0f | | %.0f | %+.0f |" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values()),
                                              sum(v[1] for v in b.values()) - sum(v[1] for v in a.values())))
