"""Timeline of ONE training iteration from a rocprofv3 --kernel-trace run (rocpd sqlite db):
every kernel of the last complete iteration (iterations are delimited by the step's first launch: k_stage_weights,
k_set_coef before round 3) with its queue,
start offset, duration and the idle gap in front of it on the same queue, plus per-queue busy time
and the union busy time (any queue active)."""
import sqlite3
import sys
from prof_summary import short


def main(db, which=-2):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, queue_id, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "k_stage_weights" in r[0]] or [i for i, r in enumerate(rows) if "k_set_coef" in r[0]]
    a, b = marks[which], marks[which + 1]
    step = rows[a:b]
    t0 = step[0][2]
    last_end = {}
    busy = {}
    print("| t0 us | dur us | gap us | q | kernel |")
    print("|---|---|---|---|---|")
    for name, q, s, e in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
        print("| %.1f | %.1f | %.1f | %d | %s |" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, short(name)))
    # union of busy intervals
    iv = sorted((s, e) for _, _, s, e in step)
    tot, cs, ce = 0.0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    tot += ce - cs
    wall = (rows[b][2] - t0) / 1e3
    print("\niteration wall %.1f us; any-queue busy %.1f us; idle %.1f us; per-queue busy: %s" %
          (wall, tot / 1e3, wall - tot / 1e3, {q: round(v, 1) for q, v in busy.items()}))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -2)
