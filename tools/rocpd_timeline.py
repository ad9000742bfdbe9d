"""GPU occupancy of a rocprofv3 kernel trace (rocpd sqlite): how much of the wall time has at least one kernel
running, how much has two or more (member streams), and how large the idle gaps are.  Looks at the LAST `frac` of the
trace (the steady-state steps).  Usage: python tools/rocpd_timeline.py <db> [frac=0.5]"""
import sqlite3
import sys


def main(path, frac=0.5):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t] or [t for t in tabs if "kernel_dispatch" in t]
    rows = []
    for t in disp:
        cols = [r[1] for r in c.execute("pragma table_info('%s')" % t)]
        if "start" in cols and "end" in cols:
            rows = c.execute("select start, end from '%s'" % t).fetchall()
            if rows:
                break
    if not rows:
        print("no dispatch table with start/end found in", tabs[:20])
        return
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t1 - (t1 - t0) * frac
    ev = []
    for s, e in rows:
        if e <= lo:
            continue
        ev.append((max(s, lo), 1))
        ev.append((e, -1))
    ev.sort()
    depth, last = 0, lo
    busy1 = busy2 = 0
    gaps = []
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth == 0 and t > last:
            gaps.append(t - last)
        depth += d
        last = t
    span = t1 - lo
    n = sum(1 for s, e in rows if e > lo)
    print("window %.1f ms, %d kernels: >=1 kernel running %.1f %%, >=2 running %.1f %%, idle %.1f %%"
          % (span / 1e6, n, 100.0 * busy1 / span, 100.0 * busy2 / span, 100.0 * (span - busy1) / span))
    gaps.sort(reverse=True)
    tot = sum(gaps)
    print("idle gaps: %d, total %.2f ms; > 100 us: %d (%.2f ms); 20-100 us: %d (%.2f ms); < 20 us: %d (%.2f ms)"
          % (len(gaps), tot / 1e6, sum(g > 1e5 for g in gaps), sum(g for g in gaps if g > 1e5) / 1e6,
             sum(2e4 < g <= 1e5 for g in gaps), sum(g for g in gaps if 2e4 < g <= 1e5) / 1e6,
             sum(g <= 2e4 for g in gaps), sum(g for g in gaps if g <= 2e4) / 1e6))
    print("sum of kernel durations in window %.1f ms (%.2fx the window)" % (sum(e - max(s, lo) for s, e in rows if e > lo) / 1e6,
          sum(e - max(s, lo) for s, e in rows if e > lo) / span))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
