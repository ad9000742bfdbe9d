"""Text reports of a rocprofv3 kernel trace (rocpd sqlite database), as tools/prof_bench.sh writes them into profiles/:
    python tools/rocpd_report.py summary <db> [top=60]          per-kernel calls / total / average / share
    python tools/rocpd_report.py timeline <db> [frac=0.5]       GPU occupancy of the last `frac` of the trace: >= 1 / >= 2 kernels running, idle gaps
    python tools/rocpd_report.py top <db> [n=40] [exclude-regex] the longest individual dispatches that are NOT convolutions
    python tools/rocpd_report.py alone <db> [frac=0.4 | window ms] [ms_per_step] per kernel: the wall time during which it was the ONLY kernel running
                                                                (what a multi-stream step actually waits for), idle gaps by predecessor
"""
import re
import sqlite3
import sys


def summary(path, top=60):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats : %s" % path)
    print("# total kernel time %.3f ms over %d kernels" % (tot / 1e3 if tot > 1e6 else tot / 1e3, len(rows)))
    print("%-92s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for name, calls, total, avg, pct in rows[:top]:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(cg_conv_geom.*", "", name)[:92]
        print("%-92s %8d %12.1f %10.2f %7.2f" % (name, calls, total, avg, pct))



def timeline(path, frac=0.5):
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



def top(path, top=40, excl=r"conv_|head_fwd"):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    t = next((t for t in tabs if t == "kernels"), None) or next(t for t in tabs if "kernel" in t and "dispatch" in t)
    cols = [r[1] for r in c.execute("pragma table_info('%s')" % t)]
    namec = next(cn for cn in ("name", "kernel_name", "kernel") if cn in cols)
    gc = [cn for cn in ("grid_x", "grid_size_x", "grid_size") if cn in cols]
    q = "select %s, start, end%s from '%s'" % (namec, (", " + gc[0]) if gc else "", t)
    rows = c.execute(q).fetchall()
    t1 = max(r[2] for r in rows)
    t0 = min(r[1] for r in rows)
    lo = t1 - (t1 - t0) * 0.4
    sel = [(r[2] - r[1], r[0], r[3] if gc else 0) for r in rows if r[1] >= lo and not re.search(excl, r[0])]
    sel.sort(reverse=True)
    print("# longest non-convolution dispatches of the last 40 %% of the trace (%d dispatches, %.2f ms in total)" % (len(sel), sum(s[0] for s in sel) / 1e6))
    for d, name, g in sel[:top]:
        name = re.sub(r"\(anonymous namespace\)::", "", name)[:80]
        print("%9.1f us  grid %-9s %s" % (d / 1e3, g, name))



def _named_dispatches(c):
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    t = next((t for t in tabs if t == "kernels"), None) or next(t for t in tabs if "kernel" in t and "dispatch" in t)
    cols = [r[1] for r in c.execute("pragma table_info('%s')" % t)]
    namec = next(cn for cn in ("name", "kernel_name", "kernel") if cn in cols)
    return c.execute("select %s, start, end from '%s'" % (namec, t)).fetchall()


def alone(path, frac=0.4, ms_per_step=0.0):
    """Sweep over the dispatch intervals of the last `frac` of the trace: time with exactly one kernel running is credited to
    that kernel, an idle gap to the kernel that ended last before it."""
    rows = _named_dispatches(sqlite3.connect(path))
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    lo = t1 - ((t1 - t0) * frac if frac <= 1.0 else frac * 1e6)      # frac > 1: the window in milliseconds
    ev = []
    for i, (name, s, e) in enumerate(rows):
        if e > lo:
            ev.append((max(s, lo), 1, i))
            ev.append((e, -1, i))
    ev.sort(key=lambda x: (x[0], x[1]))
    running, last, last_ended = set(), lo, None
    solo, shared, gap, calls = {}, {}, {}, {}
    short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", n)).replace("void ", "")[:70]
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            if len(running) == 1:
                k = short(rows[next(iter(running))][0])
                solo[k] = solo.get(k, 0) + dt
            elif len(running) > 1:
                for j in running:
                    k = short(rows[j][0])
                    shared[k] = shared.get(k, 0) + dt / len(running)
            elif last_ended is not None:
                k = short(rows[last_ended][0])
                gap[k] = gap.get(k, 0) + dt
        if d > 0:
            running.add(i)
            k = short(rows[i][0])
            calls[k] = calls.get(k, 0) + 1
        else:
            running.discard(i)
            last_ended = i
        last = t
    span = t1 - lo
    unit = "ms per step (window / %.2f steps of %.3f ms)" % (span / 1e6 / ms_per_step, ms_per_step) if ms_per_step else "ms over the window"
    print("# last %.1f ms of the trace; alone = only kernel running, shared = its 1/n share of overlapped time," % (span / 1e6))
    print("# gap = idle time that followed it.  alone + shared + gap over all kernels = the window.  Columns in %s." % unit)
    names = sorted(set(solo) | set(shared) | set(gap), key=lambda k: -(solo.get(k, 0) + gap.get(k, 0)))
    sc = (span / ms_per_step) if ms_per_step else 1e6
    print("%-70s %7s %9s %9s %9s" % ("kernel", "calls", "alone", "shared", "gap"))
    for k in names[:70]:
        print("%-70s %7d %9.3f %9.3f %9.3f" % (k, calls.get(k, 0), solo.get(k, 0) / sc, shared.get(k, 0) / sc, gap.get(k, 0) / sc))
    print("%-70s %7d %9.3f %9.3f %9.3f" % ("TOTAL", sum(calls.values()), sum(solo.values()) / sc, sum(shared.values()) / sc,
                                           sum(gap.values()) / sc))


if __name__ == "__main__":
    cmd, path, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
    if cmd == "summary":
        summary(path, int(rest[0]) if rest else 60)
    elif cmd == "timeline":
        timeline(path, float(rest[0]) if rest else 0.5)
    elif cmd == "top":
        top(path, int(rest[0]) if rest else 40, rest[1] if len(rest) > 1 else r"conv_|head_fwd")
    elif cmd == "alone":
        alone(path, float(rest[0]) if rest else 0.4, float(rest[1]) if len(rest) > 1 else 0.0)
    else:
        raise SystemExit(__doc__)
