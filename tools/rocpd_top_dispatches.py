"""The longest individual dispatches of a rocprofv3 kernel trace (rocpd sqlite), optionally only kernels whose name does NOT
match a pattern (default: the MFMA convolutions): which single launches of the bandwidth passes are expensive.
Usage: python tools/rocpd_top_dispatches.py <db> [top=40] [exclude-regex]"""
import re
import sqlite3
import sys


def main(path, top=40, excl=r"conv_|head_fwd"):
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


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else r"conv_|head_fwd")
