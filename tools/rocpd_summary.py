"""Text summary (per-kernel calls / total / average / share) of a rocprofv3 rocpd sqlite database."""
import re
import sqlite3
import sys


def main(path, top=60):
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


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
