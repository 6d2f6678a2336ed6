"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the per-kernel stats table kept under profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                       "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % db, "# total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)),
             "%-112s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for name, n, s, a, mn, mx in rows:
        lines.append("%-112s %7d %12.3f %10.1f %10.1f %10.1f %6.2f" % (short(name), n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
