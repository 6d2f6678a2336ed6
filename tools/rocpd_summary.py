"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the per-kernel stats table kept under profiles/.

  python tools/rocpd_summary.py trace.db out.txt [--last-ms X]

--last-ms X: only the dispatches that START within the last X milliseconds of the trace.  bench.py's timed region is the last
steps * ms_per_step milliseconds before its final synchronise (warm-up steps and the BN-statistics calibration of the joint workload come
before it), so X = steps * ms_per_step of the same run gives per-step figures: divide `calls` and `total_ms` by `steps`."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main(db, out=None, last_ms=None):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)").fetchall()]
    where, note = "", ""
    if last_ms is not None:
        if "start" not in cols or "end" not in cols:
            raise SystemExit("kernels view has no start/end columns (%s)" % cols)
        t_end = con.execute("select max(end) from kernels").fetchone()[0]
        where = " where start >= %d" % int(t_end - float(last_ms) * 1e6)
        note = " — dispatches starting in the last %.1f ms of the trace (the timed region)" % float(last_ms)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels" + where +
                       " group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)%s" % (db, note),
             "# total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)),
             "%-112s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct")]
    for name, n, s, a, mn, mx in rows:
        lines.append("%-112s %7d %12.3f %10.1f %10.1f %10.1f %6.2f" % (short(name), n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    a = sys.argv[1:]
    last = None
    if "--last-ms" in a:
        i = a.index("--last-ms")
        last = float(a[i + 1])
        del a[i:i + 2]
    main(a[0], a[1] if len(a) > 1 else None, last)
