"""Summarise a rocprofv3 (rocpd sqlite) run.
  python tools/rocpd_summary.py <results.db> [header line ...]          per-kernel calls / total / avg / min / max (--stats as text)
  python tools/rocpd_summary.py --pmc <results.db> [header line ...]    per-kernel, per-counter: dispatches, sum, per-dispatch average
"""
import sqlite3
import sys


def kernel_stats(db, headers):
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    dur = "duration" if "duration" in cols else '("end" - start)'
    rows = db.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    for h in headers:
        print("# " + h)
    print(f"# total kernel time {total / 1e6:.3f} ms")
    print(f"{'kernel':<100}{'calls':>8}{'total_ms':>15}{'pct':>9}{'avg_us':>15}{'min_us':>15}{'max_us':>15}")
    for r in rows:
        print(f"{r[0][:96]:<100}{r[1]:>8}{r[2] / 1e6:>15.3f}{100.0 * r[2] / total:>8.2f}%{r[3] / 1e3:>15.2f}{r[4] / 1e3:>15.2f}{r[5] / 1e3:>15.2f}")


def pmc_stats(db, headers):
    rows = db.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), max(value), sum(duration) from counters_collection "
                      "group by kernel_name, counter_name order by 4 desc").fetchall()
    for h in headers:
        print("# " + h)
    print(f"{'kernel':<90}{'counter':>14}{'dispatches':>11}{'sum':>20}{'avg/dispatch':>18}{'max':>18}{'kernel_ms':>12}")
    for r in rows:
        print(f"{r[0][:86]:<90}{r[1]:>14}{r[2]:>11}{r[3]:>20.1f}{r[4]:>18.2f}{r[5]:>18.1f}{(r[6] or 0) / 1e6:>12.3f}")


if __name__ == "__main__":
    args = sys.argv[1:]
    pmc = args and args[0] == "--pmc"
    if pmc:
        args = args[1:]
    con = sqlite3.connect(args[0])
    (pmc_stats if pmc else kernel_stats)(con, args[1:])
