"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: python tools/rocpd_summary.py <results.db> [header line ...]
Prints per-kernel calls / total / average / min / max, sorted by total time (the --stats view, as text)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    dur = "duration" if "duration" in cols else '("end" - start)'
    rows = db.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    for h in sys.argv[2:]:
        print("# " + h)
    print(f"# total kernel time {total / 1e6:.3f} ms")
    print(f"{'kernel':<100}{'calls':>8}{'total_ms':>15}{'pct':>9}{'avg_us':>15}{'min_us':>15}{'max_us':>15}")
    for r in rows:
        print(f"{r[0][:96]:<100}{r[1]:>8}{r[2] / 1e6:>15.3f}{100.0 * r[2] / total:>8.2f}%{r[3] / 1e3:>15.2f}{r[4] / 1e3:>15.2f}{r[5] / 1e3:>15.2f}")


if __name__ == "__main__":
    main()
