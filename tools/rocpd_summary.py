"""Summarise a rocprofv3 (rocpd sqlite) run.
  python tools/rocpd_summary.py <results.db> [header line ...]          per-kernel calls / total / avg / min / max (--stats as text)
  python tools/rocpd_summary.py --pmc <results.db> [header line ...]    per-kernel, per-counter: dispatches, sum, per-dispatch average
  python tools/rocpd_summary.py --gaps <results.db> [header line ...]   the phases before / between / after the two CM launches of a bench step:
                                                                        wall time, time with no kernel at all, time with only the one-workgroup-per-block
                                                                        kernels (k_lzp_driver / k_lzp_decode) running, and the longest such stretches
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


def gap_stats(db, headers):
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f'select {name}, start, "end" from kernels order by start').fetchall()
    for h in headers:
        print("# " + h)
    cm = [(s, e, n) for n, s, e in rows if n.startswith("bz3::k_cm_") and e - s > 1e9]
    if len(cm) < 2:
        print("fewer than two long CM launches in the trace")
        return
    enc, dec = cm[-2], cm[-1]
    first = min(s for n, s, e in rows if n.startswith("bz3::") and s > enc[0] - 400e9)
    last = max(e for n, s, e in rows)
    serial = ("bz3::k_lzp_driver", "bz3::k_lzp_decode")
    for label, t0, t1 in (("front end (before the CM encode launch)", first, enc[0]), ("between the CM launches", enc[1], dec[0]), ("tail (after the CM decode launch)", dec[1], last)):
        ev = []
        for n, s, e in rows:
            if e <= t0 or s >= t1:
                continue
            k = 1 if n.startswith(serial) else 0
            ev.append((max(s, t0), 1, k))
            ev.append((min(e, t1), -1, k))
        ev.sort()
        live = [0, 0]
        t = t0
        none = only_serial = 0
        stretches = []
        cur0 = None
        for x, d, k in ev:
            if live[0] == 0:
                if live[1] == 0:
                    none += x - t
                else:
                    only_serial += x - t
                if cur0 is None:
                    cur0 = t
            if live[0] == 0 and d == 1 and k == 0 and cur0 is not None:
                stretches.append((x - cur0, cur0 - t0))
                cur0 = None
            live[k] += d
            t = x
        none += t1 - t
        stretches.sort(reverse=True)
        print(f"{label}: wall {(t1 - t0) / 1e9:.2f} s, no kernel running {none / 1e9:.2f} s, only k_lzp_driver / k_lzp_decode running {only_serial / 1e9:.2f} s")
        print("   longest stretches without a whole-GPU kernel (s, at s into the phase):", [(round(a / 1e9, 3), round(b / 1e9, 2)) for a, b in stretches[:8]])


if __name__ == "__main__":
    args = sys.argv[1:]
    mode = args[0] if args and args[0] in ("--pmc", "--gaps") else ""
    if mode:
        args = args[1:]
    con = sqlite3.connect(args[0])
    {"--pmc": pmc_stats, "--gaps": gap_stats, "": kernel_stats}[mode](con, args[1:])
