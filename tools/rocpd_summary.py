"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database instead of CSVs: this turns its kernel-dispatch table into the per-kernel summary that
`--stats` used to print.  usage: python tools/rocpd_summary.py <results.db> <steps profiled> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db, steps = sqlite3.connect(sys.argv[1]), float(sys.argv[2])
    rows = db.execute("select name, count(*), sum(end - start) / 1e3, avg(end - start) / 1e3, min(end - start) / 1e3, max(end - start) / 1e3 "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    out = csv.writer(open(sys.argv[3], "w", newline="") if len(sys.argv) > 3 else sys.stdout)
    out.writerow(["kernel", "calls_per_step", "total_us_per_step", "avg_us", "min_us", "max_us", "percent"])
    for name, n, tot, avg, lo, hi in rows:
        out.writerow([name, round(n / steps, 2), round(tot / steps, 1), round(avg, 2), round(lo, 2), round(hi, 2), round(100 * tot / total, 2)])
    out.writerow(["TOTAL", round(sum(r[1] for r in rows) / steps, 1), round(total / steps, 1), "", "", "", 100.0])


if __name__ == "__main__":
    main()
