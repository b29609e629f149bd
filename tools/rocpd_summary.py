"""Text summary of a rocprofv3 result database (rocpd SQLite): per-kernel time
statistics and, when counters were collected, per-kernel counter sums.
Usage: python tools/rocpd_summary.py results.db [more.db ...]"""
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("snappy::(anonymous namespace)::", "")
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name[:70]


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print(f"== {path}")
        try:
            rows = cur.execute(
                "select name, count(*), sum(end - start), avg(end - start), min(end - start), "
                "max(end - start) from kernels group by name order by 3 desc").fetchall()
        except sqlite3.Error:
            rows = []
        if rows:
            total = sum(r[2] for r in rows) or 1
            print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'min_us':>12s} "
                  f"{'max_us':>12s} {'pct':>6s}")
            for name, calls, tot, avg, mn, mx in rows[:12]:
                print(f"{short(name):70s} {calls:6d} {tot / 1e3:12.1f} {avg / 1e3:12.1f} "
                      f"{mn / 1e3:12.1f} {mx / 1e3:12.1f} {100.0 * tot / total:6.2f}")
        try:
            rows = cur.execute(
                "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                "from counters_collection group by kernel_name, counter_name "
                "order by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            rows = []
        if rows:
            print(f"{'kernel':50s} {'counter':24s} {'sum':>20s} {'dispatches':>10s} {'per_dispatch':>18s}")
            for k, c, v, n in rows:
                if "snappy" not in k:
                    continue
                print(f"{short(k):50s} {c:24s} {v:20.0f} {n:10d} {v / max(n, 1):18.1f}")
        con.close()


if __name__ == "__main__":
    main()
