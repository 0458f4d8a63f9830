"""Per-kernel PMC summary from a rocprofv3 --pmc run (rocpd sqlite): for every (kernel, grid) the max /
average counter value per dispatch, in MiB when the counter is FETCH_SIZE / WRITE_SIZE (reported in KiB).

usage: python tools/rocpd_pmc.py <results.db> [min_MiB]"""
import sqlite3
import sys


def main(path, min_mib=50.0):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select counter_name, kernel_name, grid_size, count(*), avg(value), max(value), avg(duration) from counters_collection "
        "group by counter_name, kernel_name, grid_size order by max(value) desc"
    ).fetchall()
    print(f"# {path}")
    print(f"{'counter':>11} {'calls':>6} {'avg_MiB':>10} {'max_MiB':>10} {'avg_us':>9}  kernel grid")
    for name, kern, grid, calls, avg, mx, dur in rows:
        if mx / 1024.0 < min_mib:
            continue
        k = kern if len(kern) < 110 else kern[:110] + "..."
        print(f"{name:>11} {calls:6d} {avg/1024.0:10.1f} {mx/1024.0:10.1f} {dur/1e3:9.1f}  {k} grid={grid}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 50.0)
