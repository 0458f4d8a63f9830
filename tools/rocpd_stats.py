"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / avg / %.

usage: python tools/rocpd_stats.py <results.db> [--grid]  > profiles/<name>.stats.txt
(rocprofv3's --stats view, regenerated from the database so it can be committed as text)."""

import sqlite3
import sys


def main(path, by_grid=False):
    db = sqlite3.connect(path)
    key = "name, grid_x, grid_y, grid_z" if by_grid else "name"
    rows = db.execute(
        f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by {key} order by sum(duration) desc"
    ).fetchall()
    total = sum(r[-4] for r in rows)
    print(f"# {path}: {sum(r[-5] for r in rows)} dispatches, {total/1e6:.3f} ms total kernel time")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for r in rows:
        name = r[0]
        if len(name) > 150:
            name = name[:150] + "..."
        extra = f" grid=({r[1]},{r[2]},{r[3]})" if by_grid else ""
        calls, tot, avg, mn, mx = r[-5:]
        print(f"{calls:7d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f}  {name}{extra}")


if __name__ == "__main__":
    main(sys.argv[1], "--grid" in sys.argv)
