"""Which kernels of OTHER queues run while a given kernel runs?  (rocprofv3 --kernel-trace, rocpd sqlite)
usage: python tools/rocpd_overlap.py <results.db> <substring of the kernel name>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = db.execute(f"select name, start, end, {qcol}, grid_x from kernels order by start").fetchall()
want = [r for r in rows if sys.argv[2] in r[0]]
print(f"# {len(want)} launches of *{sys.argv[2]}*")
by_grid = collections.defaultdict(lambda: collections.Counter())
for w in want:
    for r in rows:
        if r[3] != w[3] and r[1] < w[2] and r[2] > w[1]:
            nm = r[0]
            i = nm.find("::")
            by_grid[w[4]][(nm[i + 2:i + 62] if i >= 0 else nm[:60], r[4])] += 1
for grid, c in sorted(by_grid.items()):
    n = sum(1 for w in want if w[4] == grid)
    print(f"grid_x {grid}: {n} launches; overlapping kernels of other queues (name, grid): ")
    for (nm, g), k in c.most_common(12):
        print(f"    {k:4d} x {nm}  grid_x={g}")
