"""The kernels of ONE training step of a rocprofv3 kernel trace (rocpd sqlite) in start order: offset from the step's first
kernel, duration, gap to the previous kernel's end on the same queue, queue, grid, short name - what one reads a schedule
from (which launches form a dependent chain, where the queue idles).

usage: python tools/rocpd_sequence.py <results.db> [step_from_end=1]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main(path, back=1):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else "stream_id"
    rows = db.execute(f"select name, start, end, {qcol}, grid_x, grid_y from kernels order by start").fetchall()
    opt = [i for i, r in enumerate(rows) if "FusedOptimizerTensorListMetadata" in r[0]]
    groups, cur = [], [opt[0]]
    for i in opt[1:]:
        if rows[i][1] - rows[cur[-1]][2] > 2_000_000:
            groups.append(cur)
            cur = [i]
        else:
            cur.append(i)
    groups.append(cur)
    lo, hi = groups[-back - 1][-1] + 1, groups[-back][-1]
    step = rows[lo:hi + 1]
    t0 = step[0][1]
    queues = {}
    last_end = {}
    print(f"# {len(step)} kernels, wall {(max(r[2] for r in step) - t0) / 1e3:.1f} us")
    print(f"{'t_us':>9} {'dur_us':>8} {'gap_us':>7} q {'grid':>12}  kernel")
    for name, s, e, q, gx, gy in step:
        qi = queues.setdefault(q, len(queues))
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {qi} {gx:>9}x{gy:<3} {short(name)}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
