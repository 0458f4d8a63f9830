"""Timeline reading of a rocprofv3 kernel trace (rocpd sqlite) of bench.py: for the LAST complete training step
(delimited by the fused-AdamW kernels) print wall time, GPU-busy / idle time, how much of it ran >= 2 kernels at once,
per-queue busy time, the split of kernel time into big (>= 150 us) and small launches and the largest idle gaps.

usage: python tools/rocpd_timeline.py <results.db> [step_from_end=1]"""
import sqlite3
import sys


def main(path, back=1):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    print("# columns:", cols)
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"name, start, end, {qcol or '0'}, grid_x" if "grid_x" in cols else f"name, start, end, {qcol or '0'}, 0"
    rows = db.execute(f"select {sel} from kernels order by start").fetchall()
    opt = [i for i, r in enumerate(rows) if "FusedOptimizerTensorListMetadata" in r[0]]
    # AdamW launches come in groups (one per tensor-list chunk): group boundaries = gaps of > 2 ms between them
    groups, cur = [], [opt[0]]
    for i in opt[1:]:
        if rows[i][1] - rows[cur[-1]][2] > 2_000_000:
            groups.append(cur)
            cur = [i]
        else:
            cur.append(i)
    groups.append(cur)
    if len(groups) < back + 1:
        print("not enough steps in the trace")
        return
    lo = groups[-back - 1][-1] + 1
    hi = groups[-back][-1]
    step = rows[lo:hi + 1]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print(f"# step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms")
    # sweep
    ev = []
    for r in step:
        ev.append((r[1], 1))
        ev.append((r[2], -1))
    ev.sort()
    busy = multi = 0
    depth, last = 0, t0
    gaps = []
    gap_start = None
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        if depth == 0 and t > last:
            gaps.append((t - last, last))
        depth += d
        last = t
    print(f"# GPU busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, >=2 kernels at once {multi / 1e6:.3f} ms")
    tot = sum(r[2] - r[1] for r in step)
    big = [r for r in step if r[2] - r[1] >= 150_000]
    print(f"# kernel time sum {tot / 1e6:.3f} ms: {len(big)} launches >= 150 us = {sum(r[2] - r[1] for r in big) / 1e6:.3f} ms, "
          f"{len(step) - len(big)} smaller = {(tot - sum(r[2] - r[1] for r in big)) / 1e6:.3f} ms")
    per_q = {}
    for r in step:
        per_q.setdefault(r[3], [0, 0])
        per_q[r[3]][0] += r[2] - r[1]
        per_q[r[3]][1] += 1
    for q, (t, n) in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
        print(f"#   queue {q}: {n} kernels, {t / 1e6:.3f} ms")
    gaps.sort(reverse=True)
    print("# largest idle gaps (us, at ms into the step, kernel that ended before / started after):")
    for g, at in gaps[:12]:
        before = max((r for r in step if r[2] <= at + 1), key=lambda r: r[2], default=None)
        after = min((r for r in step if r[1] >= at + g - 1), key=lambda r: r[1], default=None)
        print(f"   {g / 1e3:8.1f} us @ {(at - t0) / 1e6:7.3f} ms   {before[0][:60] if before else None} -> {after[0][:60] if after else None}")
    print(f"# idle in gaps < 20 us: {sum(g for g, _ in gaps if g < 20_000) / 1e6:.3f} ms over {sum(1 for g, _ in gaps if g < 20_000)} gaps")
    # big kernels table (duration inflation under sharing)
    agg = {}
    for r in big:
        k = (r[0][:70], r[4])
        a = agg.setdefault(k, [0, 0])
        a[0] += r[2] - r[1]
        a[1] += 1
    print("# big kernels in the step:")
    for (name, grid), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"   {n:3d} x {t / n / 1e3:8.1f} us = {t / 1e6:7.3f} ms  {name} grid_x={grid}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
