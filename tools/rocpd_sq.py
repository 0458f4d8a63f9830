"""Per-kernel SQ counter summary of a rocprofv3 --pmc run (rocpd sqlite): per (kernel, grid) the averages per dispatch and the
derived shares north_star asks for - matrix-pipe busy % against the chip's 1 024 SIMDs, waves parked on s_waitcnt / issue stalls.

usage: python tools/rocpd_sq.py <results.db> [min_avg_us]
  mfma_busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)      (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
  wait_any % = SQ_WAIT_ANY / SQ_WAVE_CYCLES,  wait_inst % = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,  lds_conf % = SQ_LDS_BANK_CONFLICT / SQ_WAVE_CYCLES"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(?:::)?(\w+(?:<[^>]*>)?)\(", name)
    return (m.group(1) if m else name)[:64]


def main(path, min_us=100.0):
    db = sqlite3.connect(path)
    acc = defaultdict(dict)
    for cname, kern, grid, calls, avg, dur in db.execute(
            "select counter_name, kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
            "group by counter_name, kernel_name, grid_size"):
        acc[(kern, grid)][cname] = avg
        acc[(kern, grid)]["_calls"], acc[(kern, grid)]["_us"] = calls, dur / 1e3
    names = sorted({c for v in acc.values() for c in v if not c.startswith("_")})
    print(f"# {path}: counters {' '.join(names)}")
    print(f"{'kernel':64s} {'grid':>9} {'calls':>5} {'avg_us':>8} {'mfma_busy%':>10} {'wait_any%':>9} {'wait_inst%':>10} {'lds_conf%':>9} "
          f"{'valu_inst_M':>11} {'mfma_inst_M':>11}")
    for (kern, grid), v in sorted(acc.items(), key=lambda kv: -kv[1]["_us"] * kv[1]["_calls"]):
        if v["_us"] < min_us and "angle" not in kern:
            continue
        gui = v.get("GRBM_GUI_ACTIVE")
        wc = v.get("SQ_WAVE_CYCLES")

        def pct(num, den):
            return f"{100.0 * num / den:.1f}" if (num is not None and den) else "-"

        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
        print(f"{short(kern):64s} {grid:9d} {v['_calls']:5d} {v['_us']:8.1f} {pct(mf, 1024.0 * gui / 8.0 if gui else None):>10} "
              f"{pct(v.get('SQ_WAIT_ANY'), wc):>9} {pct(v.get('SQ_WAIT_INST_ANY'), wc):>10} {pct(v.get('SQ_LDS_BANK_CONFLICT'), wc):>9} "
              f"{(v.get('SQ_INSTS_VALU', float('nan')) / 1e6):11.2f} {(v.get('SQ_INSTS_MFMA', float('nan')) / 1e6):11.2f}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0)
