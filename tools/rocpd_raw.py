"""Raw per-kernel counter averages from a rocprofv3 --pmc run.  usage: python tools/rocpd_raw.py <db> [kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else ""
for name, kern, calls, avg, dur in db.execute("select counter_name, kernel_name, count(*), avg(value), avg(duration) from counters_collection group by counter_name, kernel_name order by kernel_name, counter_name"):
    if sub in kern:
        print(f"{name:34s} {avg:16.1f}  calls={calls} avg_us={dur/1e3:8.1f}  {kern[:60]}")
