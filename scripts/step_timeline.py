#!/usr/bin/env python
"""Small-launch accounting from a rocprofv3 kernel trace (rocpd sqlite): per kernel name, the dispatches shorter than a
threshold (count, total, average per step), and the idle time between consecutive dispatches.
   step_timeline.py results.db <steps> [threshold_us=40]"""
import sqlite3
import sys
from collections import defaultdict

from rocprof_summary import demangle


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = list(cur.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}"))
    small = defaultdict(lambda: [0, 0.0])
    big_t = small_t = idle = 0.0
    last_end = None
    for name, s, e in rows:
        d = (e - s) / 1e3
        if d < thr:
            small[name][0] += 1
            small[name][1] += d
            small_t += d
        else:
            big_t += d
        if last_end is not None and s > last_end:
            idle += (s - last_end) / 1e3
        last_end = e if last_end is None else max(last_end, e)
    n = len(rows)
    print(f"# {n} dispatches over {steps} steps ({n / steps:.0f} per step); per step: launches >= {thr:.0f} us "
          f"{big_t / steps / 1e3:.2f} ms, shorter ones {small_t / steps / 1e3:.2f} ms in {sum(v[0] for v in small.values()) / steps:.0f} "
          f"dispatches, idle between dispatches {idle / steps / 1e3:.2f} ms")
    print(f"{'per step':>9} {'ms/step':>8} {'avg_us':>7}  kernel (dispatches under {thr:.0f} us)")
    for name, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{c / steps:9.1f} {t / steps / 1e3:8.3f} {t / c:7.2f}  {demangle(name)[:130]}")


if __name__ == "__main__":
    main()
