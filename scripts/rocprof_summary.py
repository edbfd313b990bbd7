#!/usr/bin/env python
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary kept
under profiles/.  Usage: rocprof_summary.py results.db [top_n] > profiles/xx.txt"""
import sqlite3
import sys


def demangle(name):
    """_ZN2dg12_GLOBAL__N_1<len><kernel>I...E... -> dg::<kernel><template args as mangled>"""
    import re
    m = re.match(r"_ZN2dg12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    start = m.end()
    return "dg::" + name[start:start + n] + " " + name[start + n:]


def main():
    db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats   total kernel time {total / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, dur, avg, pct in rows[:top]:
        print(f"{calls:7d} {dur / 1e3:10.3f} {avg:10.2f} {pct:6.2f}  {demangle(name)[:150]}")
    groups = {"hipBLASLt/rocBLAS GEMM (Cijk_*)": 0.0, "druggen_amd HIP kernels (dg::)": 0.0, "ATen elementwise/reduce": 0.0, "other": 0.0}
    for name, calls, dur, avg, pct in rows:
        if name.startswith("Cijk_"):
            groups["hipBLASLt/rocBLAS GEMM (Cijk_*)"] += dur
        elif "dg::" in name or name.startswith("_ZN2dg"):      # demangled or mangled (templated kernels keep the mangled name)
            groups["druggen_amd HIP kernels (dg::)"] += dur
        elif "at::native" in name:
            groups["ATen elementwise/reduce"] += dur
        else:
            groups["other"] += dur
    print("\n# by group")
    for k, v in groups.items():
        print(f"{v / 1e3:10.3f} ms {100 * v / total:6.2f} %  {k}")


if __name__ == "__main__":
    main()
