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
    # the persistent kernels run one workgroup per CU at edge level (grid = 256 workgroups) and fewer at node level:
    # the per-kernel averages above mix the two; bench.py's roofline blocks time the edge-level launches only
    # (R >= DG_EDGE_ROWS).  A node-level launch over 2B molecules (23 040 rows = 360 tiles) also fills the grid, with one
    # or two tiles per workgroup against >= 31: full-grid launches of a kernel are therefore split once more at the
    # largest gap of their sorted durations (when it is at least 3x).
    try:
        disp = {}
        for name, full, dur in cur.execute(
                "select name, (grid_x >= 256 * workgroup_x), duration from kernels where name like '%row_gemm%' "
                "or name like '%wgrad_kernel%' or name like '%wgrad_stream%' or name like '%ffn_%bf16%' or name like '%ffn_fused%' or name like '%attn_half%'"):
            disp.setdefault(name, {0: [], 1: []})[1 if full else 0].append(dur)
        table = []
        for name, d in disp.items():
            if d[0]:
                table.append((sum(d[0]), len(d[0]), "small", name))
            fl = sorted(d[1])
            cut = 0
            if len(fl) > 1:
                ratio, at = max((fl[i + 1] / max(fl[i], 1), i + 1) for i in range(len(fl) - 1))
                cut = at if ratio >= 3 else 0
            if cut:
                table.append((sum(fl[:cut]), cut, "full/short", name))
            if fl:
                table.append((sum(fl[cut:]), len(fl) - cut, "full/edge" if cut else "full", name))
        table.sort(reverse=True)
        print("\n# persistent kernels by launch size (full = at least 256 workgroups; full/edge = the edge-level launches)")
        print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10}  {'launches':>10}  kernel")
        for dur, calls, cls, name in table[:30]:
            print(f"{calls:7d} {dur / 1e6:10.3f} {dur / calls / 1e3:10.2f}  {cls:>10}  {demangle(name)[:120]}")
    except sqlite3.Error as e:      # older rocpd schemas
        print(f"\n# (no per-dispatch view in this trace: {e})")
    print("\n# by group")
    for k, v in groups.items():
        print(f"{v / 1e3:10.3f} ms {100 * v / total:6.2f} %  {k}")


if __name__ == "__main__":
    main()
