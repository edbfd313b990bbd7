#!/usr/bin/env python
"""Kernels dispatched right before / after every __amd_rocclr_copyBuffer in a rocprofv3 kernel trace (developer tool)."""
import collections, sqlite3, sys
from rocprof_summary import demangle
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
names = [demangle(r[0])[:70] for r in rows]
c = collections.Counter()
for i, n in enumerate(names):
    if "copyBuffer" in n:
        c[(names[i - 1] if i else "", names[i + 1] if i + 1 < len(names) else "")] += 1
for (a, b), n in c.most_common(20):
    print(n, "|", a, "|", b)
