"""Instruction mix of the kernels in a hipcc -S listing: python scripts/asm_mix.py file.s [name-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    c = Counter()
    loop_lines = body.split('\n')
    for l in loop_lines:
        l = l.strip()
        if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('ds_'): c[op] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_')): c[op] += 1
        elif op.startswith('s_waitcnt'): c['s_waitcnt'] += 1
        elif op.startswith('s_barrier'): c['s_barrier'] += 1
        elif op.startswith('v_exp') or op.startswith('v_rcp') or op.startswith('v_rsq'): c['trans'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    print(name, dict(sorted(c.items())))
