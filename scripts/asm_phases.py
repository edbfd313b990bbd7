"""Per-phase (between s_barrier) instruction mix of one kernel: python scripts/asm_phases.py file.s name-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    body = [l.strip() for l in m.group(2).split('\n')]
    body = [l for l in body if l and not l.startswith((';', '.p2align'))]
    seg, cur = [], []
    for l in body:
        cur.append(l)
        if l.startswith('s_barrier'):
            seg.append(cur); cur = []
    seg.append(cur)
    for i, sg in enumerate(seg):
        c = Counter()
        for l in sg:
            op = l.split()[0]
            if op.endswith(':'): continue
            if op.startswith('v_mfma'): c['mfma'] += 1
            elif op.startswith('ds_'): c['lds'] += 1
            elif op.startswith(('global_', 'buffer_', 'scratch_')): c['vmem'] += 1
            elif op.startswith(('v_exp', 'v_rcp', 'v_rsq')): c['trans'] += 1
            elif op.startswith('v_pk_'): c['vpk'] += 1
            elif op.startswith('v_'): c['valu'] += 1
            elif op.startswith('s_'): c['salu'] += 1
        print(i, len(sg), dict(c))
        if len(sys.argv) > 3 and int(sys.argv[3]) == i:
            cc = Counter(l.split()[0] for l in sg if l.split()[0].startswith('v_'))
            print(cc.most_common(50))
