#!/usr/bin/env python3
"""Compressed instruction timeline of one kernel in a hipcc -S listing: runs of global loads/stores,
vmcnt waits, barriers, scratch (spill) traffic and Philox multiplies.  usage: isa_timeline.py file.s <symbol prefix>"""
import sys
s = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
i0 = [i for i, l in enumerate(s) if l.startswith(pref) and '@' in l][0]
fend = next(i for i in range(i0, len(s)) if s[i].startswith('.Lfunc_end'))
ev = []
for k, l in enumerate(s[i0:fend]):
    t = l.strip()
    if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
        continue
    op = t.split()[0]
    if op.startswith('global_load'): c = 'GL'
    elif op.startswith('global_store'): c = 'GS'
    elif op in ('v_mul_hi_u32', 'v_mul_lo_u32'): c = 'PHILOX'
    elif op == 's_waitcnt' and 'vmcnt' in t: c = 'WAITVM(' + t.split('vmcnt(')[1].split(')')[0] + ')'
    elif op == 's_barrier': c = 'BARRIER'
    elif op.startswith('scratch_'): c = 'SCR'
    elif op.startswith('s_cbranch'): c = 'BR'
    else: continue
    ev.append((k, c))
out, last, cnt, start = [], None, 0, 0
for k, c in ev:
    if c == last: cnt += 1
    else:
        if last: out.append(f"{start}:{last}x{cnt}")
        last, cnt, start = c, 1, k
out.append(f"{start}:{last}x{cnt}")
print(fend - i0, ' '.join(out))
