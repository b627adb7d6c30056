#!/usr/bin/env python3
"""Instruction mix of the steady-state march loop of the pipelined hand-over kernel, from the device assembly
(hipcc --cuda-device-only -S). With one wave per SIMD every instruction of the wave costs an issue slot, scalar
ones included, so the counts are the thing to watch.  usage: hot_loop_mix.py file.s [mode]"""
import collections, re, sys
L = open(sys.argv[1]).read().split('\n')
modes = [int(sys.argv[2])] if len(sys.argv) > 2 else [0, 1]
for mode in modes:
    start = [i for i, l in enumerate(L) if re.match(r'^_Z10k_fused_hoILi4ELi%dE.*:' % mode, l)][0]
    end = [i for i in range(start, len(L)) if 's_endpgm' in L[i]][0]
    body = L[start:end]
    lab = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: lab[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in lab and lab[m.group(1)] < i: loops.append((lab[m.group(1)], i))
    lp = max(loops, key=lambda x: x[1] - x[0])                            # the steady-state march loop is by far the longest
    cnt = collections.Counter()
    for l in body[lp[0]:lp[1] + 1]:
        t = l.strip().split(';')[0].split()
        if not t or t[0].endswith(':') or t[0].startswith('.'): continue
        op = t[0]
        if op.startswith('v_'):
            if '_f64' in op: cnt['valu_f64'] += 1
            elif '_f32' in op: cnt['valu_f32'] += 1
            elif op.startswith('v_mov'): cnt['v_mov'] += 1
            elif op.startswith('v_accvgpr'): cnt['accvgpr'] += 1
            elif 'lane' in op: cnt['lane'] += 1
            else: cnt['valu_int'] += 1
        elif op.startswith('ds_'): cnt['lds'] += 1
        elif op.startswith('global_'): cnt['vmem'] += 1
        elif op.startswith('s_waitcnt'): cnt['waitcnt'] += 1
        elif op.startswith('s_nop'): cnt['s_nop'] += 1
        elif op.startswith('s_'): cnt['salu'] += 1
        else: cnt[op] += 1
    print('mode', mode, 'instructions', sum(cnt.values()), dict(sorted(cnt.items(), key=lambda x: -x[1])))
