#!/usr/bin/env python3
"""Instruction mix of the main loop of a kernel in a gfx950 .s file (diagnostic).
usage: isa_loop_count.py <file.s> <mangled-name-substring>"""
import re, sys, collections
src = open(sys.argv[1]).read()
names = [n for n in re.findall(r'^(_Z\S+):', src, re.M) if sys.argv[2] in n]
for tgt in names:
    s = src.index('\n' + tgt + ':'); e = src.index('.Lfunc_end', s)
    body = src[s:e].split('\n')
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\S+):', l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)', l)
        if m:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab] < i: loops.append((labels[lab], i))
    a, b = max(loops, key=lambda x: x[1] - x[0])
    cnt = collections.Counter()
    for l in body[a:b]:
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'): continue
        cnt[l.split()[0]] += 1
    def cls(op):
        if op in ('v_add_f64', 'v_mul_f64'): return 'f64 arith'
        if op.startswith('v_') and 'f64' in op: return 'f64 other'
        if op.startswith('v_'): return 'valu int'
        if op.startswith('ds_'): return 'lds'
        if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem/scratch'
        if op == 's_barrier': return 'barrier'
        return 'salu'
    cc = collections.Counter()
    for op, n in cnt.items(): cc[cls(op)] += n
    vg = re.search(r'\.vgpr_count:\s+(\d+)', src[src.index(tgt, e):]) if tgt in src[e:] else None
    print(tgt[:90], 'loop instrs', sum(cnt.values()), dict(cc), 'scratch' if any(o.startswith('scratch_') for o in cnt) else '')
