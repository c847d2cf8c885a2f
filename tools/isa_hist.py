#!/usr/bin/env python3
"""Opcode histogram of a kernel's whole body (or its largest loop with --loop) in a gfx950 .s file (diagnostic).
usage: isa_hist.py <file.s> <mangled-name-substring> [--loop]"""
import re, sys, collections
src = open(sys.argv[1]).read()
names = [n for n in re.findall(r'^(_Z\S+):', src, re.M) if sys.argv[2] in n]
for tgt in names:
    s = src.index('\n' + tgt + ':'); e = src.index('.Lfunc_end', s)
    body = src[s:e].split('\n')
    a, b = 0, len(body)
    if '--loop' in sys.argv:
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
        op = l.split()[0]
        if 'row_' in l or 'quad_perm' in l: op += '(dpp)'
        cnt[op] += 1
    print(tgt, sum(cnt.values()))
    print('  ' + '  '.join(f'{n} {op}' for op, n in cnt.most_common(40)))
