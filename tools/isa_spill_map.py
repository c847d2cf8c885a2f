"""usage: python3 tools/isa_spill_map.py <file.s> <kernel-name-substring>
Where a kernel's register spills sit: the order of scratch stores (S), scratch loads (L), global loads (G), LDS barriers (B)
and block labels over the kernel's instruction stream, from the ISA `tools/kres_ntt.sh` / `tools/isa.sh` leave in /tmp."""
import re, sys
L = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(L) if l.startswith('_ZN') and name in l and l.rstrip().endswith(':') or (name in l and '; @' in l and l.startswith('_ZN'))][0]
end = next(i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end'))
body = L[start:end]
out, nvalu = '', 0
for i, l in enumerate(body):
    m = l.strip().split(' ')[0] if l.strip() else ''
    if m.startswith('scratch_store'): out += 'S'
    elif m.startswith('scratch_load'): out += 'L'
    elif m == 's_barrier': out += 'B'
    elif m.startswith('global_load'): out += 'G'
    elif m.startswith('v_'): nvalu += 1
    elif re.match(r'\.LBB\d+_\d+:', l): out += '\n%s @%d valu=%d\n' % (l.strip(), i, nvalu)
print(len(body), 'lines,', nvalu, 'VALU')
print(out)
