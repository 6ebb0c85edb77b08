"""List the big basic blocks of one kernel in klang_mi.gfx950.s with instruction mixes.
Usage: python tools/asm_blocks.py <mangled-name-regex> [min_instructions]"""
import re
import sys
from collections import Counter

s = open('klang_amd/csrc/klang_mi.gfx950.s').read()
pat = sys.argv[1]
min_ins = int(sys.argv[2]) if len(sys.argv) > 2 else 150
m = re.search(r'^(%s\w*):' % pat, s, re.M)
name = m.group(1)
start = m.start()
end = s.index('.Lfunc_end', start)
body = s[start:end].split('\n')
print(name, len(body))
blocks = []
cur = None
for ln in body:
    if re.match(r'^\.LBB\d+_\d+:', ln):
        cur = [ln.split(':')[0], []]
        blocks.append(cur)
    elif cur and ln.startswith('\t') and not ln.strip().startswith(('.', ';')):
        cur[1].append(ln.strip())
for nm, ins in blocks:
    if len(ins) > min_ins:
        c = Counter(i.split()[0] for i in ins)
        v = sum(n for k, n in c.items() if k.startswith('v_'))
        f64 = sum(n for k, n in c.items() if 'f64' in k)
        print(nm, len(ins), 'valu', v, 'f64', f64,
              'salu', sum(n for k, n in c.items() if k.startswith('s_')),
              'ds', sum(n for k, n in c.items() if k.startswith('ds_')))
        if '-v' in sys.argv:
            for k, n in c.most_common(40):
                print('    ', k, n)
