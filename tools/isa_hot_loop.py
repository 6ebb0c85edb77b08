#!/usr/bin/env python3
"""Per-basic-block ISA accounting of a kernel's innermost sample loops.

    python tools/isa_hot_loop.py <unit>.hip <mangled-kernel-regex> [loop-header-label]

Compiles klang_amd/csrc/<unit>.hip to gfx950 assembly (device only), then for every depth-3
loop of the kernel (the eight-sample group loops of voice_kernel) lists its basic blocks in
layout order: instructions, VALU (float64 among them), SALU, s_waitcnt, LDS, vector memory and
where the block branches.  With a loop header label, prints that loop's blocks in full.
The straight-line path of a loop (speculation holds) is the header block plus the blocks it
falls / branches through without a redo; the annotated copy kept under profiles/ marks it."""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, pat = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else None
src = os.path.join(ROOT, 'klang_amd', 'csrc', unit)
out = '/tmp/%s.s' % os.path.basename(unit)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
                       '-ffp-contract=off', '-Wno-unused-function', '-S', '--cuda-device-only',
                       src, '-o', out], stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r'^(%s\w*):' % pat, s, re.M)
name = m.group(1)
body = s[m.start():s.index('.Lfunc_end', m.start())].split('\n')
blocks, order, loop_of = {}, [], {}
cur = None
for ln in body:
    mm = re.match(r'^(\.LBB\d+_\d+):(.*)', ln)
    if mm:
        cur = mm.group(1)
        blocks[cur] = []
        order.append(cur)
        h = re.search(r'Header=(BB\d+_\d+) Depth=3', mm.group(2))
        loop_of[cur] = '.L' + h.group(1) if h else None
    elif cur:
        t = ln.strip()
        if 'This Inner Loop Header: Depth=3' in t:
            loop_of[cur] = cur
        if ln.startswith('\t') and not t.startswith(('.', ';')):
            blocks[cur].append(t)


def mix(ins):
    c = Counter(i.split()[0] for i in ins)
    return dict(n=len(ins), valu=sum(v for k, v in c.items() if k.startswith('v_')),
                f64=sum(v for k, v in c.items() if 'f64' in k),
                salu=sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_waitcnt', 's_nop')),
                wait=c.get('s_waitcnt', 0), nop=c.get('s_nop', 0),
                lds=sum(v for k, v in c.items() if k.startswith('ds_')),
                vmem=sum(v for k, v in c.items() if k.startswith(('global_', 'scratch_', 'flat_', 'buffer_'))),
                to=[i.split()[-1] for i in ins if i.startswith(('s_cbranch', 's_branch'))])


print(name)
loops = [b for b in order if loop_of.get(b) == b]
for head in loops:
    if want and head != want:
        continue
    members = [b for b in order if loop_of.get(b) == head]
    tot = Counter()
    print('loop %s: %d blocks' % (head, len(members)))
    for b in members:
        x = mix(blocks[b])
        if x['n'] >= 12 or b == head:
            print('  %-12s n %3d valu %3d (f64 %3d) salu %2d wait %d lds %d vmem %d -> %s'
                  % (b, x['n'], x['valu'], x['f64'], x['salu'], x['wait'], x['lds'], x['vmem'],
                     ' '.join(x['to'])))
    if want:
        for b in members:
            print(b + ':')
            for i in blocks[b]:
                print('    ' + i)
