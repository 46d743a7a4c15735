#!/usr/bin/env python3
"""usage: tools/isa_census.py file.s [kernel-substring] [min-block-size]
Opcode census of a `hipcc -S --cuda-device-only` listing: per kernel, every basic block (label to label) with at
least min-block-size instructions, its VALU / LDS / VMEM / SALU counts and the VALU opcodes by frequency.  Used to see
what a segment's steady-state loop spends its issue slots on (DESIGN 4.1 "instruction census")."""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 150
kernel, blocks, cur, name = None, [], None, None
for ln in open(path):
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        kernel = m.group(1)
        cur = ['entry', []]
        blocks.append((kernel, cur))
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m and kernel:
        cur = [m.group(1), []]
        blocks.append((kernel, cur))
        continue
    m = re.match(r'^\t([a-z_0-9]+)', ln)
    if m and cur is not None and not ln.startswith('\t.'):
        cur[1].append(m.group(1))
        if m.group(1) == 's_endpgm':
            pass


def cls(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem'
    if op.startswith('s_waitcnt') or op.startswith('s_barrier') or op.startswith('s_setprio') or op.startswith('s_nop'):
        return 'sync'
    return 'salu'


for kernel, (label, ops) in blocks:
    if want not in kernel or len(ops) < min_block:
        continue
    c = collections.Counter(cls(o) for o in ops)
    v = collections.Counter(o for o in ops if o.startswith('v_'))
    print('%s %s: %d instr  valu %d lds %d vmem %d salu %d sync %d' % (kernel[-40:], label, len(ops), c['valu'], c['lds'], c['vmem'], c['salu'], c['sync']))
    print('    ' + '  '.join('%s %d' % kv for kv in v.most_common(24)))
