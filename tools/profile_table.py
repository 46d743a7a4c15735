#!/usr/bin/env python3
"""usage: tools/profile_table.py <round>      e.g. r04
Prints the DESIGN.md table of profiles/<round>_<config>.txt (tools/publish_profiles.py wrote them): kernel average,
algorithmic GB/s, fraction of 8 TB/s, HBM traffic over algorithmic bytes, VALU issue, clock, whole push (chains)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r05'
ORDER = ['C2', 'C2fast', 'C3', 'C4', 'C4ref', 'C5', 'C5old', 'scan8192', 'w256', 'w512', 'w1024', 'w2048', 'w8192', 'w16384', 'p1024',
         'p2048', 'p8192', 'p16384', 'chain256', 'chain512', 'chain1024', 'chain2048', 'chain4096', 'chain8192', 'chain16384']
print('| config | kernel | kernel µs | algorithmic GB/s | of 8 TB/s | traffic / algorithmic | VALU issue | clock GHz | whole push |')
print('|---|---|---|---|---|---|---|---|---|')
for cfg in ORDER:
    path = os.path.join(ROOT, 'profiles', '%s_%s.txt' % (rnd, cfg))
    if not os.path.exists(path):
        continue
    txt = open(path).read()
    m = re.search(r'kernel avg ([\d.]+) us = ([\d.]+) GB/s = ([\d.]+) % of 8000', txt)
    if not m:
        continue
    kern = re.search(r'^(?:void )?(?:oth::)?(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)', txt.split('\n')[2])
    tr = re.search(r'= ([\d.]+) x algorithmic', txt)
    vi = re.search(r'([\d.]+) GHz from GRBM_GUI_ACTIVE\) = (\d+) %', txt)
    wp = re.search(r'whole push .*?: ([\d.]+) GB/s = ([\d.]+) %', txt)
    print('| %s | `%s` | %s | %.0f | **%s %%** | %s | %s | %s | %s |' % (
        cfg, kern.group(1) if kern else '?', m.group(1), float(m.group(2)), m.group(3), tr.group(1) if tr else '-',
        (vi.group(2) + ' %') if vi else '-', vi.group(1) if vi else '-', ('**%s %%**' % wp.group(2)) if wp else ''))
