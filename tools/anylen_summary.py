#!/usr/bin/env python3
"""usage: anylen_summary.py <dir of tools/profile_anylen.sh passes> <config, e.g. w65536> <reps>
Per-call figures of a route that is several launches per call (csrc/fft_any.hip, fft_tl.hip): every kernel's rocprofv3
--kernel-trace --stats line, the sum per call of prof_driver.py's timed calls, algorithmic bytes / that time, and the HBM
bytes per call (FETCH_SIZE [KB] x 1024 x 2 - the gfx950 wide-read correction of MI355X_MICROARCH.md - + WRITE_SIZE [KB] x
1024, summed over the route's dispatches)."""
import collections
import csv
import glob
import re
import sys

d, cfg, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
log2n = 27
alg = 8 * 2 ** log2n
ROUTE = ('tl_', 'any_', 'finalize', 'pilot_mean', 'welch32k_kernel')
log = open(d + '/trace.log').read()
m = re.search(r'recipe: (.*)', log)
print('Welch %s-pt Hann 50 %% overlap + detrend, 2^%d samples (tools/prof_driver.py %s under rocprofv3, separate passes: '
      '--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE)' % (cfg[1:], log2n, cfg))
if m:
    print('recipe: ' + m.group(1))
m = re.search(r'timed kernels ([0-9.]+) ms per call', log)
timed = float(m.group(1)) if m else None
print()
stats = glob.glob(d + '/trace/*/*_kernel_stats.csv')
rows = list(csv.DictReader(open(stats[0]))) if stats else []
# the driver makes reps timed calls after a clock ramp of unknown length: per-call launch counts come from the kernel TRACE
trace = glob.glob(d + '/trace/*/*_kernel_trace.csv')
per_call = collections.OrderedDict()
if trace:
    ev = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r['Start_Timestamp']))
    fin = [i for i, r in enumerate(ev) if 'finalize' in r['Kernel_Name']]
    if len(fin) >= 2:      # one call = everything after the previous finalize up to and including this one: take the last call
        a, b = fin[-2] + 1, fin[-1] + 1
        for r in ev[a:b]:
            k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('oth::', ''))
            t = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            c = per_call.setdefault(k, [0, 0.0])
            c[0] += 1
            c[1] += t
        span = (int(ev[b - 1]['End_Timestamp']) - int(ev[a]['Start_Timestamp'])) / 1e3
for r in rows:
    if any(p in r['Name'] for p in ROUTE):
        print('%-70s calls %6s  avg %8.1f us' % (re.sub(r'\(.*', '', r['Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:70], r['Calls'], float(r['AverageNs']) / 1e3))
print()
if per_call:
    print('one call (the last of the run), launch by launch:')
    busy = 0.0
    for k, (n, t) in per_call.items():
        print('  %-50s x %4d  = %9.1f us' % (k[:50], n, t))
        busy += t
    print('  sum of kernel durations %.1f us; first start to last end %.1f us (launch gaps: %.1f us)' % (busy, span, span - busy))
    frac = alg / (span * 1e-6) / 1e9 / 8000.0
    print()
    print('roofline: algorithmic bytes per call %d / %.1f us = %.1f GB/s = %.1f %% of 8000 GB/s' % (alg, span, alg / (span * 1e-6) / 1e9, 100 * frac))
if timed:
    print('HIP-event time of the same calls in an unprofiled pass of the driver: %.4f ms per call = %.1f %% of 8000 GB/s'
          % (timed, alg / (timed * 1e-3) / 1e9 / 80.0))
tot = {}
for name in ('fetch', 'write'):
    fs = glob.glob('%s/%s/*/*_counter_collection.csv' % (d, name))
    if not fs:
        continue
    agg = collections.defaultdict(float)
    ncalls = collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        if any(p in r['Kernel_Name'] for p in ROUTE):
            k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('oth::', ''))
            agg[(k, r['Counter_Name'])] += float(r['Counter_Value'])
            ncalls[(k, r['Counter_Name'])] += 1
    tot[name] = (agg, ncalls)
if tot and per_call:
    print()
    print('HBM traffic per call (counter mean per dispatch x dispatches per call):')
    total = 0.0
    for name, (agg, ncalls) in tot.items():
        for (k, cn), v in sorted(agg.items()):
            mean = v / ncalls[(k, cn)]
            n = per_call.get(k, [0])[0]
            b = mean * n * (2048 if cn == 'FETCH_SIZE' else 1024)
            total += b
            print('  %-40s %-10s %10.4g KB per dispatch x %4d -> %.4g B' % (k[:40], cn, mean, n, b))
    print('  total %.4g B = %.2f x algorithmic (FETCH_SIZE x 2: the gfx950 wide-read correction; L2 / Infinity-Cache hits of the '
          'workspace do not reach these counters)' % (total, total / alg))
