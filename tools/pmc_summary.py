#!/usr/bin/env python3
"""Summarise the CSVs that tools/pmc_passes.sh leaves under gpurun_out/<dir> (per-launch averages)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'welch4096'
for f in glob.glob(d + '/trace/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print('%-90s calls %s avg %.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
vals = {}
for sub in ('sq1', 'sq2', 'fetch', 'write'):
    for f in glob.glob('%s/%s/*/*_counter_collection.csv' % (d, sub)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
                vals['VGPR'] = r['VGPR_Count']
                vals['LDS'] = r['LDS_Block_Size']
                vals['grid'] = r['Grid_Size']
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
for k in sorted(vals):
    print('%-24s %s' % (k, ('%.4g' % vals[k]) if isinstance(vals[k], float) else vals[k]))
if 'SQ_WAVE_CYCLES' in vals:
    wc = vals['SQ_WAVE_CYCLES']
    print('wave time split: wait_any %.1f%%  wait_inst_any %.1f%%  active %.1f%%' % (
        100 * vals['SQ_WAIT_ANY'] / wc, 100 * vals['SQ_WAIT_INST_ANY'] / wc, 100 * vals['SQ_ACTIVE_INST_ANY'] / wc))
if 'FETCH_SIZE' in vals:
    print('HBM read bytes (FETCH_SIZE KB x 1024 x 2 gfx950 correction): %.4g' % (vals['FETCH_SIZE'] * 2048))
if 'WRITE_SIZE' in vals:
    print('HBM write bytes (WRITE_SIZE KB x 1024): %.4g' % (vals['WRITE_SIZE'] * 1024))
