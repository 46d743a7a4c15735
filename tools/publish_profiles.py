#!/usr/bin/env python3
"""Copy what tools/collect_profiles.sh left under gpurun_out/profiles_<tag>/ into profiles/ (tracked) and
derive profiles/traffic.json (HBM bytes per launch of the headline kernel, read by bench.py).
usage: publish_profiles.py <tag> <round-prefix, e.g. r01>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, 'gpurun_out', 'profiles_' + tag)
dst = os.path.join(ROOT, 'profiles')


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)))
    assert hits, pattern
    return hits[-1]


shutil.copy(one('trace/*/*_kernel_stats.csv'), os.path.join(dst, rnd + '_bench_kernel_stats.csv'))
shutil.copy(os.path.join(src, 'bench_unprofiled.json'), os.path.join(dst, rnd + '_bench.json'))
shutil.copy(os.path.join(src, 'bench_under_rocprof.json'), os.path.join(dst, rnd + '_bench_under_rocprof.json'))
for sub in ('sq1', 'sq2', 'fetch', 'write'):
    shutil.copy(one(sub + '/*/*_counter_collection.csv'), os.path.join(dst, '%s_pmc_%s_counter_collection.csv' % (rnd, sub)))
shutil.copy(os.path.join(src, 'summary_welch4096.txt'), os.path.join(dst, rnd + '_pmc_welch4096.txt'))
shutil.copy(os.path.join(src, 'summary_read_probe.txt'), os.path.join(dst, rnd + '_pmc_read_probe_calibration.txt'))


def mean_counter(sub, counter, kernel):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(one(sub + '/*/*_counter_collection.csv'))):
        if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
            vals[r['Kernel_Name']].append(float(r['Counter_Value']))
    assert len(vals) == 1, (kernel, list(vals))
    name, v = next(iter(vals.items()))
    return name, sum(v) / len(v)


name, fetch_kb = mean_counter('fetch', 'FETCH_SIZE', 'welch4096')
_, write_kb = mean_counter('write', 'WRITE_SIZE', 'welch4096')
_, probe_kb = mean_counter('fetch', 'FETCH_SIZE', 'read_probe')
read_b, write_b = fetch_kb * 1024 * 2, write_kb * 1024
bench = json.load(open(os.path.join(src, 'bench_unprofiled.json')))
alg = bench['roofline']['algorithmic_bytes_per_launch']
json.dump({
    'log2_samples': 28,
    'kernel': name,
    'hbm_bytes_per_launch': read_b + write_b,
    'read_bytes': read_b,
    'write_bytes': write_b,
    'ratio_to_algorithmic': (read_b + write_b) / alg,
    'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh); '
              'FETCH_SIZE [KB] x 1024 x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); '
              'WRITE_SIZE [KB] x 1024; per-launch mean over the pass',
    'calibration': 'read_probe_kernel (float4 stream of exactly 2^31 bytes) in the same pass: FETCH_SIZE %.6g KB '
                   '-> x2048 = %.6g bytes (expected 2147483648)' % (probe_kb, probe_kb * 2048),
    'algorithmic_bytes_per_launch': alg,
    'source': 'profiles/%s_pmc_fetch_counter_collection.csv, profiles/%s_pmc_write_counter_collection.csv' % (rnd, rnd),
}, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print(open(os.path.join(dst, 'traffic.json')).read())
