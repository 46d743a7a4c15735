#!/usr/bin/env python3
"""usage: publish_profiles.py <tag> <round-prefix> [<dst-dir>]
Copy what tools/collect_all_profiles.sh (and tools/collect_profiles.sh for the bench command) left under
gpurun_out/ into profiles/ (tracked) as one text summary per configuration, with the roofline recomputation
written next to the counters, and refresh profiles/traffic.json (HBM bytes per launch of the headline kernel, read
by bench.py).   usage: publish_profiles.py <tag> <round-prefix, e.g. r02>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
# optional third argument: write there instead of profiles/ (collect_all_profiles.sh runs this on the GPU box, so that
# only the summaries - not the per-dispatch CSVs, which exceed gpurun's 64 MiB return limit - come back)
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

# config -> (kernel-name pattern, algorithmic bytes per launch, description)
CONFIGS = {
    'C2': ('welch4096ws', 8 * 2 ** 28, 'C2: 4096-pt Hann Welch, 50 % overlap, one 2^28-sample stream'),
    'C3': ('csd4096ws', 16 * 2 ** 26, 'C3: two-channel csd / coherence, 2 x 2^26 samples (16 B per sample pair); device outputs, launches back to back - the shape of bench.py::csd_bench (round 4 profiled the host-output call, whose idle gaps let the part run the kernel ~10 % faster than it sustains: profiles/r05_c3_bisect.txt)'),
    'C4': ('welch4096ws', 8 * 8 * 2 ** 25, 'C4: sweep of 8 x 2^25 samples on one GPU, Hann 4096, shift + trim + dB'),
    'C2fast': ('welch4096ws', 8 * 2 ** 28, 'C2 with OTH_DETREND_CONSTANT_FAST: the same launch on the build without the pilot (welch4096ws_kernel<true, false>), same box and passes'),
    'C4ref': ('welch4096_kernel', 8 * 8 * 2 ** 25, 'C4 reference-faithful: flattop, nperseg 1024 zero-padded to 4096 '
              '(spectrum_sweeper.py:263): 4 transforms per 2048 new samples'),
    'C5': ('welch16k1x_pipe', 8 * 64 * 2 ** 22, 'C5: 64 channel streams x 2^22 samples, 16384-pt rect |X|^2/N^2 mean (welch16k1x_pipe_kernel: '
           'one cross-wave exchange, software-pipelined, loads spread over the step)'),
    'C5old': ('welch16k_kernel', 8 * 64 * 2 ** 22, 'C5 on the round-3 kernel (welch16k_kernel<0, 4>: 4 x 4096, variant 16k4), same box and passes'),
    'w256': ('seg_kernel', 8 * 2 ** 27, 'Welch 256-pt Hann 50 % overlap, 2^27 samples (four 16-thread teams per wave)'),
    'w512': ('seg_kernel', 8 * 2 ** 27, 'Welch 512-pt Hann 50 % overlap, 2^27 samples (two 32-thread teams per wave)'),
    'w1024': ('segws_kernel', 8 * 2 ** 27, 'Welch 1024-pt Hann 50 % overlap, 2^27 samples'),
    'w2048': ('segws_kernel', 8 * 2 ** 27, 'Welch 2048-pt Hann 50 % overlap, 2^27 samples'),
    'w8192': ('welch8kws', 8 * 2 ** 27, 'Welch 8192-pt Hann 50 % overlap, 2^27 samples (welch8kws_kernel<2, true>, late round 5: the one-exchange transform split into eight producer and eight consumer waves one segment apart, window and pass-2 twiddles in registers, frequency-domain detrend, overlapped half kept in registers, new half prefetched; two transforms per sample)'),
    'scan8192': ('welch16k1x_pipe', 8 * 64 * 2 ** 22, 'the scanner\'s vectors at fft_len 8192: 64 channel streams x 2^22 samples, rect |X|^2/N^2 mean (welch16k1x_pipe_kernel<8>, round 5: the one-exchange pipelined loop on 8 waves, two workgroups per CU)'),
    'w16384': ('welch16k1x_half', 8 * 2 ** 27, 'Welch 16384-pt Hann 50 % overlap + detrend, 2^27 samples (welch16k1x_half_kernel<16, 2>: one cross-wave exchange, frequency-domain detrend, overlapped half kept in registers, new half prefetched, window from L2; two transforms per sample)'),
    'p8192': ('welch16k', 8 * 2 ** 27, 'the sweeper call at fft_len 8192: flattop, nperseg 2048 zero-padded to 8192, step 1024: 8 transforms per 8192 new samples (welch16k_kernel<1, 2, false, PAD>)'),
    'p16384': ('welch16k', 8 * 2 ** 27, 'the sweeper call at fft_len 16384: flattop, nperseg 4096 zero-padded to 16384, step 2048 (welch16k_kernel<1, 4, false, PAD>)'),
    'p1024': ('seg_kernel', 8 * 2 ** 27, 'the sweeper call at fft_len 1024 (spectrum_sweeper.py:263): flattop, nperseg 256 zero-padded to 1024, step 128: 8 transforms per 1024 new samples (seg_kernel<4, HALF, ., NA=4>)'),
    'p2048': ('seg_kernel', 8 * 2 ** 27, 'the sweeper call at fft_len 2048: flattop, nperseg 512 zero-padded to 2048, step 256 (seg_kernel<8, HALF, ., NA=4>)'),
    'chain256': ('seg_kernel', 8 * 2 ** 26, 'periodogram chain 256, 2^26 samples'),
    'chain512': ('seg_kernel', 8 * 2 ** 26, 'periodogram chain 512, 2^26 samples'),
    'chain1024': ('seg_kernel', 8 * 2 ** 26, 'periodogram chain 1024 (BH window, shift, |X|^2, IIR 0.8 + log), 2^26 samples'),
    'chain2048': ('seg_kernel', 8 * 2 ** 26, 'periodogram chain 2048, 2^26 samples'),
    'chain4096': ('seg_kernel', 8 * 2 ** 26, 'periodogram chain 4096, 2^26 samples'),
    'chain8192': ('chain16k', 8 * 2 ** 26, 'periodogram chain 8192 (BH window, shift, |X|^2, IIR 0.8 + log), 2^26 samples (chain16k_kernel<2, windowed>)'),
    'chain16384': ('chain16k', 8 * 2 ** 26, 'periodogram chain 16384 (BH window, shift, |X|^2, IIR 0.8 + log), 2^26 samples (chain16k_kernel<4, windowed>)'),
}


def one(src, pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)      # newest: a re-collected pass wins
    return hits[-1] if hits else None


def counters(src, pat):
    vals = {}
    for sub in ('sq1', 'sq2', 'fetch', 'write'):
        f = one(src, sub + '/*/*_counter_collection.csv')
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
                vals['_vgpr'] = r['VGPR_Count']
                vals['_grid'] = r['Grid_Size']
                vals['_wg'] = r['Workgroup_Size']
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
    return vals


written = []
for cfg, (pat, alg, desc) in CONFIGS.items():
    src = os.path.join(ROOT, 'gpurun_out', 'prof_%s_%s' % (tag, cfg))
    stats = one(src, 'trace/*/*_kernel_stats.csv')
    if not stats:
        continue
    lines = ['%s  (tools/prof_driver.py %s under rocprofv3, separate passes: --kernel-trace --stats | --pmc SQ.. | --pmc '
             'FETCH_SIZE | --pmc WRITE_SIZE..)' % (desc, cfg), '']
    avg_ns = None
    tail_ns = 0.0      # chains: the cross-team reduction + state kernels that close every push
    for r in csv.DictReader(open(stats)):
        lines.append('%-92s calls %5s  avg %10.1f us' % (r['Name'][:92], r['Calls'], float(r['AverageNs']) / 1e3))
        if pat in r['Name'] and avg_ns is None:
            avg_ns = float(r['AverageNs'])
        if cfg.startswith('chain') and ('chain_reduce_kernel' in r['Name'] or 'chain_state_kernel' in r['Name']
                                        or 'chain_tail_kernel' in r['Name']):
            tail_ns += float(r['AverageNs'])
    v = counters(src, pat)
    lines.append('')
    for k in sorted(v):
        if not k.startswith('_'):
            lines.append('%-24s %.4g' % (k, v[k]))
    lines.append('VGPR_Count %s  Grid_Size %s  Workgroup_Size %s' % (v.get('_vgpr'), v.get('_grid'), v.get('_wg')))
    if 'SQ_WAVE_CYCLES' in v:
        wc = v['SQ_WAVE_CYCLES']
        lines.append('wave time split: wait_any %.1f%%  wait_inst_any %.1f%%  active %.1f%%' % (
            100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_ANY'] / wc))
    lines.append('')
    if avg_ns:
        gbps = alg / avg_ns
        lines.append('roofline: algorithmic bytes per launch %d / kernel avg %.1f us = %.1f GB/s = %.1f %% of 8000 GB/s'
                     % (alg, avg_ns / 1e3, gbps, gbps / 80.0))
        if tail_ns:
            whole = avg_ns + tail_ns
            lines.append('whole push (transform kernel %.1f us + cross-team reduction and state kernels %.1f us = %.1f us): '
                         '%.1f GB/s = %.1f %% of 8000 GB/s' % (avg_ns / 1e3, tail_ns / 1e3, whole / 1e3, alg / whole,
                                                                 alg / whole / 80.0))
    if 'FETCH_SIZE' in v:
        rd = v['FETCH_SIZE'] * 2048.0
        wr = v.get('WRITE_SIZE', 0.0) * 1024.0
        lines.append('HBM traffic per launch: read %.4g B (FETCH_SIZE KB x 1024 x 2, the gfx950 wide-read correction of '
                     'MI355X_MICROARCH.md) + write %.4g B (WRITE_SIZE KB x 1024) = %.3f x algorithmic'
                     % (rd, wr, (rd + wr) / alg))
        if cfg in ('C2', 'C3', 'C4', 'C4ref', 'C5'):      # bench.py reads these (roofline.traffic of the line and its sub-objects)
            tpath = os.path.join(dst, 'traffic.json')
            try:
                tj = json.load(open(tpath))
                if 'C2' not in tj and 'hbm_bytes_per_launch' in tj:      # round 1-4 layout: the C2 entry alone
                    tj = {'C2': tj}
            except (OSError, ValueError):
                tj = {}
            tj[cfg] = {'kernel': pat, 'hbm_bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr,
                       'ratio_to_algorithmic': (rd + wr) / alg,
                       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                                 '(tools/pmc_passes.sh); FETCH_SIZE [KB] x 1024 x 2 (gfx950 wide-read correction, '
                                 'MI355X_MICROARCH.md HBM section); WRITE_SIZE [KB] x 1024; per-launch mean',
                       'algorithmic_bytes_per_launch': alg, 'source': 'profiles/%s_%s.txt' % (rnd, cfg)}
            json.dump(tj, open(tpath, 'w'), indent=1)
    valu_pct = None
    if 'SQ_INSTS_VALU' in v and avg_ns:
        # issue slots: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles; the clock from GRBM_GUI_ACTIVE / 8 XCDs
        clk = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / (avg_ns * 1e-9) if v.get('GRBM_GUI_ACTIVE') else 2.0e9
        slots = 1024 * clk * avg_ns * 1e-9 / 2.0
        valu_pct = 100 * v['SQ_INSTS_VALU'] / slots
        lines.append('VALU issue: %.4g wave-instructions per launch / %.4g issue slots (1024 SIMDs, 2 cycles each, '
                     '%.2f GHz from GRBM_GUI_ACTIVE) = %.0f %%' % (v['SQ_INSTS_VALU'], slots, clk / 1e9, valu_pct))
        hbm_pct = alg / avg_ns / 80.0
        wait_pct = 100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'] if 'SQ_WAVE_CYCLES' in v else 0.0
        act_pct = 100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES'] if 'SQ_WAVE_CYCLES' in v else 100.0
        if wait_pct > act_pct:
            # the waves spend more time waiting than issuing: naming VALU issue as the ceiling would be wrong (round 4 verdict)
            lines.append('binding: STALLS, not issue - the waves wait %.0f %% of their time and issue %.0f %% of it (VALU issue %.0f %% '
                         'of the slots); at 100 %% issue this launch would take %.1f us = %.1f %% of 8000 GB/s, which it does not '
                         'approach' % (wait_pct, act_pct, valu_pct, avg_ns / 1e3 * valu_pct / 100, hbm_pct * 100 / valu_pct))
        elif valu_pct > 1.5 * hbm_pct:
            lines.append('binding ceiling: VALU issue (%.0f %% of the issue slots against %.1f %% of the byte roofline): at 100 %% '
                         'issue this launch would take %.1f us = %.1f %% of 8000 GB/s'
                         % (valu_pct, hbm_pct, avg_ns / 1e3 * valu_pct / 100, hbm_pct * 100 / valu_pct))
    out = os.path.join(dst, '%s_%s.txt' % (rnd, cfg))
    open(out, 'w').write('\n'.join(lines) + '\n')
    st = one(src, 'trace/*/*_kernel_stats.csv')
    written.append(out)
print('\n'.join(written))

# the bench command itself (tools/collect_profiles.sh <tag>)
src = os.path.join(ROOT, 'gpurun_out', 'profiles_' + tag)
if os.path.isdir(src):
    import shutil
    for name, out in (('bench_unprofiled.json', rnd + '_bench.json'), ('bench_under_rocprof.json', rnd + '_bench_under_rocprof.json')):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, out))
    st = one(src, 'trace/*/*_kernel_stats.csv')
    if st:
        shutil.copy(st, os.path.join(dst, rnd + '_bench_kernel_stats.csv'))
        print(os.path.join(dst, rnd + '_bench_kernel_stats.csv'))
