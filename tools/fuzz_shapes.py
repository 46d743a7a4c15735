#!/usr/bin/env python3
"""Random plan shapes: the tuned route (whatever resolve_recipe picks) against the coverage kernel on the same
device-resident samples.  usage: fuzz_shapes.py [cases] [seed]      (prints every mismatch above 3e-5, exits 1 if any)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, windows  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _hip.Context(0)
cap = 1 << 23
d = ctx.alloc(cap * 8)
ctx.synth_iq(d, cap, 99, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.3 - 0.2j)
out = ctx.alloc(64 * 16384 * 4)
bad = 0
for it in range(cases):
    nfft = int(rng.choice([256, 512, 1024, 2048, 4096, 8192, 16384]))
    nperseg = int(rng.choice([nfft, nfft, nfft // 2, nfft // 4]))
    nov = int(rng.choice([0, nperseg // 2, nperseg // 2, int(rng.integers(0, nperseg))]))
    step = nperseg - nov
    wname = str(rng.choice(['hann', 'flattop', 'rect', 'hamming']))
    # 'hamming': symmetric, 1/k sidelobes - its spectrum is not confined, i.e. the time-domain detrend builds
    win = None if wname == 'rect' else (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(nperseg) / (nperseg - 1)) if wname == 'hamming'
                                        else windows.get_window(wname, nperseg))
    det = int(rng.choice([_hip.DETREND_NONE, _hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST]))
    nstreams = int(rng.choice([1, 1, 3, 17, 64]))
    nseg = int(rng.choice([1, 2, 7, 33, 257, 1000, 4000]))
    n = nperseg + step * (nseg - 1) + int(rng.integers(0, step))
    stride = n + int(rng.integers(0, 3)) * 64
    if stride * nstreams + 8 > cap:
        nstreams = max(1, (cap - 8) // stride)
        if stride + 8 > cap:
            continue
    base = d + 8 * int(rng.integers(0, 8))      # any sample offset: nothing may assume more than the 8-byte alignment of a sample
    sched = int(rng.choice([-1, -1, _hip.SCHED_CONTIGUOUS, _hip.SCHED_INTERLEAVED, _hip.SCHED_DYNAMIC]))
    chunk = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 16, 32]))
    kw = dict(nperseg=nperseg, noverlap=nov, window=win, detrend=det, fs=1.0)
    try:
        a = ctx.welch_plan(nfft, kernel=_hip.KERNEL_AUTO, **kw)
        b = ctx.welch_plan(nfft, kernel=_hip.KERNEL_GENERIC, **kw)
        if sched >= 0:
            a.set_schedule(sched)
        if chunk:
            a.set_tuning(None, chunk=chunk)
        ka = a.exec_dev(base, n, out, nstreams=nstreams, stream_stride=stride)
        ga = ctx.d2h(out, (nstreams, nfft), np.float32).astype(np.float64)
        rec = a.last_recipe()
        kb = b.exec_dev(base, n, out, nstreams=nstreams, stream_stride=stride)
        gb = ctx.d2h(out, (nstreams, nfft), np.float64 if False else np.float32).astype(np.float64)
        a.close(), b.close()
    except _hip.HipError as e:
        print('case %d: %s  (nfft %d nperseg %d nov %d %s det %d streams %d n %d)' % (it, e, nfft, nperseg, nov, wname, det, nstreams, n))
        bad += 1
        continue
    # (a rectangular window with a constant detrend leaves bin 0 at rounding noise in both kernels: floor the divisor)
    few = nseg * step < 7 * nperseg      # fewer than seven segments' worth of NEW samples: as good as a lone periodogram
    floor = (1e-3 if few else 1e-5) * np.median(gb, axis=1, keepdims=True)      # (deep nulls of few periodograms are fp32 noise)
    err = float(np.max(np.abs(ga - gb) / np.maximum(gb, floor)))
    # a lone periodogram / the raw-sample detrend under a line sit at the fp32 floor in both kernels: looser there
    bound = 2e-3 if few else 3e-5
    flag = '' if (err < bound and ka == kb == nseg) else '   <-- MISMATCH'
    if flag:
        bad += 1
    if flag or it % 25 == 0:
        print('case %3d nfft %5d nperseg %5d nov %5d %-7s det %d streams %2d nseg %4d sched %2d chunk %2d  err %.1e  %s%s' % (
            it, nfft, nperseg, nov, wname, det, nstreams, nseg, sched, chunk, err, rec.split(' nfft')[0], flag), flush=True)
# ---- two-channel form: y = the same buffer a few samples in
for it in range(cases // 4):
    nfft = int(rng.choice([256, 1024, 2048, 4096, 4096, 8192]))
    nov = int(rng.choice([0, nfft // 2, nfft // 2, int(rng.integers(0, nfft))]))
    step = nfft - nov
    wname = str(rng.choice(['hann', 'flattop', 'hamming']))
    win = (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(nfft) / (nfft - 1)) if wname == 'hamming' else windows.get_window(wname, nfft))
    det = int(rng.choice([_hip.DETREND_NONE, _hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST]))
    nseg = int(rng.choice([7, 33, 257, 1000, 3000]))
    n = nfft + step * (nseg - 1) + int(rng.integers(0, step))
    delay = int(rng.integers(1, 9))
    if n + delay > cap or nseg * step < 7 * nfft:
        continue
    try:
        a = ctx.welch_plan(nfft, noverlap=nov, window=win, detrend=det, fs=1.0, kernel=_hip.KERNEL_AUTO)
        b = ctx.welch_plan(nfft, noverlap=nov, window=win, detrend=det, fs=1.0, kernel=_hip.KERNEL_GENERIC)
        ra = a.csd_device_src(d + 8 * delay, d, n)
        rec = a.last_recipe()
        rb = b.csd_device_src(d + 8 * delay, d, n)
        a.close(), b.close()
    except _hip.HipError as e:
        print('csd case %d: %s' % (it, e))
        bad += 1
        continue
    norm = np.sqrt(rb[0].astype(np.float64) * rb[1])
    errs = [float(np.max(np.abs(ra[0].astype(np.float64) - rb[0]) / rb[0])), float(np.max(np.abs(ra[1].astype(np.float64) - rb[1]) / rb[1])),
            float(np.max(np.abs(ra[2].astype(np.complex128) - rb[2]) / norm))]
    flag = '' if max(errs) < 3e-5 else '   <-- MISMATCH'
    if flag:
        bad += 1
    if flag or it % 10 == 0:
        print('csd %3d nfft %5d nov %5d %-7s det %d nseg %4d delay %d  err %.1e %.1e %.1e  %s%s' % (
            it, nfft, nov, wname, det, nseg, delay, errs[0], errs[1], errs[2], rec.split(' nfft')[0], flag), flush=True)
print('%d cases, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
