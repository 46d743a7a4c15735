#!/usr/bin/env python3
"""Throughput of every BASELINE config on one MI355X, device-resident inputs, HIP-event kernel time.
Writes one JSON line per config (not the driver's bench: that is bench.py)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

TONES = ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071))
DC = 0.1 + 0.05j
ctx = _hip.Context(0)
REPS = int(os.environ.get('REPS', '20'))


def timed(fn, reps=REPS, warm=30):
    for _ in range(warm):
        fn()
    ctx.sync()
    ctx.set_timing(True)
    ctx.get_timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    wall = (time.perf_counter() - t0) / reps
    ms, k = ctx.get_timing()
    ctx.set_timing(False)
    return wall * 1e3, ms / max(k, 1) * (k / reps)      # wall ms per call, FFT-kernel ms per call


def report(name, nsamples, bytes_per_sample, wall_ms, kern_ms, **extra):
    d = dict(config=name, samples=nsamples, wall_ms=round(wall_ms, 4), kernel_ms=round(kern_ms, 4),
             Msamples_per_s=round(nsamples / wall_ms / 1e3, 1),
             kernel_GBps=round(bytes_per_sample * nsamples / kern_ms / 1e6, 1),
             frac_of_8TBps=round(bytes_per_sample * nsamples / kern_ms / 1e6 / 8000.0, 4))
    d.update(extra)
    print(json.dumps(d), flush=True)


def ramp():
    n = 1 << 26
    d = ctx.alloc(n * 8)
    o = ctx.alloc(4096 * 4)
    ctx.synth_iq(d, n, 1, TONES, DC)
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(8):
            plan.exec_dev(d, n, o)
        ctx.sync()
    ctx.free(d)
    ctx.free(o)


ramp()
which = sys.argv[1:] or ['C1', 'C2', 'C3', 'C4', 'C4ref', 'C5', 'H2D']

if 'C2' in which:
    n = 1 << 28
    d = ctx.alloc(n * 8)
    o = ctx.alloc(4096 * 4)
    ctx.synth_iq(d, n, 1002, TONES, DC)
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096))
    w, k = timed(lambda: plan.exec_dev(d, n, o))
    report('C2 welch hann 4096 50% 2^28', n, 8, w, k)
    gen = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), kernel=_hip.KERNEL_GENERIC)
    w, k = timed(lambda: gen.exec_dev(d, n, o), reps=3, warm=3)
    report('C2 (generic Stockham kernel)', n, 8, w, k)
    ctx.free(d)
    ctx.free(o)

if 'H2D' in which:
    # host buffer -> PSD on the host (pageable numpy memory through oth_welch_exec): PCIe-inclusive
    n = 1 << 26
    x = np.zeros(n, np.complex64)
    x.real = 1.0
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096))
    plan.exec(x)
    t0 = time.perf_counter()
    for _ in range(3):
        plan.exec(x)
    w = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps(dict(config='C2 host->host (PCIe-inclusive, pageable numpy buffer, 2^26 samples)', samples=n,
                          wall_ms=round(w, 3), Msamples_per_s=round(n / w / 1e3, 1),
                          GBps=round(8.0 * n / w / 1e6, 2))), flush=True)

if 'C1' in which:
    n = 1 << 20
    x = np.zeros(n, np.complex64)
    d = ctx.alloc(n * 8)
    o = ctx.alloc(1024 * 4)
    ctx.synth_iq(d, n, 1001, TONES, DC)
    # 1024-pt rect, no overlap, |X|^2/N^2 averaged = welch plan with OVER_N2 scaling (device resident)
    plan = ctx.welch_plan(1024, noverlap=0, window=None, detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_OVER_N2,
                          fftshift=True)
    w, k = timed(lambda: plan.exec_dev(d, n, o))
    report('C1 v2 chain 1024 rect mean (device resident)', n, 8, w, k)
    ctx.free(d)
    ctx.free(o)

if 'C3' in which:
    n = 1 << 26
    dx, dy = ctx.alloc(n * 8), ctx.alloc(n * 8)
    ctx.synth_iq(dx, n, 1003, TONES, DC)
    ctx.synth_iq(dy, n, 1004, TONES, DC)
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096))
    import ctypes as C
    outs = [np.empty(4096, np.float32) for _ in range(3)] + [np.empty(8192, np.float32)]

    def run():
        nseg = C.c_uint64()
        ctx.check(ctx.lib.oth_csd_exec(plan.h, C.c_void_p(dx), C.c_void_p(dy), n, 1, _hip._fptr(outs[0]),
                                       _hip._fptr(outs[1]), _hip._fptr(outs[3]), _hip._fptr(outs[2]),
                                       C.byref(nseg)), 'csd')
    w, k = timed(run, reps=10, warm=30)
    report('C3 csd/coherence hann 4096 2x2^26', n, 16, w, k)
    ctx.free(dx)
    ctx.free(dy)

for tag, kw in (('C4', dict(window=windows.get_window('hann', 4096))),
                ('C4ref', dict(nperseg=1024, window=windows.get_window('flattop', 1024)))):
    if tag in which:
        S, nseg_rf = 1 << 25, 8
        d = ctx.alloc(nseg_rf * S * 8)
        o = ctx.alloc(nseg_rf * 3584 * 4)
        for i in range(nseg_rf):
            ctx.synth_iq(d + i * S * 8, S, 2000 + i, TONES, DC)
        plan = ctx.welch_plan(4096, fs=2.0e6, fftshift=True, trim_bins=256, db=True, **kw)
        w, k = timed(lambda: plan.exec_dev(d, S, o, nstreams=nseg_rf, stream_stride=S), reps=10, warm=30)
        report('%s sweep 8 x 2^25, 1 GPU (%s)' % (tag, 'hann 4096' if tag == 'C4' else 'flattop nperseg 1024 -> 4096'),
               nseg_rf * S, 8, w, k)
        ctx.free(d)
        ctx.free(o)

if 'C5' in which:
    nch, S, N = 64, 1 << 22, 16384
    d = ctx.alloc(nch * S * 8)
    o = ctx.alloc(nch * N * 4)
    for i in range(nch):
        ctx.synth_iq(d + i * S * 8, S, 3000 + i, TONES, DC)
    plan = ctx.welch_plan(N, noverlap=0, window=None, detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_OVER_N2,
                          fftshift=True)
    w, k = timed(lambda: plan.exec_dev(d, S, o, nstreams=nch, stream_stride=S), reps=10, warm=30)
    report('C5 scanner 64 ch x 2^22, 16384-pt rect mean', nch * S, 8, w, k)
    ctx.free(d)
    ctx.free(o)
