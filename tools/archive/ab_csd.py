#!/usr/bin/env python3
"""Interleaved timing of csd4096 / welch16k launch parameters in one process.
usage: ab_csd.py [csd|16k] [rounds] sched:chunk[:tail] ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'csd'
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
variants = sys.argv[3:] or ['2:8', '2:4', '2:16', '0:8']
ctx = _hip.Context(0)
TONES = ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071))
if which == 'csd':
    n = 1 << 26
    dx, dy = ctx.alloc(n * 8), ctx.alloc(n * 8)
    ctx.synth_iq(dx, n, 1, TONES, 0.1 + 0.05j)
    ctx.synth_iq(dy, n, 2, TONES, 0.1 + 0.05j)
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0)
    nbytes = 16 * n

    import ctypes as C
    outs = [np.empty(4096, np.float32) for _ in range(3)] + [np.empty(8192, np.float32)]

    def run():
        nseg = C.c_uint64()
        ctx.check(ctx.lib.oth_csd_exec(plan.h, C.c_void_p(dx), C.c_void_p(dy), n, 1, _hip._fptr(outs[0]),
                                       _hip._fptr(outs[1]), _hip._fptr(outs[3]), _hip._fptr(outs[2]),
                                       C.byref(nseg)), 'csd')
else:
    nch, per = 64, 1 << 22
    n = nch * per
    dx = ctx.alloc(n * 8)
    dout = ctx.alloc(nch * 16384 * 4)
    ctx.synth_iq(dx, n, 1, TONES, 0.1 + 0.05j)
    plan = ctx.welch_plan(16384, window=None, noverlap=0, detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_OVER_N2, fftshift=True)
    nbytes = 8 * n

    def run():
        plan.exec_dev(dx, per, dout, nstreams=nch, stream_stride=per)


def select(v):
    f = v.split(':')
    sched, chunk, tail = f + ['2', '8', ''][len(f):]
    plan.set_tuning(None, int(sched), int(chunk), int(tail) if tail else 0)


for _ in range(100):
    run()
ctx.sync()
ctx.set_timing(True)
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        select(v)
        ctx.get_timing(reset=True)
        for _ in range(20):
            run()
        ctx.sync()
        ms, k = ctx.get_timing(reset=True)
        times[v].append(ms / k)
for v in variants:
    t = np.median(times[v])
    print('%-10s median %.4f ms  min %.4f -> %.0f GB/s (%.1f%% of 8 TB/s)' % (v, t, min(times[v]), nbytes / t / 1e6,
                                                                            nbytes / t / 1e6 / 80))
