#!/usr/bin/env python3
"""Interleaved A/B timing of the 256 ... 2048-point Welch builds and launch parameters in one process (AB_LOG2N: samples).
usage: ab_seg.py nfft rounds build:sched:chunk ...   (build = seg3 | seg4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

nfft = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
variants = sys.argv[3:] or ['segws:2:16', 'segws:2:32', 'seg3:2:32', 'segws:0:16', 'segws:2:8']
ctx = _hip.Context(0)
n = 1 << int(os.environ.get('AB_LOG2N', '27'))
d, o = ctx.alloc(n * 8), ctx.alloc(nfft * 4)
ctx.synth_iq(d, n, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
plan = ctx.welch_plan(nfft, window=windows.get_window('hann', nfft), fs=1.0)


def select(v):
    b, sched, chunk = (v.split(':') + ['2', '16'])[:3]
    plan.set_tuning(b, int(sched), int(chunk), 0)


for _ in range(100):
    plan.exec_dev(d, n, o)
ctx.sync()
ctx.set_timing(True)
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        select(v)
        ctx.get_timing(reset=True)
        for _ in range(20):
            plan.exec_dev(d, n, o)
        ctx.sync()
        ms, k = ctx.get_timing(reset=True)
        times[v].append(ms / k)
for v in variants:
    t = np.median(times[v])
    print('%-14s median %.4f ms  min %.4f -> %.0f GB/s (%.1f%% of 8 TB/s)' % (v, t, min(times[v]), 8 * n / t / 1e6, 8 * n / t / 1e6 / 80))
