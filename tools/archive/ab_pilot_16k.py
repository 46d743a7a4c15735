#!/usr/bin/env python3
"""Whole-step time (pipelined exec_dev, device output) of the 8192 / 16384-point Welch at 50 % overlap with the pilot formed
in the kernel's prologue (default) or by its own launch (tuning variant "plaunch"), interleaved in one process."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, windows  # noqa: E402

ctx = _hip.Context(0)
n = 1 << 27
d, o = ctx.alloc(n * 8), ctx.alloc(16384 * 4)
ctx.synth_iq(d, n, 1002, ((0.5, 0.1234), (2.0, 0.4071)), 0.1 + 0.05j)
for N in (8192, 16384):
    plans = {}
    for tag in ('inline', 'plaunch'):
        plans[tag] = ctx.welch_plan(N, window=windows.get_window('hann', N), fs=1.0)
        if tag == 'plaunch':
            plans[tag].set_tuning('plaunch')
        for _ in range(5):
            plans[tag].exec_dev(d, n, o)
    ctx.sync()
    t = {k: [] for k in plans}
    for rnd in range(6):
        for tag, plan in plans.items():
            t0 = time.perf_counter()
            for _ in range(25):
                plan.exec_dev(d, n, o)
            ctx.sync()
            t[tag].append((time.perf_counter() - t0) * 1e3 / 25)
    for tag, plan in plans.items():
        print('%5d %-8s %.4f ms per step (min %.4f)   %s' % (N, tag, sorted(t[tag])[len(t[tag]) // 2], min(t[tag]),
                                                             plan.last_recipe().split(' sched')[0]), flush=True)
