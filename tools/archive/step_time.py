#!/usr/bin/env python3
"""usage: step_time.py nfft [nfft ...]   - whole pipelined Welch step (exec_dev, device output; every launch of the step) at
2^27 samples, Hann, 50 % overlap, default plan: median of 8 rounds of 25 steps.  OFDM_TOOLS_HIP_LIB selects an A/B library."""
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
for N in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    plan = ctx.welch_plan(N, window=windows.get_window('hann', N), fs=1.0)
    for _ in range(10):
        plan.exec_dev(d, n, o)
    ctx.sync()
    t = []
    for rnd in range(8):
        t0 = time.perf_counter()
        for _ in range(25):
            plan.exec_dev(d, n, o)
        ctx.sync()
        t.append((time.perf_counter() - t0) * 1e3 / 25)
    ms = sorted(t)[len(t) // 2]
    print('%5d  %.4f ms per step = %.1f %% of 8 TB/s (whole step)   %s' % (N, ms, 8 * n / ms / 1e6 / 8000 * 100,
                                                                       plan.last_recipe().split(' sched')[0]), flush=True)
