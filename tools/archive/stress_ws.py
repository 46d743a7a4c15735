#!/usr/bin/env python3
"""Randomised stress of the tuned welch4096 builds against the generic kernel: segment counts, stream counts,
schedules and chunk sizes drawn at random for a given number of seconds.  usage: stress_ws.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _hip.Context(0)
nmax = 4096 + 2048 * 20000
d_in = ctx.alloc(4 * nmax * 8)
d_a, d_b = ctx.alloc(4 * 4096 * 4), ctx.alloc(4 * 4096 * 4)
ctx.synth_iq(d_in, 4 * nmax, 3, ((0.5, 0.1234), (2.0, 0.4071)), 0.3 - 0.2j)
w = windows.get_window('hann', 4096)
plans = {d: (ctx.welch_plan(4096, window=w, detrend=d, kernel=_hip.KERNEL_TUNED),
             ctx.welch_plan(4096, window=w, detrend=d, kernel=_hip.KERNEL_GENERIC))
         for d in (_hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST, _hip.DETREND_NONE)}
t0, n_cases, worst = time.time(), 0, 0.0
while time.time() - t0 < secs:
    nseg = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(1, 3000)), int(rng.integers(3000, 20000))]))
    n = 4096 + 2048 * (nseg - 1) + int(rng.integers(0, 2048))
    ns = int(rng.integers(1, 5))
    det = int(rng.choice([_hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST, _hip.DETREND_NONE]))      # pilot builds, raw-sample builds, none
    variant = str(rng.choice(['ws', 'ws', 'pipe', 'dpp']))
    chunk = int(rng.choice([0, 1, 2, 3, 4, 7, 20, 33]))
    tuned, gen = plans[det]
    tuned.set_tuning(variant, chunk=chunk)
    tuned.set_schedule(int(rng.integers(0, 3)))
    assert tuned.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
    assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
    a = ctx.d2h(d_a, (ns, 4096), np.float32).astype(np.float64)
    b = ctx.d2h(d_b, (ns, 4096), np.float32).astype(np.float64)
    err = float(np.max(np.abs(a - b) / np.maximum(b, 0.1 * np.median(b))))
    worst = max(worst, err)
    assert err < 5e-5, (nseg, ns, det, variant, chunk, err)
    n_cases += 1
    if n_cases % 200 == 0:
        print('%d cases, worst %.2e' % (n_cases, worst), flush=True)
print('done: %d cases in %.0f s, worst deviation %.2e' % (n_cases, time.time() - t0, worst))
