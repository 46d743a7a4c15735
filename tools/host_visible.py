#!/usr/bin/env python3
"""Host-visible Welch step ("PSD available on host"): one blocking oth_welch_exec(src_is_device=1) per
step, host clock around the call, median.  Next to it the pipelined device-output step (bench.py's `ms_per_step`) and
the A/B legs of round 5:
  plaunch        the pilot from its own launch in front of the transform (round 4) instead of the kernel's prologue
  OTH_HOSTWAIT=sync (environment, read once per process)   hipStreamSynchronize instead of polling the completion word
usage: host_visible.py [log2_samples ...]      (default 28 20)
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, windows  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [28, 20]
ctx = _hip.Context(0)
nmax = 1 << max(sizes)
d, o = ctx.alloc(nmax * 8), ctx.alloc(4096 * 4)
ctx.synth_iq(d, nmax, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
hann = windows.get_window('hann', 4096)
print('wait mode: %s' % (os.environ.get('OTH_HOSTWAIT') or 'poll'))
for lg in sizes:
    n = 1 << lg
    reps = 80 if lg >= 26 else 400
    plans = {}
    for tag in ('default', 'plaunch'):
        plans[tag] = ctx.welch_plan(4096, window=hann, fs=1.0)
        if tag != 'default':
            plans[tag].set_tuning(tag)
        for _ in range(10):
            plans[tag].exec_device_src(d, n)
    t = {tag: [] for tag in plans}
    for _ in range(reps):                      # interleaved: both legs see the same clock / power state
        for tag, plan in plans.items():
            t0 = time.perf_counter()
            plan.exec_device_src(d, n)
            t[tag].append((time.perf_counter() - t0) * 1e3)
    row = []
    for tag, plan in plans.items():
        # the same step with device output, back to back (pipelined)
        for _ in range(10):
            plan.exec_dev(d, n, o)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.exec_dev(d, n, o)
        ctx.sync()
        pipe = (time.perf_counter() - t0) * 1e3 / reps
        row.append('%s: host-visible %.4f ms (min %.4f), pipelined %.4f ms' % (tag, statistics.median(t[tag]), min(t[tag]), pipe))
        plan.close()
    print('2^%d samples   ' % lg + '   |   '.join(row), flush=True)
ctx.free(d)
ctx.free(o)
