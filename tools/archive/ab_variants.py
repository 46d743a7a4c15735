#!/usr/bin/env python3
"""Interleaved A/B timing of the welch4096 build variants in one process.
usage: ab_variants.py [log2_samples] [rounds] [variant[:sched[:chunk[:tail]]] ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
variants = sys.argv[3:] or ['dpp:0:8', 'dpp:2:8', 'pipe:0:8', 'pipe:2:8', 'pipe:2:16']


def select(v):
    f = v.split(':')
    tag, sched, chunk, tail = f + ['', '0', '8', ''][len(f):]
    plan.set_tuning(tag or None, int(sched), int(chunk), int(tail) if tail else 0)


n = 1 << log2n
ctx = _hip.Context(0)
d_in = ctx.alloc(n * 8)
d_out = ctx.alloc(4096 * 4)
ctx.synth_iq(d_in, n, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0, kernel=_hip.KERNEL_TUNED)
gen = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0, kernel=_hip.KERNEL_GENERIC)
ref = gen.exec_device_src(d_in, min(n, 1 << 24)).astype(np.float64)
burst = int(os.environ.get('AB_BURST', '20'))
times = {v: [] for v in variants}
ctx.set_timing(True)
# clock ramp
select(variants[0])
for _ in range(200):
    plan.exec_dev(d_in, n, d_out)
ctx.get_timing()
# interleaved rounds of sustained bursts (what bench.py does): `rounds` bursts of `burst` launches per variant
for r in range(rounds):
    for v in variants:
        select(v)
        for _ in range(burst):
            plan.exec_dev(d_in, n, d_out)
        ms, k = ctx.get_timing()
        times[v].append(ms / k)
for v in variants:
    select(v)
    got = plan.exec_device_src(d_in, min(n, 1 << 24)).astype(np.float64)
    err = float(np.max(np.abs(got - ref) / ref))
    t = sorted(times[v])
    med = t[len(t) // 2]
    print('%-16s burst-of-%d median %.4f ms  min %.4f  max %.4f -> %.0f GB/s (%.1f%% of 8 TB/s)  dev vs generic %.2e'
          % (v, burst, med, t[0], t[-1], 8.0 * n / med / 1e6, 8.0 * n / med / 1e6 / 80.0, err))
ctx.free(d_in)
ctx.free(d_out)
