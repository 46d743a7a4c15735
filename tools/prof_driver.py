#!/usr/bin/env python3
"""Profiling driver: a few launches of the Welch kernel on the C2 workload, nothing else
(run under rocprofv3 --kernel-trace or --pmc).  usage: prof_driver.py [log2_samples] [reps] [kernel]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, windows  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else _hip.KERNEL_AUTO
n = 1 << log2n
ctx = _hip.Context(0)
d_in = ctx.alloc(n * 8)
d_out = ctx.alloc(4096 * 4)
ctx.synth_iq(d_in, n, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0, kernel=kernel)
ctx.set_timing(True)
for _ in range(reps):
    plan.exec_dev(d_in, n, d_out)
ms, k = ctx.get_timing()
probe = ctx.stream_read_probe(d_in, n * 8, 2)
print('kernel avg ms %.4f over %d launches -> %.1f GB/s; read probe %.1f GB/s' % (ms / k, k, 8.0 * n / (ms / k) / 1e6, 8.0 * n / probe / 1e6))
ctx.free(d_in)
ctx.free(d_out)
