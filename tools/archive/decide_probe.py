#!/usr/bin/env python3
"""Time of the scanner's decision stage alone (oth_scan_decide_dev_out: moving average + noise floor, then mask and
channel sums) on rows that are already in HBM - BASELINE config 5's 64 rows of 16384 bins by default.
usage: decide_probe.py [rows] [nfft] [reps]        (OFDM_TOOLS_HIP_LIB selects an A/B library)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, scan_batch  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
ctx = _hip.Context(0)
bp = scan_batch.BatchScanPlan(ctx, N, 1000000, 15625.0, 10e3, thr_leveler=10)
lo, hi = bp._slices()
rng = np.random.default_rng(5)
psd = (rng.standard_exponential((rows, N)) * 1e-6).astype(np.float32)
o = ctx.alloc(rows * N * 4)
ctx.h2d(o, psd)
noise, power, mask = ctx.alloc(rows * 4), ctx.alloc(rows * max(len(lo), 1) * 4), ctx.alloc(rows * N)
run = lambda: ctx.scan_decide_dev_out(o, rows, N, bp.scanner.srch_bins, bp.thr_leveler, lo, hi, noise, power, mask)  # noqa: E731
for _ in range(20):
    run()
ctx.sync()
ctx.set_timing(True)
t0 = time.perf_counter()
for _ in range(reps):
    run()
ctx.sync()
host = (time.perf_counter() - t0) * 1e6 / reps
ms, n = ctx.get_timing()
print('%d rows x %d bins, M = %d: %.2f us of GPU time per decide call (HIP events around its two launches, %d scopes), '
      '%.1f us of host time per call' % (rows, N, int(bp.scanner.srch_bins), ms * 1e3 / max(n, 1), n, host))
