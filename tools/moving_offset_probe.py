#!/usr/bin/env python3
"""Per seed and case: what the default plan, the time-domain builds ('td') and the reference's own float32 arithmetic
(SciPy on complex64, oracle.welch_c64) lose against float64 on the inputs of
tests/test_hip_parity.py::test_pilot_under_a_transient_and_a_drifting_offset (4096-point Hann, 2047 segments).
usage (GPU box): python tools/moving_offset_probe.py > gpurun_out/moving_offset.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from oracle import ref_cpu as R  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / np.abs(b)))


ctx = _hip.Context(0)
N, nseg = 4096, 2047
n = N + (N // 2) * (nseg - 1)
opening = np.zeros(n)
opening[:N] = 1.0
ramp = np.linspace(0.0, 1.0, n)
print('%-6s %-15s %10s %10s %10s' % ('seed', 'case', 'default', 'td', 'scipy c64'))
for seed in (5150, 1, 2, 3, 4, 5, 6, 7):
    rng = np.random.default_rng(seed)
    noise = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)
    for name, dc in (('transient 1000', 1000.0 * opening), ('drift 400', 400.0 * ramp), ('transient 3000', 3000.0 * opening),
                     ('drift 1200', 1200.0 * ramp)):
        x = (noise + dc * np.exp(0.54j)).astype(np.complex64)
        _, ref = R.welch_np(x, nperseg=N, nfft=N)
        c64 = relerr(R.welch_c64(x, nperseg=N, nfft=N), ref)
        errs = []
        for force in (None, 'td'):
            plan = ctx.welch_plan(N, window=windows.get_window('hann', N), kernel=_hip.KERNEL_TUNED)
            plan.set_tuning(force)
            plan.set_schedule(_hip.SCHED_CONTIGUOUS)
            errs.append(relerr(plan.exec(x), ref))
            plan.close()
        print('%-6d %-15s %10.2e %10.2e %10.2e' % (seed, name, errs[0], errs[1], c64), flush=True)
