#!/usr/bin/env python3
"""Where does the Welch PSD of a stream that opens with a 3000-sigma DC transient lose accuracy?  Prints the worst bins
of the default plan (pilot formed in the launch), of 'plaunch' (pilot_mean_kernel) and of the time-domain builds against
the float64 oracle (tests/test_hip_parity.py::test_pilot_under_a_transient_and_a_drifting_offset)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

ctx = _hip.Context(0)
N, nseg = 4096, 2047
n = N + (N // 2) * (nseg - 1)
opening = np.zeros(n)
opening[:N] = 3000.0
ramp = np.linspace(0.0, 1200.0, n)
AMP = float(os.environ.get('DIAG_AMP', '3000'))
opening *= AMP / 3000.0
for seed, (name, dc) in [(s_, c_) for s_ in (5150, 1, 2, 3) for c_ in (('transient', opening), ('drift', ramp))]:
    rng = np.random.default_rng(seed)
    noise = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)
    x = (noise + dc * np.exp(0.54j)).astype(np.complex64)
    _, ref = R.welch_np(x, nperseg=N, nfft=N)
    for force in (None, 'td'):
        for sched in (0,):
            plan = ctx.welch_plan(N, window=windows.get_window('hann', N), kernel=_hip.KERNEL_TUNED)
            plan.set_tuning(force, sched)
            got = plan.exec(x).astype(np.float64)
            e = np.abs(got - ref) / ref
            worst = np.argsort(e)[-4:][::-1]
            print('seed %4d %-9s %-8s sched %2d: max %.2e  worst bins %s  (errors %s)  median %.1e' % (
                seed, name, force or 'default', sched, e.max(), list(worst), ['%.1e' % e[k] for k in worst], np.median(e)), flush=True)
            plan.close()
