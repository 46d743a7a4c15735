#!/usr/bin/env python3
"""Per-workgroup start/end stamps of the diagnostic welch4096 build: how even is the finish time?"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

n = 1 << 28
ctx = _hip.Context(0)
d_in = ctx.alloc(n * 8)
d_out = ctx.alloc(4096 * 4)
ctx.synth_iq(d_in, n, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0, kernel=_hip.KERNEL_TUNED)
os.environ['OTH_W4096_VARIANT'] = 'diag'
for rep in range(3):
    plan.exec_dev(d_in, n, d_out)
    ctx.sync()
    buf = np.zeros(1024 * (4 + 48), np.uint64)
    nwg = C.c_int()
    fn = ctx.lib.oth__debug_stamps
    fn.restype = C.c_int
    rc = fn(plan.h, buf.ctypes.data_as(C.c_void_p), 1024, C.byref(nwg))
    assert rc == 0 and nwg.value == 1024
    b = buf[:1024 * 4].reshape(1024, 4).astype(np.int64)
    ph = buf[1024 * 4:].reshape(1024, 4, 12).astype(np.float64)
    tot = ph.sum(axis=2, keepdims=True)
    share = (ph / tot).mean(axis=(0, 1)) * 100
    print('   phase shares %%: prefetch-wait + copy + loads + sums %.1f | wait A %.1f | pass1+ex1 %.1f | wait B %.1f | ex1 read wait %.1f | dft2+tw2+ex2 write %.1f | ex2 landed %.1f | dft3+acc %.1f | loop/chunk %.1f | prefetch wait %.1f  '
          '(cycles per wave %.3g)' % (share[0], share[1], share[2], share[3], share[7], share[8], share[9], share[4], share[5], share[6], tot.mean()))
    t0 = b[:, 0].min()
    start = (b[:, 0] - t0) / 100.0   # us (100 MHz)
    end = (b[:, 1] - t0) / 100.0
    dur = end - start
    xcc = b[:, 2] & 15
    print('rep %d: %d WGs, kernel span %.1f us; start spread %.1f us; duration min/med/max %.1f/%.1f/%.1f us; '
          'end min/med/max %.1f/%.1f/%.1f' % (rep, nwg.value, end.max(), start.max(), dur.min(), np.median(dur),
                                               dur.max(), end.min(), np.median(end), end.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print('   xcc %d: %4d WGs  dur med %.1f  max %.1f  end max %.1f' % (x, m.sum(), np.median(dur[m]),
                                                                              dur[m].max(), end[m].max()))
    print('   mean wave life / span = %.3f' % (dur.mean() / end.max()))
