#!/usr/bin/env python3
"""Where a step of the two-channel kernel goes, per wave role: needs a library built with -DOTH_CSDWS_DIAG=1
(make BUILD=build_diag LIB=lib/libofdmtools_hip_diag.so CXXFLAGS="... -DOTH_CSDWS_DIAG=1"), selected with
OFDM_TOOLS_HIP_LIB.  The stamps (s_memtime, which waits for the wave's outstanding LDS / scalar operations) perturb
the kernel: read the shares, not the absolute time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

n = 1 << 26
ctx = _hip.Context(0)
dx, dy = ctx.alloc(n * 8), ctx.alloc(n * 8)
ctx.synth_iq(dx, n, 1003, ((0.5, 0.1234), (2.0, 0.4071)), 0.1 + 0.05j)
ctx.synth_iq(dy, n, 1004, ((0.5, 0.1234), (1.0, -0.2)), 0.1 + 0.05j)
plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0)
for rep in range(3):
    ctx.set_timing(True)
    ctx.get_timing()
    plan.csd_device_src(dx, dy, n)
    ms, k = ctx.get_timing()
    W = 256                                         # one 1024-thread workgroup per CU
    buf = np.zeros(W * 128, np.uint64)
    fn = ctx.lib.oth__debug_partial_raw
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rc = fn(plan.h, W * 4 * 4096, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, rc
    ph = buf.reshape(W, 16, 8).astype(np.float64)
    prod, cons = ph[:, :8, :], ph[:, 8:, :]
    tp, tc = prod.sum(axis=2).mean(), cons.sum(axis=2).mean()
    sp = prod.mean(axis=(0, 1)) / tp * 100
    sc = cons.mean(axis=(0, 1)) / tc * 100
    print('kernel %.3f ms; cycles per wave: producers %.3g consumers %.3g (100 MHz ticks x clock ratio)' % (ms, tp, tc))
    print('  producer %%: bookkeeping %.1f | prefetch wait %.1f | window+sums+pass 1+twiddles+ex1 writes %.1f | barrier %.1f'
          % (sp[0], sp[1], sp[2], sp[3]))
    print('  consumer %%: barrier %.1f | ex1 reads+pass 2 %.1f | twiddles+ex2 writes %.1f | ex2 reads+pass 3 %.1f | exchange+accumulate %.1f'
          % (sc[0], sc[1], sc[2], sc[3], sc[4]))
