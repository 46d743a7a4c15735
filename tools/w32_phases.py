#!/usr/bin/env python3
"""Phase shares of welch32k_kernel from a -DW32_DIAG=1 build (per-wave s_memtime stamps behind the partial rows):
  make -C gr-ofdm_tools_amd EXP=1 EXTRA=-DW32_DIAG=1
  OFDM_TOOLS_HIP_LIB=gr-ofdm_tools_amd/lib/libofdmtools_hip_exp.so python3 tools/w32_phases.py [nfft]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S = 1 << 27
ctx = _hip.Context(0)
d, o = ctx.alloc(S * 8), ctx.alloc(N * 4)
ctx.synth_iq(d, S, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
fn = ctx.lib.oth__debug_partial_raw
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
names = ['loads', 'sums + mean barrier', 'window + radix 2', 'pass 1 a', 'exch A + barrier a', 'rest of a + pass 1 b',
         'handover + exch A + barrier b', 'rest of b']
for detrend in (_hip.DETREND_CONSTANT, _hip.DETREND_NONE):
    plan = ctx.welch_plan(N, detrend=detrend)
    for _ in range(30):
        plan.exec_dev(d, S, o)
    ctx.sync()
    G = 256
    W = G if N == 32768 else G // 2
    nseg = (S - N // 2) // (N // 2)
    for rep in range(2):
        plan.exec_dev(d, S, o)
        ctx.sync()
        buf = np.zeros(G * 128, np.uint64)
        rc = fn(plan.h, W * N, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        assert rc == 0, rc
        ph = buf.reshape(G, 16, 8).astype(np.float64)
        tot = ph.sum(axis=2)
        per = nseg / W
        print('%s detrend %d: %d workgroups x 16 waves; ticks per wave: mean %.4g (min %.4g max %.4g); per segment (%.1f per WG) %.0f'
              % (plan.last_recipe().split()[0], detrend, G, tot.mean(), tot.min(), tot.max(), per, tot.mean() / per))
        sh = ph.mean(axis=(0, 1)) / tot.mean() * 100
        print('  ' + ' | '.join('%s %.1f%%' % (n, v) for n, v in zip(names, sh)))
        byw = ph.mean(axis=0) / per
        for r in range(4):
            row = byw[4 * r:4 * r + 4].mean(axis=0)
            print('   waves %2d-%2d ' % (4 * r, 4 * r + 3) + ' '.join('%6.0f' % v for v in row) + ' | %6.0f' % row.sum())
    plan.close()
