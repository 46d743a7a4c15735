#!/usr/bin/env python3
"""Phase shares of welch16k1x_kernel (BASELINE config 5) from a -DOTH_X1_DIAG=1 build:
  make -C gr-ofdm_tools_amd variant TAG=x1diag VSRC=welch16k1x VFLAGS=-DOTH_X1_DIAG=1
  OFDM_TOOLS_HIP_LIB=gr-ofdm_tools_amd/lib/libofdmtools_hip_x1diag.so python3 tools/archive/diag_x1.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip  # noqa: E402

nch, S, N = 64, 1 << 22, 16384
ctx = _hip.Context(0)
d, o = ctx.alloc(nch * S * 8), ctx.alloc(nch * N * 4)
for i in range(nch):
    ctx.synth_iq(d + i * S * 8, S, 3000 + i, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
plan = ctx.welch_plan(N, noverlap=0, window=None, detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_OVER_N2, fftshift=True)
for _ in range(50):
    plan.exec_dev(d, S, o, nstreams=nch, stream_stride=S)
ctx.sync()
fn = ctx.lib.oth__debug_partial_raw
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
pipe = os.environ.get('OTH_16K1X_MODE', '') != 'plain'
names = ['load wait', 'pass 1', 'barrier 1', 'tw + exch A write', 'tail (pass 3, 4, acc) of the segment before', 'barrier 2',
         'loads + A read + pass 2 + B write + B read issue', 'loop'] if pipe else ['load wait', 'pass 1', 'barrier 1', 'tw + exch A write', 'barrier 2', 'A read + pass 2 + tw + B write',
         'B read + pass 3 + tw + pass 4 + acc', 'loop + load issue']
for rep in range(2):
    plan.exec_dev(d, S, o, nstreams=nch, stream_stride=S)
    ctx.sync()
    W = 4
    buf = np.zeros(nch * W * 128, np.uint64)
    rc = fn(plan.h, nch * W * N, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, rc
    ph = buf.reshape(nch * W, 16, 8).astype(np.float64)
    tot = ph.sum(axis=2)
    print('%d workgroups x 16 waves; cycles per wave: mean %.4g (min %.4g max %.4g); per segment (64 per WG) %.0f'
          % (nch * W, tot.mean(), tot.min(), tot.max(), tot.mean() / 64))
    sh = ph.mean(axis=(0, 1)) / tot.mean() * 100
    print('  ' + ' | '.join('%s %.1f%%' % (n, v) for n, v in zip(names, sh)))
    byw = ph.mean(axis=0) / 64
    print('  cycles per segment by wave rank (rows: waves 0-3, 4-7, 8-11, 12-15; columns: the phases above, then total)')
    for r in range(4):
        row = byw[4 * r:4 * r + 4].mean(axis=0)
        print('   ' + ' '.join('%6.0f' % v for v in row) + ' | %6.0f' % row.sum())
    b2 = 5 if pipe else 4
    print('  per wave, cycles per segment [load wait, barrier 1, barrier 2]: ' +
          ' '.join('w%d:%.0f/%.0f/%.0f' % (i, byw[i, 0], byw[i, 2], byw[i, b2]) for i in range(16)))
