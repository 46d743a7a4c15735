#!/usr/bin/env python3
"""Phase shares of the wave-specialised welch4096 kernel (a build with -DOTH_WS_DIAG=1, e.g. tag wsx1 of
`make EXP=1 WSFLAGS1=-DOTH_WS_DIAG=1`).  usage: diag_ws.py [variant] [sched] [chunk]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

os.environ['OTH_W4096_VARIANT'] = sys.argv[1] if len(sys.argv) > 1 else 'wsx1'
os.environ['OTH_W4096_SCHED'] = sys.argv[2] if len(sys.argv) > 2 else '2'
os.environ['OTH_W4096_CHUNK'] = sys.argv[3] if len(sys.argv) > 3 else '8'
n = 1 << 28
ctx = _hip.Context(0)
d_in = ctx.alloc(n * 8)
d_out = ctx.alloc(4096 * 4)
ctx.synth_iq(d_in, n, 1002, ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071)), 0.1 + 0.05j)
plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0, kernel=_hip.KERNEL_TUNED)
for rep in range(50):
    plan.exec_dev(d_in, n, d_out)
ctx.sync()
for rep in range(2):
    plan.exec_dev(d_in, n, d_out)
    ctx.sync()
    nwg = C.c_int()
    fn = ctx.lib.oth__debug_tail
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    buf = np.zeros(512 * 64, np.uint64)
    rc = fn(plan.h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(nwg))
    assert rc == 0, rc
    ph = buf.reshape(-1, 8, 8)[:nwg.value].astype(np.float64)
    A, B = ph[:, :4], ph[:, 4:]
    ta, tb = A.sum(axis=2).mean(), B.sum(axis=2).mean()
    sa = A.mean(axis=(0, 1)) / ta * 100
    sb = B.mean(axis=(0, 1)) / tb * 100
    print('%d WGs, cycles per wave: producer %.3g consumer %.3g' % (nwg.value, ta, tb))
    print('  producer %%: data wait %.1f | window+sums+loads %.1f | dft16 %.1f | twiddles+ex1 write %.1f | barrier %.1f | other %.1f'
          % (sa[0], sa[1], sa[2], sa[3], sa[4], sa[5]))
    print('  consumer %%: ex1 read %.1f | dft16 %.1f | tw2+ex2 %.1f | dft16+acc %.1f | barrier %.1f | other %.1f'
          % (sb[0], sb[1], sb[2], sb[3], sb[4], sb[5]))
