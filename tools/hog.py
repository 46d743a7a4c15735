#!/usr/bin/env python3
"""usage: tools/hog.py [seconds] [GiB]   - streams reads over a buffer on its own context until the time is up: a
noisy neighbour for timing-dependence experiments (the GPU suite run next to it must not change a single verdict)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nbytes = int(float(sys.argv[2]) * (1 << 30)) if len(sys.argv) > 2 else 8 << 30
ctx = _hip.Context(0)
buf = ctx.alloc(nbytes)
t0, n = time.time(), 0
while time.time() - t0 < seconds:
    ctx.stream_read_probe(buf, nbytes, 8)
    n += 8
print('hog: %d passes over %.0f GiB in %.0f s' % (n, nbytes / (1 << 30), time.time() - t0))
