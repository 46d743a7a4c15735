#!/usr/bin/env python3
"""The library's two streaming-read probes (16-byte and 8-byte loads per lane, oth_stream_read_probe) on a 2 GiB
buffer.  usage: probe_widths.py"""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip
ctx = _hip.Context(0)
n = 1 << 28
d = ctx.alloc(n * 8)
ctx.synth_iq(d, n, 1, (), 0j)
for _ in range(3):
    a = ctx.stream_read_probe(d, n * 8, 20)
    b = ctx.stream_read_probe(d, n * 8, -20)
    print('float4 probe %.1f GB/s   float2-nt probe %.1f GB/s' % (n * 8 / a / 1e6, n * 8 / b / 1e6))
