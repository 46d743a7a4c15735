#!/usr/bin/env python3
"""What a GNU Radio scheduler's chunks cost (round 6, verdict item 5): work() is called with 4 Ki - 32 Ki items
(python/spectrum_sensor.py:71-75, spectrum_sensor_v2.py:85-97), not with the 2^26-sample pushes of profiles/*chain*.

For the five chain blocks and the legacy sensor, pushes of 4096 / 8192 / 32768 items at fft_len 1024 / 4096:
  work() us          time.perf_counter around blk.work() with threaded=False: enqueue AND wait for the row on the same thread
                     (median; ctypes + Python included) - the deterministic hosts' form
  enqueue us         oth_chain_push_async alone (what work() costs the scheduler thread with the default threaded=True: the
                     watcher thread waits for the row)
  ops / push         stream operations the push enqueued (oth_chain_last_push_ops: async copies + launches)
  device us / push   HIP-event time of the push's kernels (oth_ctx_set_timing scopes)
  Msamples/s         sustained: N pushes back to back, then one synchronisation
  watcher us         from the entry of work() to _on_vector on the watcher thread (threaded=True; median / p95)
usage (GPU box): python tools/work_latency.py > gpurun_out/work_latency.txt
"""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import ofdm_tools  # noqa: E402
from ofdm_tools import _hip  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

ctx = _hip.Context(0)
REPS = 400


def blocks(N, threaded):
    Sf = N * 1000
    kw = dict(ctx=ctx, threaded=threaded)
    return [
        ('spectrum_sensor_v2', lambda: ofdm_tools.spectrum_sensor_v2(N, 1000, Sf, channel_space=Sf / 40.0, search_bw=Sf / 80.0,
                                                                       trunc_band=Sf, stats=True, **kw)),
        ('sensor_v2 decim 8', lambda: ofdm_tools.spectrum_sensor_v2(N, 125, Sf, channel_space=Sf / 40.0, search_bw=Sf / 80.0,
                                                                      trunc_band=Sf, stats=True, **kw)),
        ('sensor_v2 decim 100', lambda: ofdm_tools.spectrum_sensor_v2(N, 10, Sf, channel_space=Sf / 40.0, search_bw=Sf / 80.0,
                                                                        trunc_band=Sf, stats=True, **kw)),
        ('multichannel_scanner', lambda: ofdm_tools.multichannel_scanner(N, 1000, Sf, channel_space=Sf / 40.0, search_bw=Sf / 80.0,
                                                                           tune_freq=0, trunc_band=Sf, subject_channels=[0.0], **kw)),
        ('psd_logger', lambda: ofdm_tools.psd_logger(N, 1000, Sf, mat_file=os.devnull, **kw)),
        ('local_worker', lambda: ofdm_tools.local_worker(N, Sf, 0.3, 1000, 1472, True, **kw)),
        ('ascii_plot', lambda: ofdm_tools.ascii_plot(N, Sf, 0.0, 0.3, 1000, 64, 20, **kw)),
    ]


def measure(name, make, items):
    x = R.synth_iq(items, 1)
    blk = make()
    on = getattr(blk, '_on_vector')
    blk._on_vector = lambda row: None          # the watcher's own work (logging, PDUs) is not what is measured here
    for _ in range(20):
        blk.work([x], [])
    ctx.sync()
    host, ops = [], []
    for _ in range(REPS):
        t0 = time.perf_counter()
        blk.work([x], [])
        host.append((time.perf_counter() - t0) * 1e6)
        ops.append(blk._chain.last_push_ops())
        if len(host) % 4 == 0:
            ctx.sync()                          # keep the four-slot ring from filling: a scheduler delivers at the sample rate
    ctx.sync()
    enq = []
    for i in range(REPS):                       # the enqueue alone: what the scheduler thread pays when the watcher thread collects
        t0 = time.perf_counter()
        blk._chain.push_async(x)
        enq.append((time.perf_counter() - t0) * 1e6)
        if i % 3 == 2:
            ctx.sync()
    ctx.sync()
    ctx.set_timing(True)
    ctx.get_timing()
    for _ in range(50):
        blk.work([x], [])
    dev_ms, scopes = ctx.get_timing()
    ctx.set_timing(False)
    t0 = time.perf_counter()
    for _ in range(REPS):
        blk.work([x], [])
    ctx.sync()
    rate = REPS * items / (time.perf_counter() - t0) / 1e6
    blk._on_vector = on
    blk.stop()
    return statistics.median(host), statistics.median(enq), statistics.mean(ops), max(ops), 1e3 * dev_ms / 50, rate


def watcher_latency(N, items):
    Sf = N * 1000
    blk = ofdm_tools.spectrum_sensor_v2(N, 1000, Sf, channel_space=Sf / 40.0, search_bw=Sf / 80.0, trunc_band=Sf, stats=True,
                                        ctx=ctx, threaded=True)
    x = R.synth_iq(items, 2)
    stamps = []
    t_call = [0.0]
    blk._on_vector = lambda row: stamps.append((time.perf_counter() - t_call[0]) * 1e6)
    for _ in range(200):
        t_call[0] = time.perf_counter()
        blk.work([x], [])
        time.sleep(0.0005)                      # a scheduler's pace: the watcher keeps up, every vector is seen
    blk.drain()
    blk.stop()
    stamps.sort()
    return statistics.median(stamps), stamps[int(0.95 * len(stamps))], len(stamps)


print('device: %s' % ctx.device_name())
print('%-22s %6s %7s | %10s %10s %9s %9s %11s' % ('block', 'fft', 'items', 'work() us', 'enqueue us', 'ops/push', 'dev us', 'Msamples/s'))
for N in (1024, 4096):
    for items in (4096, 8192, 32768):
        for name, make in blocks(N, False):
            h, e, om, ox, d, r = measure(name, make, items)
            print('%-22s %6d %7d | %10.1f %10.1f %5.2f (%d) %9.1f %11.1f' % (name, N, items, h, e, om, ox, d, r), flush=True)
# the legacy sensor: work() only stores samples (python/spectrum_sensor.py:71-75); the scan is a message handler
for items in (4096, 8192, 32768):
    blk = ofdm_tools.spectrum_sensor(items, sample_rate=1000000, fft_len=1024, channel_space=25e3, search_bw=12.5e3, ctx=ctx)
    x = R.synth_iq(items, 3)
    host = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        blk.work([x], [])
        host.append((time.perf_counter() - t0) * 1e6)
    print('%-22s %6s %7d | %10.1f %10s %9s %9s %11s' % ('spectrum_sensor (legacy)', '-', items, statistics.median(host), '-', '0', '-', '-'))
for N in (1024, 4096):
    for items in (4096, 8192, 32768):
        med, p95, n = watcher_latency(N, items)
        print('watcher latency spectrum_sensor_v2 fft %d items %d: median %.1f us, p95 %.1f us (%d vectors of 200 pushes)' % (N, items, med, p95, n))
