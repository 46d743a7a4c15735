#!/usr/bin/env python3
"""Profiling driver: a few launches of ONE configuration's kernels on device-resident synthetic IQ, nothing
else (run under rocprofv3 --kernel-trace or --pmc; the python program goes directly after `--`).

usage: prof_driver.py <config> [reps] [log2_samples]
  C2        4096-pt Hann Welch, 50 % overlap, one 2^28 stream            (welch4096ws_kernel)
  C3        two-channel csd/coherence, 2 x 2^26                          (csd4096 kernel)
  C4        sweep 8 x 2^25, Hann 4096, shift + trim + dB                 (welch4096ws_kernel, 8 streams)
  C4ref     reference-faithful sweep: flattop nperseg 1024 -> nfft 4096  (welch4096_kernel<NA=4>)
  C5        64 channels x 2^22, 16384-pt rect |X|^2/N^2 mean             (welch16k_kernel)
  w<N>               Hann Welch 50 % at nfft N (256 ... 16384: tuned kernels; any other length: fft_any.hip), 2^27 samples
  chain1024 / chain2048 / chain4096   periodogram chain (BH window, shift, |X|^2, IIR + log), 2^26 samples
Prints the HIP-event average of the timed kernels (the averaging kernel; for the chains the whole push: transform
kernel + cross-team reduction + state kernel) and the algorithmic GB/s.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
from ofdm_tools import _hip, windows  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
TONES = ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071))
DC = complex(os.environ.get('PROF_DC', '0.1+0.05j'))      # PROF_DC=70+35j: a DC line far above the signal (the pilot's slow path)
ctx = _hip.Context(0)
hann = lambda n: windows.get_window('hann', n)      # noqa: E731
bufs = []
# PROF_DETREND=fast: OTH_DETREND_CONSTANT_FAST (the builds without the pilot) wherever a configuration detrends
DETMODE = _hip.DETREND_CONSTANT_FAST if os.environ.get('PROF_DETREND') == 'fast' else _hip.DETREND_CONSTANT


def welch_plan(*a, **k):
    k.setdefault('detrend', DETMODE)
    return ctx.welch_plan(*a, **k)



def dev(nbytes):
    p = ctx.alloc(nbytes)
    bufs.append(p)
    return p


if cfg in ('28', '26'):          # old calling convention: prof_driver.py <log2n> <reps>
    cfg, sys.argv = 'C2', sys.argv[:1] + ['C2', sys.argv[2] if len(sys.argv) > 2 else '3', cfg]
log2n = int(sys.argv[3]) if len(sys.argv) > 3 else None
nbytes = 0
if cfg == 'C2':
    n = (1 << (log2n or 28)) - (2048 if os.environ.get('PROF_EVEN') else 0)      # PROF_EVEN: an even segment count (variant ws2)
    d, o = dev(n * 8), dev(4096 * 4)
    ctx.synth_iq(d, n, 1002, TONES, DC)
    plan = welch_plan(4096, window=hann(4096), fs=1.0)
    run = lambda: plan.exec_dev(d, n, o)      # noqa: E731
    nbytes = 8 * n
elif cfg == 'C3':
    n = 1 << (log2n or 26)
    dx, dy = dev(n * 8), dev(n * 8)
    ctx.synth_iq(dx, n, 1003, TONES, DC)
    ctx.synth_iq(dy, n, 1004, TONES, DC)
    plan = welch_plan(4096, window=hann(4096), fs=1.0)
    # Device outputs, launches back to back - the shape of bench.py::csd_bench.  Round 4 timed the HOST-output call
    # here (a stream synchronisation + four D2H copies between launches): the idle gaps let the part run the kernel
    # ~10 % faster than it sustains (tools/c3_bisect.py, profiles/r05_c3_bisect.txt).  PROF_C3_HOST=1
    # restores that form for the A/B.
    outs = [dev(4096 * 4), dev(4096 * 4), dev(4096 * 8), dev(4096 * 4)]
    if os.environ.get('PROF_C3_HOST') == '1':
        run = lambda: plan.csd_device_src(dx, dy, n)      # noqa: E731
    else:
        run = lambda: plan.csd_exec_dev(dx, dy, n, *outs)      # noqa: E731
    nbytes = 16 * n
elif cfg in ('C4', 'C4ref'):
    S, nrf = 1 << (log2n or 25), 8
    d, o = dev(nrf * S * 8), dev(nrf * 3584 * 4)
    for i in range(nrf):
        ctx.synth_iq(d + i * S * 8, S, 2000 + i, TONES, DC)
    kw = dict(window=hann(4096)) if cfg == 'C4' else dict(nperseg=1024, window=windows.get_window('flattop', 1024))
    plan = welch_plan(4096, fs=2.0e6, fftshift=True, trim_bins=256, db=True, **kw)
    run = lambda: plan.exec_dev(d, S, o, nstreams=nrf, stream_stride=S)      # noqa: E731
    nbytes = 8 * nrf * S
elif cfg == 'C5':
    nch, S, N = 64, 1 << (log2n or 22), 16384
    stride = S + int(os.environ.get('PROF_STRIDE_PAD', '0'))      # samples between channel streams (default: back to back)
    d, o = dev(nch * stride * 8), dev(nch * N * 4)
    for i in range(nch):
        ctx.synth_iq(d + i * stride * 8, S, 3000 + i, TONES, DC)
    plan = welch_plan(N, noverlap=0, window=None, detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_OVER_N2,
                          fftshift=True)
    run = lambda: plan.exec_dev(d, S, o, nstreams=nch, stream_stride=stride)      # noqa: E731
    nbytes = 8 * nch * S
elif cfg == 'scan8192':      # the scanner's vectors at fft_len 8192 (multichannel_scanner.py:78-86): 64 channel streams, rect |X|^2/N^2 mean
    nch, S, N = 64, 1 << (log2n or 22), 8192
    d, o = dev(nch * S * 8), dev(nch * N * 4)
    for i in range(nch):
        ctx.synth_iq(d + i * S * 8, S, 3000 + i, TONES, DC)
    plan = welch_plan(N, noverlap=0, window=None, detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_OVER_N2, fftshift=True)
    if os.environ.get('PROF_VARIANT'):
        plan.set_tuning(os.environ['PROF_VARIANT'])
    run = lambda: plan.exec_dev(d, S, o, nstreams=nch, stream_stride=S)      # noqa: E731
    nbytes = 8 * nch * S
elif cfg == 'C5d':      # config 5 as bench.py's scan_c5 runs it: PSD rows + the device decision stage (device outputs)
    from ofdm_tools import scan_batch
    nch, S, N = 64, 1 << (log2n or 22), 16384
    d, o = dev(nch * S * 8), dev(nch * N * 4)
    for i in range(nch):
        ctx.synth_iq(d + i * S * 8, S, 3000 + i, TONES, DC)
    bp = scan_batch.BatchScanPlan(ctx, N, 1000000, 15625.0, 10e3, thr_leveler=10)
    lo, hi = bp._slices()
    noise, power, mask = dev(nch * 4), dev(nch * max(len(lo), 1) * 4), dev(nch * N)

    def run():
        bp.psd_rows_dev(d, S, nch, S, o)
        ctx.scan_decide_dev_out(o, nch, N, bp.scanner.srch_bins, bp.thr_leveler, lo, hi, noise, power, mask)
    nbytes = 8 * nch * S
elif cfg[0] == 'w' and cfg[1:].isdigit():      # w<N>: Hann Welch, 50 % overlap, any length (w1000, w32768, w65536: fft_any.hip)
    N = int(cfg[1:])
    n = 1 << (log2n or 27)
    d, o = dev(n * 8), dev(N * 4)
    ctx.synth_iq(d, n, 1002, TONES, DC)
    plan = welch_plan(N, window=hann(N), fs=1.0)
    if os.environ.get('PROF_VARIANT'):      # e.g. 16k4: the 4 x 4096 / 2 x 4096 kernels of rounds 1-3
        plan.set_tuning(os.environ['PROF_VARIANT'])
    run = lambda: plan.exec_dev(d, n, o)      # noqa: E731
    nbytes = 8 * n
elif cfg in ('p1024', 'p2048', 'p8192', 'p16384'):      # the sweeper's call at these sizes: flattop, nperseg = nfft / 4 zero-padded, 50 % overlap
    N = int(cfg[1:])
    n = 1 << (log2n or 27)
    d, o = dev(n * 8), dev(N * 4)
    ctx.synth_iq(d, n, 1002, TONES, DC)
    plan = welch_plan(N, nperseg=N // 4, window=windows.get_window('flattop', N // 4), fs=2.0e6, fftshift=True, db=True)
    run = lambda: plan.exec_dev(d, n, o)      # noqa: E731
    nbytes = 8 * n
elif cfg.startswith('chain'):
    rect = cfg.startswith('chainr')          # chainrN: rectangular window (ascii_plot's chain), chainN: Blackman-Harris
    N = int(cfg[6:] if rect else cfg[5:])
    n = 1 << (log2n or 26)
    d = dev(n * 8)
    ctx.synth_iq(d, n, 1001, TONES, DC)
    ch = ctx.chain(N, None if rect else windows.blackmanharris(N), True, _hip.EPI_MAG2, 1)
    ch.set_iir_log(0.8, -10.0)
    run = lambda: ch.push_dev(d, n)      # noqa: E731
    nbytes = 8 * n
else:
    sys.exit('unknown config ' + cfg)

import time  # noqa: E402
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:      # clock ramp: an idle MI355X sits at its lowest sclk level
    for _ in range(4):
        run()
    ctx.sync()
hold = float(os.environ.get('PROF_LOOP_SECS', '0'))      # tools/power_probe.sh: keep the kernel running while rocm-smi samples
t0 = time.perf_counter()
while time.perf_counter() - t0 < hold:
    for _ in range(50):
        run()
    ctx.sync()
ctx.set_timing(True)
ctx.get_timing()
for _ in range(reps):
    run()
ms, k = ctx.get_timing()
ctx.set_timing(False)
try:
    print('recipe: ' + plan.last_recipe())
except NameError:
    pass
per_call = ms / reps
print('%s: timed kernels %.4f ms per call (%d timed scopes / %d calls; chain*: transform + cross-team reduction + state) -> %.1f GB/s algorithmic = %.1f %% of 8 TB/s'
      % (cfg, per_call, k, reps, nbytes / per_call / 1e6, nbytes / per_call / 1e6 / 80.0))
if cfg == 'C2':
    probe = ctx.stream_read_probe(bufs[0], nbytes, 2)
    print('read probe %.1f GB/s' % (nbytes / probe / 1e6))
for p in bufs:
    ctx.free(p)
