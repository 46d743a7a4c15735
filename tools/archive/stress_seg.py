#!/usr/bin/env python3
"""Randomised stress of the 256 ... 2048-point Welch builds (role-split, one-role, zero-padded), the 8192 / 16384 builds,
the two-channel kernels and the fused chain (256 ... 16384) against the coverage kernels on device-resident data.  usage: stress_seg.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
import numpy as np  # noqa: E402
from ofdm_tools import _hip, windows  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _hip.Context(0)
nmax = 1 << 24
d_x, d_y = ctx.alloc(4 * nmax * 8), ctx.alloc(nmax * 8)
d_a, d_b = ctx.alloc(16 * 16384 * 4), ctx.alloc(16 * 16384 * 4)      # up to 16 rows of 16384 floats / 5 x 4096 CSD outputs
ctx.synth_iq(d_x, 4 * nmax, 3, ((0.5, 0.1234), (2.0, 0.4071)), 0.3 - 0.2j)
ctx.synth_iq(d_y, nmax, 4, ((0.7, 0.1234), (1.0, -0.2)), -0.1 + 0.4j)
t0, cases, worst = time.time(), 0, 0.0
while time.time() - t0 < secs:
    kind = rng.choice(['welch', 'welch', 'csd', 'chain'])
    if kind == 'welch':
        nfft = int(rng.choice([256, 512, 1024, 2048, 8192, 16384]))
        # zero-padded segments (the sweeper's nperseg = nfft / 4, and nfft / 2) at 1024 / 2048, nfft / 4 at 8192 / 16384
        nps = nfft // int(rng.choice([1, 1, 2, 4])) if nfft in (1024, 2048) else (nfft // int(rng.choice([1, 1, 4])) if nfft >= 8192 else nfft)
        nov = int(rng.choice([nps // 2, nps // 2, nps // 2, 0, nps // 4, nps - 1]))
        det = int(rng.choice([_hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST, _hip.DETREND_NONE]))
        step = nps - nov
        nseg = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(1, 3000)), int(rng.integers(3000, 30000))]))
        ns = int(rng.integers(1, 5))
        n = min(nps + step * (nseg - 1) + int(rng.integers(0, step)), nmax)
        nseg = (n - nov) // step
        w = windows.get_window(str(rng.choice(['hann', 'flattop', 'blackmanharris'])), nps)
        tuned = ctx.welch_plan(nfft, nperseg=nps, noverlap=nov, window=w, detrend=det, kernel=_hip.KERNEL_TUNED)
        gen = ctx.welch_plan(nfft, nperseg=nps, noverlap=nov, window=w, detrend=det, kernel=_hip.KERNEL_GENERIC)
        build = str(rng.choice(['segws', 'segws', 'seg3'])) if nfft <= 2048 and nps == nfft else ''
        tuned.set_tuning(build or None, sched=int(rng.integers(-1, 3)), chunk=int(rng.choice([0, 1, 2, 3, 7, 16, 33])))
        assert tuned.exec_dev(d_x, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
        assert gen.exec_dev(d_x, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
        a = ctx.d2h(d_a, (ns, nfft), np.float32).astype(np.float64)
        b = ctx.d2h(d_b, (ns, nfft), np.float32).astype(np.float64)
        err = float(np.max(np.abs(a - b) / np.maximum(b, 0.1 * np.median(b))))
        if nseg < 8:      # a handful of segments: single-row rounding of two fp32 transforms (8 ulp of the row's peak amplitude)
            amp = float(np.max(np.abs(np.sqrt(a) - np.sqrt(b)) / np.sqrt(b.max(axis=1, keepdims=True)))) * 2.0 ** 23
            err = min(err, amp * 1e-4 / 8.0)
        info = (kind, nfft, nps, nov, det, nseg, ns, build)
        tuned.close()
        gen.close()
    elif kind == 'csd':
        det = int(rng.choice([_hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT, _hip.DETREND_CONSTANT_FAST, _hip.DETREND_NONE]))
        nseg = int(rng.choice([int(rng.integers(1, 20)), int(rng.integers(1, 3000)), int(rng.integers(3000, 8000))]))
        n = 4096 + 2048 * (nseg - 1) + int(rng.integers(0, 2048))
        w = windows.get_window('hann', 4096)
        tuned = ctx.welch_plan(4096, window=w, detrend=det, kernel=_hip.KERNEL_TUNED)
        gen = ctx.welch_plan(4096, window=w, detrend=det, kernel=_hip.KERNEL_GENERIC)
        build = str(rng.choice(['', '', 'csd1']))
        tuned.set_tuning(build or None, sched=int(rng.integers(-1, 3)), chunk=int(rng.choice([0, 1, 2, 3, 8, 16])))
        res = []
        for plan, o in ((tuned, d_a), (gen, d_b)):
            assert plan.csd_exec_dev(d_x, d_y, n, o, o + 4 * 4096, o + 8 * 4096, o + 16 * 4096) == nseg
            res.append(ctx.d2h(o, (5, 4096), np.float32).astype(np.float64))
        a, b = res
        fl = [np.maximum(b[i], 0.1 * np.median(b[i])) for i in (0, 1)]
        lvl = np.sqrt(fl[0] * fl[1])
        err = max(float(np.max(np.abs(a[0] - b[0]) / fl[0])), float(np.max(np.abs(a[1] - b[1]) / fl[1])),
                  float(np.max(np.abs(a[2:4].reshape(-1, 2) - b[2:4].reshape(-1, 2)) / lvl[:, None])))
        info = (kind, det, nseg, build)
        tuned.close()
        gen.close()
    else:
        nfft = int(rng.choice([256, 512, 1024, 2048, 4096, 8192, 16384]))
        keep = int(rng.integers(1, 5))
        mode = str(rng.choice(['iir', 'peak', 'plain']))
        nrows = int(rng.choice([int(rng.integers(1, 12)), int(rng.integers(1, 600))]))
        n = min(nfft * keep * nrows + int(rng.integers(0, nfft)), nmax)
        give = int(rng.choice([1, 1, 3, 16]))
        shift = bool(rng.integers(2))
        win_bh = bool(rng.integers(2))      # Blackman-Harris, or rectangular (8192 / 16384: the prefetching build)
        cut = int(rng.integers(0, n + 1))
        outs = []
        for kern, o in ((_hip.KERNEL_AUTO, d_a), (_hip.KERNEL_GENERIC, d_b)):
            ch = ctx.chain(nfft, windows.blackmanharris(nfft) if win_bh else None, shift,
                           _hip.EPI_MAG if mode == 'peak' else _hip.EPI_MAG2, keep)
            ch.set_kernel(kern)
            if mode == 'iir':
                ch.set_iir_log(0.2, -3.0)
            elif mode == 'peak':
                ch.set_peak_hold(True)
            got = ch.push_dev(d_x, cut, o, give) if cut else 0
            got2 = ch.push_dev(d_x + 8 * cut, n - cut, o, give) if n - cut else 0
            k = min(got2, give)
            rows = ctx.d2h(o, (max(k, 1), nfft), np.float32).astype(np.float64)[:k]
            state = ch.iir() if mode == 'iir' else (ch.peak() if mode == 'peak' and got + got2 else np.zeros(nfft, np.float32))
            outs.append((got + got2, rows, state.astype(np.float64)))
            ch.close()
        assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
        # single rows of two fp32 transforms: amplitude difference in ulps of the row's peak amplitude (each
        # implementation is within ~1.5 of the float64 result, tools/acc_probe.py; float32 dB adds ~7)
        err = 0.0

        def ulps(a, b, power):
            a, b = (np.sqrt(a), np.sqrt(b)) if power else (a, b)
            return float(np.max(np.abs(a - b) / np.max(b, axis=-1, keepdims=True))) * 2.0 ** 23

        if outs[0][1].size:
            ra, rb = outs[0][1], outs[1][1]
            if mode == 'iir':
                ra, rb = 10 ** (ra / 10), 10 ** (rb / 10)
            err = ulps(ra, rb, mode != 'peak')
        if outs[0][0] and mode != 'plain':
            err = max(err, ulps(outs[0][2], outs[1][2], mode != 'peak'))
        err *= 1e-4 / (32.0 if mode == 'iir' else 8.0)          # normalised so that the common 1e-4 bound applies
        info = (kind, nfft, keep, mode, nrows, give)
    worst = max(worst, err)
    assert err < 1e-4, (info, err)
    cases += 1
    if cases % 100 == 0:
        print('%d cases, worst %.2e' % (cases, worst), flush=True)
print('done: %d cases in %.0f s, worst deviation %.2e' % (cases, time.time() - t0, worst))
