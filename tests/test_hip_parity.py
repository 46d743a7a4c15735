"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the
committed golden vectors and the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): <= 1e-4 relative on linear power over ALL
bins; coherence <= 1e-4 absolute.  dB outputs are compared after conversion back
to linear power.
"""
import os

import numpy as np
import pytest

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.abs(b)))


def check_single_rows(rows, ref, power=True, ulps=4, c64=None, plain=None, floor_rel=None):
    """SINGLE periodogram rows (the per-vector outputs of the GNU Radio chains: |X|^2 rows when `power`, |X| rows
    otherwise).  A single transform of noise + a strong tone has bins down to 1e-6 of the row's peak (Rayleigh
    nulls, leakage skirts); there the amplitude rounding of ANY fp32 FFT (FFTW3f, which fft_vcc runs, included) -
    about one ulp of the row's LARGEST amplitude, because the tone's energy passes through every butterfly - is
    1e-4 or more of the bin's own value.  Every call holds the rows to
      * amplitude error <= `ulps` (4: 2^-21) ulp of the row's peak amplitude on EVERY bin,
      * the plain 1e-4 on every bin at or above the row's median level,
      * mean relative error < 2e-6.
    Where the committed fixture carries them, the round-1 criteria are asserted again and the HIP rows are measured
    against a single-precision CPU FFT of the same samples:
      * `plain`: max relative error over ALL bins < plain (1e-4 where a CPU fp32 FFT meets it too),
      * `floor_rel`: |d| <= 1e-4 max(ref, floor_rel * median(ref)) on all bins (the a1 criterion of round 1),
      * `c64` (scipy.fft on complex64 = FFTW3f-class arithmetic, stored in the fixture): no HIP row's worst amplitude
        error exceeds 1.5 x the worst error of the CPU fp32 rows of the fixture (measured 1.05 x on a1, 1.00 x on a2),
        and the mean over rows of the per-row worst error stays within 2 x the CPU's (measured 1.5 x at 1024 points,
        where pocketfft is pure radix 4; 1.0 x at 256 / 512 / 2048 / 4096).  The ratio row by row is NOT asserted: a
        row's worst bin is an extreme value over 1024-4096 roundings, and two CPU fp32 FFTs of the same samples
        (pocketfft against a textbook radix-2 in complex64, tools/acc_rows.py) differ by up to 4.7 x per row while
        their fixture-level worst agree within 12 %.
    Averaged quantities (the 8-row mean, peak hold over many rows, every Welch PSD) keep the plain 1e-4 on every bin."""
    rows = np.atleast_2d(np.asarray(rows, np.float64))
    ref = np.atleast_2d(np.asarray(ref, np.float64))
    assert rows.shape == ref.shape
    amp, amp_ref = (np.sqrt(rows), np.sqrt(ref)) if power else (rows, ref)
    amp_err = np.abs(amp - amp_ref) / amp_ref.max(axis=1, keepdims=True)
    assert amp_err.max() <= ulps * 2.0 ** -23, amp_err.max() * 2.0 ** 23
    upper = ref >= np.median(ref, axis=1, keepdims=True)
    assert np.max(np.abs(rows - ref)[upper] / ref[upper]) < RTOL
    assert np.mean(np.abs(rows - ref) / ref) < 2e-6
    if plain is not None:
        assert np.max(np.abs(rows - ref) / ref) < plain, np.max(np.abs(rows - ref) / ref)
    if floor_rel is not None:
        v = np.max(np.abs(rows - ref) / np.maximum(ref, floor_rel * np.median(ref)))
        assert v < RTOL, v
    if c64 is not None:
        c64 = np.atleast_2d(np.asarray(c64, np.float64))
        camp = np.sqrt(c64) if power else c64
        cpu_err = np.abs(camp - amp_ref) / amp_ref.max(axis=1, keepdims=True)
        eh, ec = amp_err.max(axis=1), cpu_err.max(axis=1)
        assert np.all(eh <= 1.5 * ec.max()), (eh.max() * 2.0 ** 23, ec.max() * 2.0 ** 23)
        assert eh.mean() <= 2.0 * ec.mean(), (eh.mean() * 2.0 ** 23, ec.mean() * 2.0 ** 23)


@pytest.fixture(scope='module')
def hip():
    from ofdm_tools import _hip
    return _hip


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def hann(n):
    from ofdm_tools import windows
    return windows.get_window('hann', n)


def flattop(n):
    from ofdm_tools import windows
    return windows.get_window('flattop', n)


# ---------------------------------------------------------------- Welch ----

@pytest.mark.parametrize('kernel', ['tuned', 'generic'])
def test_welch_hann_4096_golden(ctx, hip, golden, kernel):
    g = golden('welch_hann_4096_50.npz')
    plan = ctx.welch_plan(4096, window=hann(4096), fs=float(g['fs']),
                          kernel=hip.KERNEL_TUNED if kernel == 'tuned' else hip.KERNEL_GENERIC)
    psd = plan.exec(g['x'])
    assert plan.last_nseg == 31
    assert relerr(psd, g['expected_psd']) < RTOL


def test_welch_schedules_agree_and_static_ones_are_bit_reproducible(ctx, hip):
    x = R.synth_iq(3_000_000, 21)          # 1463 segments: more chunks than resident workgroups
    _, ref = R.welch_np(x, nperseg=4096, nfft=4096)
    plan = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_TUNED)
    outs = {}
    for sched in (hip.SCHED_CONTIGUOUS, hip.SCHED_INTERLEAVED, hip.SCHED_DYNAMIC):
        plan.set_schedule(sched)
        outs[sched] = [plan.exec(x) for _ in range(3)]
        assert plan.last_nseg == 1463
        for o in outs[sched]:
            assert relerr(o, ref) < RTOL
    for sched in (hip.SCHED_CONTIGUOUS, hip.SCHED_INTERLEAVED):
        assert np.array_equal(outs[sched][0], outs[sched][1]) and np.array_equal(outs[sched][0], outs[sched][2])
    assert relerr(outs[hip.SCHED_DYNAMIC][0], outs[hip.SCHED_CONTIGUOUS][0]) < 2e-6


def test_welch_ragged_length_and_fs(ctx, golden):
    g = golden('welch_hann_1024_ragged.npz')
    plan = ctx.welch_plan(1024, window=hann(1024), fs=float(g['fs']))
    assert relerr(plan.exec(g['x']), g['expected_psd']) < RTOL
    assert plan.last_nseg == (50000 - 512) // 512


def test_welch_flattop_2048(ctx, golden):
    g = golden('welch_flattop_2048.npz')
    plan = ctx.welch_plan(2048, window=flattop(2048), fs=float(g['fs']))
    assert relerr(plan.exec(g['x']), g['expected_psd']) < RTOL


def test_sweeper_segment_zero_padded_shift_trim_db(ctx, golden):
    g = golden('welch_flattop_nperseg_quarter.npz')
    nfft, ex = int(g['nfft']), int(g['excess_bins'])
    lin = ctx.welch_plan(nfft, nperseg=nfft // 4, window=flattop(nfft // 4), fs=float(g['fs']), fftshift=True,
                         trim_bins=ex).exec(g['x'])
    assert lin.shape == (nfft - 2 * ex,)
    assert relerr(lin, g['expected_psd_lin']) < RTOL
    db = ctx.welch_plan(nfft, nperseg=nfft // 4, window=flattop(nfft // 4), fs=float(g['fs']), fftshift=True,
                        trim_bins=ex, db=True).exec(g['x'])
    assert relerr(10 ** (db.astype(np.float64) / 10), g['expected_psd_lin']) < RTOL
    assert np.max(np.abs(db - g['expected_psd_db'])) < 1e-3


@pytest.mark.parametrize('kernel', ['tuned', 'generic'])
def test_many_segments_shift_trim_db_vs_oracle(ctx, hip, kernel):
    """The sweep segment of the multi-GPU bench at a size the oracle still handles: 1023 segments, i.e. a
    full grid of workgroups and the one-launch cross-workgroup reduction, with fftshift + trim + dB."""
    n = 4096 + 2048 * 1022
    x = np.concatenate([R.synth_iq(1 << 19, 40 + i, n0=i << 19) for i in range(5)])[:n]
    k = hip.KERNEL_TUNED if kernel == 'tuned' else hip.KERNEL_GENERIC
    plan = ctx.welch_plan(4096, window=hann(4096), fs=2.5e6, fftshift=True, trim_bins=256, db=True, kernel=k)
    got = plan.exec(x)
    assert plan.last_nseg == 1023 and got.shape == (3584,)
    _, ref = R.welch_np(x, fs=2.5e6, nperseg=4096, nfft=4096)
    ref = np.fft.fftshift(ref)[256:-256]
    assert relerr(10 ** (got.astype(np.float64) / 10), ref) < RTOL
    plan.close()


@pytest.mark.parametrize('nperseg', [256, 512, 1024, 2048])
def test_welch4096_zero_padded_segments_tuned_and_generic(ctx, hip, nperseg):
    x = R.synth_iq(60000, 300 + nperseg)
    for nov in (nperseg // 2, 0, nperseg - 1):
        _, ref = R.welch_np(x, fs=5.0, window='flattop', nperseg=nperseg, noverlap=nov, nfft=4096)
        for kern in (hip.KERNEL_TUNED, hip.KERNEL_GENERIC):
            plan = ctx.welch_plan(4096, nperseg=nperseg, noverlap=nov, window=flattop(nperseg), fs=5.0, kernel=kern)
            assert relerr(plan.exec(x), ref) < RTOL
            plan.close()


@pytest.mark.parametrize('nfft', [64, 128, 256, 512, 8192, 16384])
def test_welch_all_sizes_vs_oracle(ctx, nfft):
    x = R.synth_iq(max(8 * nfft, 16384) + 37, 100 + nfft)
    _, ref = R.welch_np(x, fs=3.0, nperseg=nfft, nfft=nfft)
    psd = ctx.welch_plan(nfft, window=hann(nfft), fs=3.0).exec(x)
    assert relerr(psd, ref) < RTOL


@pytest.mark.parametrize('N', [8192, 16384])
@pytest.mark.parametrize('kernel', ['tuned', 'generic'])
def test_welch16k_scanner_config_batched(ctx, hip, kernel, N):
    """BASELINE config 5 shape: channel streams x 16384-point (and the 8192-point build of the same kernel), rect
    window, no overlap, |X|^2/N^2 mean."""
    ns, n = 6, N * 9 + 100
    xs = [R.synth_iq(n, 3000 + i) for i in range(ns)]
    buf = np.concatenate(xs)
    d_in, d_out = ctx.alloc(buf.nbytes), ctx.alloc(ns * N * 4)
    try:
        ctx.h2d(d_in, buf)
        plan = ctx.welch_plan(N, noverlap=0, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_OVER_N2,
                              fftshift=True, kernel=hip.KERNEL_TUNED if kernel == 'tuned' else hip.KERNEL_GENERIC)
        assert plan.exec_dev(d_in, n, d_out, nstreams=ns, stream_stride=n) == 9
        out = ctx.d2h(d_out, (ns, N), np.float32)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    for i, x in enumerate(xs):
        ref = R.chain_sensor_v2(x, N).mean(axis=0)
        assert relerr(out[i], ref) < RTOL


def test_welch16k1x_scanner_kernels_schedules_and_counts(ctx, hip):
    """16384-point vectors that do not overlap, no detrend (BASELINE config 5, multichannel_scanner.py:78-86 averaged):
    the one-exchange kernels (pipelined default, 'plain', windowed) and the 4 x 4096 build ('16k4') against the float64
    oracle on a short run, and against the coverage kernel over segment counts around chunk and grid multiples (odd
    counts end in a one-segment chunk), 1-3 streams, all three schedules, chunk sizes 0-5, steps >= nfft."""
    N = 16384
    rng = np.random.default_rng(5)
    x = R.synth_iq(N * 7 + 33, 77)
    ref = R.chain_sensor_v2(x, N).mean(axis=0)                    # rect, shift, |X|^2 / N^2, mean over 7 vectors
    for variant in (None, '16kplain', '16k4'):
        plan = ctx.welch_plan(N, noverlap=0, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_OVER_N2,
                              fftshift=True, kernel=hip.KERNEL_TUNED)
        plan.set_tuning(variant)
        assert relerr(plan.exec(x), ref) < RTOL and plan.last_nseg == 7, variant
        plan.close()
    _, refw = R.welch_np(x, fs=2.0, window='hann', nperseg=N, noverlap=0, nfft=N, detrend=False)
    plan = ctx.welch_plan(N, noverlap=0, window=hann(N), detrend=hip.DETREND_NONE, fs=2.0, kernel=hip.KERNEL_TUNED)
    assert relerr(plan.exec(x), refw) < RTOL                        # the windowed build
    plan.close()
    nmax = N * 300 + 5000
    d_in, d_a, d_b = ctx.alloc(3 * nmax * 8), ctx.alloc(3 * N * 4), ctx.alloc(3 * N * 4)
    try:
        ctx.synth_iq(d_in, 3 * nmax, 78, R.TONES, R.DC)
        kw = dict(noverlap=0, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_OVER_N2, fftshift=True)
        tuned = ctx.welch_plan(N, kernel=hip.KERNEL_TUNED, **kw)
        gen = ctx.welch_plan(N, kernel=hip.KERNEL_GENERIC, **kw)
        for nseg in [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 299] + [int(v) for v in rng.integers(1, 300, 5)]:
            n = N * nseg + int(rng.integers(0, N))
            ns = int(rng.integers(1, 4))
            assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
            b = ctx.d2h(d_b, (ns, N), np.float32)
            for variant in (None, '16kplain'):
                sched, chunk = int(rng.integers(0, 3)), int(rng.integers(0, 6))
                tuned.set_schedule(sched)
                tuned.set_tuning(variant, chunk=chunk)
                assert tuned.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
                a = ctx.d2h(d_a, (ns, N), np.float32).astype(np.float64)
                err = np.max(np.abs(a - b) / np.maximum(b, 0.1 * np.median(b)))
                assert err < (5e-5 if nseg >= 8 else 2e-4), (nseg, ns, variant, sched, chunk, err)
        tuned.close()
        gen.close()
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


@pytest.mark.parametrize('N', [16384, 8192])
def test_welch16k1x_kernels_repeat_without_drift(ctx, hip, N):
    """200 launches of the one-exchange kernels (16384 points on 16 waves; 8192 points on 8, round 5) on one input - the
    pipelined scanner kernel, the plain one (16384 only), the 50 %-overlap form with its prefetched half - schedules and
    chunk sizes drawn at random, two streams: every result must stay within rounding of the first (a missed barrier, a
    stale ticket or a load that is read before it has landed shows up as an occasional outlier, not as a steady error)."""
    rng = np.random.default_rng(123)
    n = N * 131 + 777
    d_in = ctx.alloc(2 * n * 8)
    d_a, d_b = ctx.alloc(2 * N * 4), ctx.alloc(2 * N * 4)
    try:
        ctx.synth_iq(d_in, 2 * n, 9, R.TONES, R.DC)
        scan = ctx.welch_plan(N, noverlap=0, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_OVER_N2,
                              kernel=hip.KERNEL_TUNED)
        half = ctx.welch_plan(N, window=hann(N), kernel=hip.KERNEL_TUNED)
        gens = {}
        for name, plan, kw in (('scan', scan, dict(noverlap=0, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_OVER_N2)),
                               ('half', half, dict(window=hann(N)))):
            g = ctx.welch_plan(N, kernel=hip.KERNEL_GENERIC, **kw)
            g.exec_dev(d_in, n, d_b, nstreams=2, stream_stride=n)
            gens[name] = ctx.d2h(d_b, (2, N), np.float32).astype(np.float64)
            g.close()
        worst = 0.0
        for it in range(200):
            name, plan = (('scan', scan), ('half', half))[it % 2]
            variant = None if (name == 'half' or N == 8192) else (None, '16kplain')[(it // 2) % 2]
            plan.set_tuning(variant, chunk=int(rng.choice([0, 2, 3, 5, 8, 16])))
            plan.set_schedule(int(rng.integers(0, 3)))
            nseg = plan.exec_dev(d_in, n, d_a, nstreams=2, stream_stride=n)
            assert nseg == (131 if name == 'scan' else 261)
            got = ctx.d2h(d_a, (2, N), np.float32)
            err = float(np.max(np.abs(got - gens[name]) / gens[name]))
            worst = max(worst, err)
            assert err < 2e-5, (it, name, variant, err)
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


@pytest.mark.parametrize('N', [8192, 16384])
def test_welch16k_hann_overlap_detrend_many_segments(ctx, hip, N):
    nseg = 701 if N == 16384 else 1403                # at 75 % overlap: more segments than resident workgroups
    x = R.synth_iq(N + (N // 4) * (nseg - 1), 16)
    _, ref = R.welch_np(x, fs=4.0, nperseg=N, noverlap=3 * N // 4, nfft=N)
    for sched in (hip.SCHED_DYNAMIC, hip.SCHED_CONTIGUOUS, hip.SCHED_INTERLEAVED):
        plan = ctx.welch_plan(N, noverlap=3 * N // 4, window=hann(N), fs=4.0, kernel=hip.KERNEL_TUNED)
        plan.set_schedule(sched)
        assert relerr(plan.exec(x), ref) < RTOL and plan.last_nseg == nseg
    # 50 % overlap, DC offset 30x the noise, no detrend / detrend
    rng = np.random.default_rng(N)
    xdc = (rng.standard_normal(N * 12) + 1j * rng.standard_normal(N * 12) + (30.0 - 18.0j)).astype(np.complex64)
    for det in (True, False):
        _, ref = R.welch_np(xdc, nperseg=N, nfft=N, detrend='constant' if det else False)
        plan = ctx.welch_plan(N, window=hann(N), detrend=hip.DETREND_CONSTANT if det else hip.DETREND_NONE,
                              kernel=hip.KERNEL_TUNED)
        assert relerr(plan.exec(xdc), ref) < RTOL
    # the detrend at 50 % overlap runs in the frequency domain when the window's spectrum is confined (periodic
    # cosine-sum windows: Hann above, flattop here), in the time domain otherwise (a symmetric Hamming); both keep the
    # overlapped half in registers at 8192, the 16384-point build only with the frequency-domain form
    import scipy.signal as sg
    for wname, w in (('flattop', flattop(N)), ('hamming_sym', sg.windows.hamming(N, sym=True).astype(np.float32))):
        _, ref = R.welch_np(xdc, window=w.astype(np.float64), nperseg=N, nfft=N)
        # |m| = 35 sigma over 23 segments.  The default plan (OTH_DETREND_CONSTANT: computed on x - pilot) is held to the
        # flat 1e-4 on EVERY bin, the bins under the removed DC line included (round 4 still allowed them the float32-mean
        # bound here); that bound belongs to OTH_DETREND_CONSTANT_FAST alone, which works on the raw samples.
        plan = ctx.welch_plan(N, window=w, kernel=hip.KERNEL_TUNED)
        e = np.abs(plan.exec(xdc) - ref) / ref
        assert e.max() < RTOL, (wname, e.max(), int(np.argmax(e)))
        plan.close()
        fast = ctx.welch_plan(N, window=w, kernel=hip.KERNEL_TUNED, detrend=hip.DETREND_CONSTANT_FAST)
        e = np.abs(fast.exec(xdc) - ref) / ref
        fast.close()
        lobe = [0, 1, 2, 3, 4, N - 4, N - 3, N - 2, N - 1]      # the window's main lobe: |k| <= 4 for a flat-top
        Wk = np.abs(np.fft.fft(w.astype(np.float64)))[lobe]
        bound = 2 * (2 * 2.0 ** -23 * abs(30.0 - 18.0j)) * Wk / np.sqrt(ref[lobe] * float(np.sum(w.astype(np.float64) ** 2)))
        assert np.delete(e, lobe).max() < RTOL and np.all(e[lobe] <= np.maximum(RTOL, bound)), (wname, 'fast', e.max())
    # many segments, device-resident, against the coverage kernel: counts around chunk and grid multiples, 1-3 streams
    step = N // 2
    nmax = N + step * 2100
    d_in, d_a, d_b = ctx.alloc(3 * nmax * 8), ctx.alloc(3 * N * 4), ctx.alloc(3 * N * 4)
    try:
        ctx.synth_iq(d_in, 3 * nmax, 77, R.TONES, 3.0 - 2.0j)
        tuned = ctx.welch_plan(N, window=hann(N), kernel=hip.KERNEL_TUNED)
        gen = ctx.welch_plan(N, window=hann(N), kernel=hip.KERNEL_GENERIC)
        for nseg in [1, 2, 3, 7, 8, 9, 255, 256, 257, 511, 513, 2047, 2100] + [int(v) for v in rng.integers(1, 2100, 4)]:
            n = N + step * (nseg - 1) + int(rng.integers(0, step))
            ns = int(rng.integers(1, 4))
            assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
            b = ctx.d2h(d_b, (ns, N), np.float32)
            # the plan's own choice (fewer than 8 segments per stream: time-domain detrend, as the coverage kernel), then
            # the frequency-domain build forced at every count ('fd': any tag that is not a 4096 variant keeps the size's
            # default kernel and switches the few-segment routing off)
            for force in (None, 'fd'):
                tuned.set_schedule(int(rng.integers(0, 3)))
                tuned.set_tuning(force, chunk=int(rng.integers(0, 6)))
                assert tuned.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
                a = ctx.d2h(d_a, (ns, N), np.float32).astype(np.float64)
                err = np.max(np.abs(a - b) / np.maximum(b, 0.1 * np.median(b)))
                if nseg >= 8:
                    assert err < 5e-5, (nseg, ns, force, err)
                else:
                    # a handful of segments: the comparison is between two fp32 transforms of (nearly) single
                    # periodogram rows, whose low bins differ by the rows' own rounding - held to ulps of the row's peak
                    # amplitude, like single rows elsewhere.  With the plan's own choice both kernels detrend in the
                    # time domain, each with its own order of adds for the mean (4 ulp); the FORCED frequency-domain form also carries, in bins 0 and +-1, the
                    # rounding of the DC line (3.6 x the noise here) the transform saw - what the routing avoids (8 ulp).
                    # Parity against the float64 oracle: test_detrend_forms_few_segments_and_large_dc.
                    amp = np.abs(np.sqrt(a) - np.sqrt(b)) / np.sqrt(b.max(axis=1, keepdims=True))
                    ulp = amp.max() * 2.0 ** 23
                    print('welch16k N=%d nseg=%d streams=%d %s: %.2f ulp of the peak, rel %.1e' %
                          (N, nseg, ns, force or 'auto', ulp, err))
                    assert ulp <= (8 if force else 4) and err < 1e-3, (nseg, ns, force, err, ulp)
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


def test_welch_no_detrend_rect_raw_scaling(ctx, hip):
    x = R.synth_iq(20000, 9)
    _, ref = R.welch_np(x, window='boxcar', nperseg=4096, noverlap=1000, nfft=4096, detrend=False, scaling='none')
    for kern in (hip.KERNEL_TUNED, hip.KERNEL_GENERIC):
        plan = ctx.welch_plan(4096, noverlap=1000, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_RAW,
                              kernel=kern)
        assert relerr(plan.exec(x), ref) < RTOL


def test_welch_exactly_one_segment_and_too_short(ctx, hip):
    x = R.synth_iq(4096, 3)
    _, ref = R.welch_np(x, nperseg=4096, nfft=4096)
    plan = ctx.welch_plan(4096, window=hann(4096))
    assert relerr(plan.exec(x), ref) < RTOL and plan.last_nseg == 1
    with pytest.raises(hip.HipError) as ei:
        plan.exec(x[:4095])
    assert ei.value.code == -1
    with pytest.raises(hip.HipError):
        plan.exec(np.zeros(0, np.complex64))


def test_welch_bad_plans_are_rejected(ctx, hip):
    # (round 6: nfft = 1000, 32, 32768 are plans now - tests/test_anylen_gpu.py; what stays refused is a length the
    # any-length engine cannot reach, and the argument errors)
    for kw in (dict(nfft=0), dict(nfft=(1 << 21)), dict(nfft=600001), dict(nfft=1024, nperseg=2048),
               dict(nfft=1024, noverlap=1024), dict(nfft=1024, trim_bins=512), dict(nfft=1024, fs=0.0)):
        with pytest.raises(hip.HipError):
            ctx.welch_plan(**kw)
    with pytest.raises(hip.HipError):       # tuned kernel forced on a size it does not cover
        ctx.welch_plan(128, kernel=hip.KERNEL_TUNED).exec(R.synth_iq(4096, 1))


def test_welch_streaming_chunks_equal_one_shot(ctx):
    x = R.synth_iq(100000, 12)
    plan = ctx.welch_plan(4096, window=hann(4096), fs=2.0)
    one = plan.exec(x)
    nseg = plan.last_nseg
    rng = np.random.default_rng(0)
    pos = 0
    while pos < len(x):       # ragged chunk sizes incl. chunks shorter than a segment
        n = int(rng.choice([1, 100, 2047, 2048, 4096, 5000, 30000]))
        plan.accumulate(x[pos:pos + n])
        pos += n
    out = plan.finalize()
    assert plan.last_nseg == nseg
    assert relerr(out, one) < 2e-6
    # finalize() resets: a second stream gives its own result
    plan.accumulate(x[:50000])
    _, ref = R.welch_np(x[:50000], fs=2.0, nperseg=4096, nfft=4096)
    assert relerr(plan.finalize(), ref) < RTOL


def test_welch_finalize_without_data_is_a_state_error(ctx, hip):
    plan = ctx.welch_plan(1024)
    with pytest.raises(hip.HipError) as ei:
        plan.finalize()
    assert ei.value.code == -5


def test_welch_batched_streams_device_resident(ctx):
    ns, n, nfft = 5, 40000, 4096
    xs = [R.synth_iq(n, 3000 + i) for i in range(ns)]
    stride = n + 123
    buf = np.zeros(ns * stride, np.complex64)
    for i, x in enumerate(xs):
        buf[i * stride:i * stride + n] = x
    d_in = ctx.alloc(buf.nbytes)
    d_out = ctx.alloc(ns * nfft * 4)
    try:
        ctx.h2d(d_in, buf)
        plan = ctx.welch_plan(nfft, window=hann(nfft))
        plan.exec_dev(d_in, n, d_out, nstreams=ns, stream_stride=stride)
        out = ctx.d2h(d_out, (ns, nfft), np.float32)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    for i, x in enumerate(xs):
        _, ref = R.welch_np(x, nperseg=nfft, nfft=nfft)
        assert relerr(out[i], ref) < RTOL


@pytest.mark.parametrize('ns', [64, 65, 130, 700])
def test_welch_many_streams_tuned_vs_generic(ctx, hip, ns):
    """Stream counts around and beyond the 64 ticket counters and the resident workgroup count: up to 64 streams
    draw chunks from per-stream tickets, more fall back to the interleaved schedule, and beyond one workgroup
    per stream every stream still gets its own."""
    n = 4096 + 2048 * 10 + 37
    d_in = ctx.alloc(ns * n * 8)
    d_a, d_b = ctx.alloc(ns * 4096 * 4), ctx.alloc(ns * 4096 * 4)
    try:
        ctx.synth_iq(d_in, ns * n, 900 + ns, R.TONES, R.DC)
        tuned = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_TUNED)
        gen = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_GENERIC)
        assert tuned.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=n) == 11
        assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=n) == 11
        a = ctx.d2h(d_a, (ns, 4096), np.float32).astype(np.float64)
        b = ctx.d2h(d_b, (ns, 4096), np.float32).astype(np.float64)
        assert np.max(np.abs(a - b) / np.maximum(b, 0.1 * np.median(b))) < 5e-5
        x0 = ctx.d2h(d_in, (n,), np.complex64)
        _, ref = R.welch_np(x0, nperseg=4096, nfft=4096)
        assert relerr(a[0], ref) < RTOL
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


def test_time_sharded_partials_add_up(ctx):
    """Long-stream sharding: partial sums of halo-overlapped chunks add to the whole."""
    nfft, step = 4096, 2048
    x = R.synth_iq(64 * step + nfft - step, 77)          # 64 segments
    plan = ctx.welch_plan(nfft, window=hann(nfft))
    whole = plan.exec(x)
    d_in = ctx.alloc(x.nbytes)
    d_sum = ctx.alloc(2 * nfft * 4)
    d_out = ctx.alloc(nfft * 4)
    try:
        ctx.h2d(d_in, x)
        tot = np.zeros(nfft, np.float64)
        nseg = 0
        for s0, s1 in ((0, 21), (21, 50), (50, 64)):
            n = (s1 - s0 - 1) * step + nfft
            nseg += plan.partial_dev(d_in + 8 * s0 * step, n, d_sum)
            tot += ctx.d2h(d_sum, (nfft,), np.float32)
        assert nseg == 64
        ctx.h2d(d_sum, tot.astype(np.float32))
        plan.scale_dev(d_sum, nseg, d_out)
        out = ctx.d2h(d_out, (nfft,), np.float32)
    finally:
        for p in (d_in, d_sum, d_out):
            ctx.free(p)
    assert relerr(out, whole) < 2e-6


# ------------------------------------------------------------------ CSD ----

@pytest.mark.parametrize('kernel', ['tuned', 'generic'])
def test_csd_coherence_golden(ctx, hip, golden, kernel):
    g = golden('coherence_csd_4096.npz')
    plan = ctx.welch_plan(4096, window=hann(4096), fs=float(g['fs']),
                          kernel=hip.KERNEL_TUNED if kernel == 'tuned' else hip.KERNEL_GENERIC)
    pxx, pyy, pxy, cxy = plan.csd(g['x'], g['y'])
    assert relerr(pxx, g['expected_pxx']) < RTOL
    assert relerr(pyy, g['expected_pyy']) < RTOL
    e = g['expected_pxy']
    assert np.max(np.abs(pxy.astype(np.complex128) - e) / np.abs(e)) < RTOL   # relative to |Pxy| (coherence >= 0.35 here; measured 2e-6)
    assert np.max(np.abs(pxy.astype(np.complex128) - e) / np.sqrt(g['expected_pxx'] * g['expected_pyy'])) < RTOL
    assert np.max(np.abs(cxy - g['expected_cxy'])) < RTOL


def test_csd_many_segments_no_detrend_shifted(ctx, hip):
    """More chunks than workgroups, fftshift + trim, detrend off: tuned CSD kernel vs oracle."""
    n = 4096 + 2048 * 2999
    x = R.synth_iq(n, 61)
    y = (0.5 * np.roll(x, 3) + 0.8 * R.synth_iq(n, 62)).astype(np.complex64)
    _, pxy = R.csd_np(x, y, fs=2.0, nperseg=4096, nfft=4096, detrend=False)
    _, pxx = R.welch_np(x, fs=2.0, nperseg=4096, nfft=4096, detrend=False)
    _, pyy = R.welch_np(y, fs=2.0, nperseg=4096, nfft=4096, detrend=False)
    plan = ctx.welch_plan(4096, window=hann(4096), fs=2.0, detrend=hip.DETREND_NONE, fftshift=True, trim_bins=100,
                          kernel=hip.KERNEL_TUNED)
    gxx, gyy, gxy, gc = plan.csd(x, y)
    assert plan.last_nseg == 3000
    sl = slice(100, -100)
    assert relerr(gxx, np.fft.fftshift(pxx)[sl]) < RTOL and relerr(gyy, np.fft.fftshift(pyy)[sl]) < RTOL
    e = np.fft.fftshift(pxy)[sl]
    scale = np.sqrt(np.fft.fftshift(pxx)[sl] * np.fft.fftshift(pyy)[sl])
    assert np.max(np.abs(gxy.astype(np.complex128) - e) / scale) < RTOL
    assert np.max(np.abs(gc - np.abs(e) ** 2 / scale ** 2)) < RTOL


def test_coherence_of_identical_channels_is_one(ctx):
    x = R.synth_iq(32768, 5)
    _, _, _, cxy = ctx.welch_plan(1024, window=hann(1024)).csd(x, x)
    assert np.max(np.abs(cxy - 1.0)) < 1e-5


# --------------------------------------------------------------- chains ----

def test_chain_sensor_v2_rows_and_mean8(ctx, hip, golden):
    g = golden('gr_chain_rect_1024.npz')
    ch = ctx.chain(1024, None, True, hip.EPI_MAG2_OVER_N2, 1)
    rows, n = ch.push(g['x'])
    assert n == 64 and rows.shape == (64, 1024)
    # round-1 criteria again (plain 1e-4 over all bins happens to hold on this fixture for a CPU fp32 FFT too: 5.2e-5)
    # + the CPU fp32 comparator; measured on MI355X: plain 4.6e-5, floor form 4.1e-5, 0.96 ulp of the peak
    check_single_rows(rows, g['expected_rows'], c64=g['c64_rows'], plain=RTOL, floor_rel=1e-3)
    assert relerr(ctx.rows_group_mean(rows, 8), g['expected_mean8']) < RTOL


def test_chain_psd_logger_mag_and_peak(ctx, hip, golden):
    g = golden('gr_chain_bh_mag_peak_4096.npz')
    ch = ctx.chain(4096, g['window'], False, hip.EPI_MAG, 1)
    ch.set_peak_hold(True)
    rows, n = ch.push(g['x'][:5 * 4096])
    assert n == 5
    # the round-1 criterion of this fixture was the plain 1e-4 on every |X| bin (a CPU fp32 FFT reaches 2.3e-4 here;
    # the HIP rows measure 2.9e-5 with the multiply-then-add butterflies of the chain build)
    check_single_rows(rows, g['expected_mag'][:5], power=False, c64=g['c64_mag'][:5], plain=RTOL)
    check_single_rows(ch.peak(), g['expected_peak'][4], power=False, plain=RTOL)      # the max of five rows still has nulls
    rows, n = ch.push(g['x'][5 * 4096:])
    assert n == 11
    check_single_rows(rows, g['expected_mag'][5:], power=False, c64=g['c64_mag'][5:], plain=RTOL)
    check_single_rows(ch.peak(), g['expected_peak'][-1], power=False, plain=RTOL)


def test_chain_local_worker_iir_log(ctx, hip, golden):
    g = golden('gr_chain_bh_iir_log_2048.npz')
    N, Sf, alpha = 2048, int(g['sample_rate']), float(g['average'])
    k = -10 * np.log10(N) - 10 * np.log10(Sf)
    ch = ctx.chain(N, g['window'], True, hip.EPI_MAG2, 1)
    ch.set_iir_log(alpha, k)
    db_rows = []
    x = g['x']
    for lo, hi in ((0, 3000), (3000, 3001), (3001, 20000), (20000, len(x))):   # chunks that split vectors
        rows, n = ch.push(x[lo:hi])
        db_rows.append(rows)
    db = np.concatenate(db_rows)
    assert db.shape == g['expected_db'].shape
    assert relerr(10 ** ((db.astype(np.float64) - k) / 10), g['expected_lin']) < RTOL
    assert relerr(ch.iir(), g['expected_lin'][-1]) < RTOL


def test_chain_keep_one_in_n_matches_gnuradio_rule(ctx, hip):
    N = 256
    x = R.synth_iq(N * 23 + 17, 8)
    ref = R.chain_sensor_v2(x, N, decim=5)
    ch = ctx.chain(N, None, True, hip.EPI_MAG2_OVER_N2, 5)
    got = []
    for lo, hi in ((0, 700), (700, 701), (701, 4000), (4000, len(x))):
        rows, n = ch.push(x[lo:hi])
        assert n == len(rows)
        got.append(rows)
    got = np.concatenate(got)
    assert got.shape == ref.shape == (4, N)
    check_single_rows(got, ref)
    # latest-wins: a small rows_capacity returns the LAST row only
    ch.reset()
    rows, n = ch.push(x, max_rows=1)
    assert n == 4
    check_single_rows(rows[0], ref[3])


def test_chain_latest_wins_on_the_coverage_kernel(ctx, hip):
    """Sizes the fused launch does not cover (64, 128) and the same calls forced onto the coverage kernels at 8192: a
    stateless chain computes only the rows it hands back, a stateful one (peak hold) all of them."""
    for N in (128, 8192):
        x = R.synth_iq(N * 40 + 11, 70 + N)
        ref = R.chain_sensor_v2(x, N, decim=2)
        ch = ctx.chain(N, None, True, hip.EPI_MAG2_OVER_N2, 2)
        ch.set_kernel(hip.KERNEL_GENERIC)
        rows, n = ch.push(x, max_rows=3)
        assert n == len(ref) == 20 and rows.shape == (3, N)
        check_single_rows(rows, ref[-3:])
        ch.reset()
        rows, n = ch.push(x[:N * 7 + 5], max_rows=1)          # kept vectors 1, 3, 5 -> rows 0..2
        rows2, n2 = ch.push(x[N * 7 + 5:], max_rows=1)
        assert n + n2 == 20
        check_single_rows(rows[0], ref[n - 1])
        check_single_rows(rows2[0], ref[-1])
        from ofdm_tools import windows
        mag, peak = R.chain_psd_logger(x, N, decim=2)
        ch = ctx.chain(N, windows.blackmanharris(N), False, hip.EPI_MAG, 2)
        ch.set_kernel(hip.KERNEL_GENERIC)
        ch.set_peak_hold(True)
        rows, n = ch.push(x, max_rows=1)
        assert n == 20 and relerr(ch.peak(), peak[-1]) < RTOL
        check_single_rows(rows[0], mag[-1], power=False)


# ------------------------------------------------------- channel power ----

def test_channel_power_cases(ctx, golden):
    g = golden('src_power_cases.npz')
    from ofdm_tools import ofdm_cr_tools as T
    for i in range(int(g['n'])):
        Sf, N = int(g['Sf_%d' % i]), int(g['N_%d' % i])
        cs, sbw = float(g['cs_%d' % i]), float(g['sbw_%d' % i])
        Fr = float(Sf) / N
        psd = g['psd_%d' % i].astype(np.float32)
        bb = T.frange(-Sf // 2, Sf // 2, cs)
        ref = R.src_power(psd, N, Fr, Sf, bb, sbw / Fr)
        got = T.src_power(psd, N, Fr, Sf, bb, sbw / Fr)
        assert len(got) == len(ref)
        assert np.allclose(got, ref, rtol=1e-5)
        assert np.allclose(T.movingaverage(psd, sbw / Fr), R.movingaverage(psd, sbw / Fr), rtol=1e-5)


# ------------------------------------------------------------ xcorr/fac ----

def test_xcorr_fac_golden(ctx, golden):
    g = golden('xcorr_fac.npz')
    L = int(g['L'])
    xc = ctx.xcorr(g['a'], g['b'], L)
    ref = g['expected_xcorr']
    assert np.max(np.abs(xc - ref)) / np.max(ref) < 1e-5
    assert int(np.argmax(xc)) == 37
    fc = ctx.fac(g['a'], L)
    assert np.max(np.abs(fc - g['expected_fac'])) / np.max(g['expected_fac']) < 1e-5
    # the same against the reference's OWN xcorr / fac output (ref_xcorr_fac.npz)
    r = golden('ref_xcorr_fac.npz')
    assert np.max(np.abs(xc - r['expected_xcorr'])) / np.max(r['expected_xcorr']) < 1e-5
    assert np.max(np.abs(fc - r['expected_fac'])) / np.max(r['expected_fac']) < 1e-5
    xs = ctx.xcorr(g['a'][:700], g['b'][:900], 1024)
    assert np.max(np.abs(xs - r['expected_xcorr_short'])) / np.max(r['expected_xcorr_short']) < 1e-5


# ------------------------------ launches large enough for the static default schedules ----

@pytest.mark.parametrize('nfft', [256, 512, 1024])
def test_large_launch_default_schedules(ctx, hip, nfft):
    """2^26 samples on the device: at this size the library's own defaults pick the interleaved static schedules
    (256 / 512 points at 50 % overlap; whole-segment loads at other steps) - tuned against the coverage kernel,
    and the first 2^18 samples against the float64 oracle."""
    n = 1 << 26
    d_in = ctx.alloc(n * 8)
    try:
        ctx.synth_iq(d_in, n, 1002, R.TONES, R.DC)
        for nov in (nfft // 2, 0, nfft // 4):
            tuned = ctx.welch_plan(nfft, noverlap=nov, window=hann(nfft), fs=1.0, kernel=hip.KERNEL_TUNED)
            gen = ctx.welch_plan(nfft, noverlap=nov, window=hann(nfft), fs=1.0, kernel=hip.KERNEL_GENERIC)
            a, b = tuned.exec_device_src(d_in, n), gen.exec_device_src(d_in, n)
            assert tuned.last_nseg == gen.last_nseg == (n - nov) // (nfft - nov)
            assert relerr(a, b) < 2e-5, (nfft, nov)
            again = tuned.exec_device_src(d_in, n)
            assert np.array_equal(a, again), 'a static schedule is bit-reproducible'
            m = 1 << 18
            x = ctx.d2h(d_in, (m,), np.complex64)
            _, ref = R.welch_np(x, fs=1.0, nperseg=nfft, noverlap=nov, nfft=nfft)
            assert relerr(tuned.exec_device_src(d_in, m), ref) < RTOL
    finally:
        ctx.free(d_in)


# -------------------------------------------- full size (BASELINE config 2) ----

def test_full_size_256M_properties(ctx, hip):
    """2^28 samples (2 GiB) generated on the device: the oracle cannot run here, so check
    (1) the tuned kernel against the independent generic kernel, (2) Parseval against a
    separate reduction of the raw samples, (3) the prefix against the float64 oracle."""
    n, nfft = 1 << 28, 4096
    d_in = ctx.alloc(n * 8)
    try:
        ctx.synth_iq(d_in, n, 1002, R.TONES, R.DC)
        plan = ctx.welch_plan(nfft, window=hann(nfft), fs=1.0, kernel=hip.KERNEL_TUNED)
        tuned = plan.exec_device_src(d_in, n)
        assert plan.last_nseg == 131071
        gen = ctx.welch_plan(nfft, window=hann(nfft), fs=1.0, kernel=hip.KERNEL_GENERIC).exec_device_src(d_in, n)
        assert relerr(tuned, gen) < 2e-5
        mean, var = ctx.iq_power(d_in, n)
        assert abs(mean - R.DC) < 1e-3
        # sum_k P[k] * fs / nfft ~= mean |x - mean|^2 (Hann-weighted estimate; 131071 segments)
        assert abs(tuned.astype(np.float64).sum() / nfft - var) / var < 2e-3
        pre = ctx.d2h(d_in, (1 << 20,), np.complex64)
        _, ref = R.welch_np(pre, nperseg=nfft, nfft=nfft)
        assert relerr(plan.exec(pre), ref) < RTOL
    finally:
        ctx.free(d_in)


def test_welch8192_role_split_variant_against_the_oracle(ctx, hip):
    """The role-split 8192-point build (csrc/welch16k1x.hip `welch8kws_kernel`; the default at 50 % overlap since late
    round 5, "8k1role" selects the one-role kernel): producers and consumers one segment apart on two LDS images.  Runs of
    one, two, a few and many segments per workgroup, detrend through the pilot / raw / none, a DC offset 30 x the noise -
    against the float64 oracle and against the one-role build."""
    N = 8192
    rng = np.random.default_rng(8192)
    for nseg in (1, 2, 9, 300, 1100):
        n = N + (N // 2) * (nseg - 1) + (17 if nseg > 2 else 0)
        x = (R.synth_iq(n, 31 + nseg) + (30.0 - 18.0j) * (nseg >= 300)).astype(np.complex64)      # (a lone periodogram under a 30 sigma line sits at the fp32 floor)
        for det, name in ((hip.DETREND_CONSTANT, 'constant'), (hip.DETREND_CONSTANT_FAST, 'constant'), (hip.DETREND_NONE, False)):
            _, ref = R.welch_np(x, nperseg=N, nfft=N, detrend=name)
            plan = ctx.welch_plan(N, window=hann(N), detrend=det, fs=1.0, kernel=hip.KERNEL_TUNED)
            plan.set_tuning('8kws')
            got = plan.exec(x)
            assert plan.last_nseg == nseg and ':ws' in plan.last_recipe(), plan.last_recipe()
            e = np.abs(got - ref) / ref
            # _FAST works on the raw samples: the bins under the removed DC line carry the float32 mean's rounding
            bound = RTOL if det != hip.DETREND_CONSTANT_FAST else 2e-3
            if nseg <= 2:      # one or two periodograms of noise + tones: bins in the nulls sit at the fp32 floor of ANY single-precision FFT
                bound = 1e-3
            assert e.max() < bound, (nseg, det, e.max(), int(np.argmax(e)))
            if det != hip.DETREND_CONSTANT_FAST:
                one = ctx.welch_plan(N, window=hann(N), detrend=det, fs=1.0, kernel=hip.KERNEL_TUNED)
                one.set_tuning('8k1role')
                assert relerr(got, one.exec(x)) < (2e-5 if nseg >= 8 else 1e-3)      # (few segments: the default is the time-domain form)
                assert ':ws' not in one.last_recipe()
                one.close()
            plan.close()
    # a schedule the build does not walk falls back to the one-role kernel
    plan = ctx.welch_plan(N, window=hann(N), fs=1.0, kernel=hip.KERNEL_TUNED)
    plan.set_schedule(hip.SCHED_INTERLEAVED)
    plan.set_tuning('8kws')
    plan.exec(R.synth_iq(N * 20, 3))
    assert ':ws' not in plan.last_recipe()


def test_results_do_not_depend_on_timing_under_a_bandwidth_hog(ctx, hip):
    """A register read in front of the wait for its load gives the right answer as long as the load happens to be back -
    the latent kind of bug tools/isa_async_hazard.py looks for statically.  This is the dynamic side: every tuned build
    with a static (bit-reproducible) schedule is run alone, then again and again while a second context streams 8 GiB
    reads over the same HBM from another stream, which stretches every load's latency; the results must be IDENTICAL bit
    for bit (Welch / two-channel sums) resp. row for row (fused chain)."""
    import threading
    n = 1 << 24
    d = ctx.alloc((n + 5) * 8)
    rows_d = 0
    hog = hip.Context(0)
    hog_bytes = 8 << 30
    hog_buf = hog.alloc(hog_bytes)
    try:
        ctx.synth_iq(d, n + 5, 4242, R.TONES, R.DC)
        cases = []
        for nfft in (256, 512, 1024, 2048, 4096, 8192, 16384):
            for det in (hip.DETREND_CONSTANT, hip.DETREND_CONSTANT_FAST, hip.DETREND_NONE):
                cases.append(('welch %d det %d' % (nfft, det), dict(nfft=nfft, window=hann(nfft), detrend=det), None))
        cases.append(('welch 8192 one-role', dict(nfft=8192, window=hann(8192)), '8k1role'))
        cases.append(('welch 8192 one-role, detrend none', dict(nfft=8192, window=hann(8192), detrend=hip.DETREND_NONE), '8k1role'))
        cases.append(('welch 4096 pipe', dict(nfft=4096, window=hann(4096)), 'pipe'))
        cases.append(('welch 4096 zero-padded', dict(nfft=4096, nperseg=1024, window=flattop(1024)), None))
        for nfft in (8192, 16384):      # the scanner's vectors: no overlap, rectangular
            cases.append(('scan %d' % nfft, dict(nfft=nfft, noverlap=0, window=None, detrend=hip.DETREND_NONE,
                                                 scaling=hip.SCALE_OVER_N2), None))
        plans = []
        for name, kw, variant in cases:
            plan = ctx.welch_plan(fs=1.0, kernel=hip.KERNEL_TUNED, **kw)
            plan.set_schedule(hip.SCHED_CONTIGUOUS)
            if variant:
                try:
                    plan.set_tuning(variant)
                except hip.HipError:      # (an A/B library of an earlier commit, tools/ab_*.sh)
                    continue
            plans.append((name, plan, lambda p=plan: p.exec_device_src(d, n)))
        csd = ctx.welch_plan(4096, window=hann(4096), fs=1.0, kernel=hip.KERNEL_TUNED)
        csd.set_schedule(hip.SCHED_CONTIGUOUS)
        plans.append(('csd 4096', csd, lambda: np.concatenate([np.asarray(v).view(np.float32).ravel()
                                                             for v in csd.csd_device_src(d + 40, d, n)])))
        chains = []
        for nfft in (1024, 4096, 8192, 16384):
            ch = ctx.chain(nfft, None, True, hip.EPI_MAG2, 64)
            chains.append(ch)
            plans.append(('chain %d' % nfft, ch, lambda c=ch: c.push(ctx.d2h(d, (1 << 20,), np.complex64), 4)[0].ravel()))
        # the default (ticket) schedules: the summation order follows the timing, the values may move by rounding only
        loose = []
        for nfft in (1024, 2048, 4096):
            plan = ctx.welch_plan(nfft, window=hann(nfft), fs=1.0, kernel=hip.KERNEL_TUNED)
            loose.append(('welch %d default schedule' % nfft, plan, lambda p=plan: p.exec_device_src(d, n)))
        quiet = [run().copy() for _, _, run in plans]
        quiet_loose = [run().astype(np.float64) for _, _, run in loose]
        stop = threading.Event()

        def stream_reads():
            while not stop.is_set():
                hog.stream_read_probe(hog_buf, hog_bytes, 4)
        th = threading.Thread(target=stream_reads, daemon=True)
        th.start()
        try:
            for rep in range(10):
                for (name, _, run), want in zip(plans, quiet):
                    got = run()
                    assert got.tobytes() == want.tobytes(), (name, rep, float(np.max(np.abs(got - want) / np.abs(want))))
                for (name, _, run), want in zip(loose, quiet_loose):
                    assert relerr(run(), want) < 5e-6, (name, rep)
        finally:
            stop.set()
            th.join(60)
        for _, plan, _ in plans + loose:
            plan.close()
    finally:
        ctx.free(d)
        hog.free(hog_buf)
        hog.close()


@pytest.mark.parametrize('nfft', [256, 1024, 4096, 8192, 16384])
def test_chain_random_ragged_pushes_carry_their_state(ctx, hip, nfft):
    """The streaming state of the fused chain - the samples of a vector split across pushes, the phase of keep_one_in_n,
    the peak and the filter - over twenty pushes of random length (shorter than a vector, empty, thousands of vectors),
    for random keep values and both window / shift forms: the number of rows of every push, the mean of ALL rows, the
    final peak and the last rows against the oracle run once over the concatenated samples."""
    from ofdm_tools import windows
    rng = np.random.default_rng(nfft)
    for trial in range(3):
        keep = int(rng.choice([1, 2, 3, 7]))
        shift = bool(rng.integers(0, 2))
        w = windows.blackmanharris(nfft) if rng.integers(0, 2) else None
        total = nfft * int(rng.integers(150, 400)) + int(rng.integers(0, nfft))
        x = R.synth_iq(total, 7000 + nfft + trial)
        cuts = np.sort(np.r_[0, rng.integers(0, total, 17), rng.integers(0, total, 1).repeat(2), total])      # (one empty push)
        X = R.gr_fft_vcc(R.gr_kept_vectors(x, nfft, keep), w, shift)
        want = X.real ** 2 + X.imag ** 2
        ch = ctx.chain(nfft, w, shift, hip.EPI_MAG2, keep)
        ch.set_peak_hold(True)
        got, produced = [], 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            rows, n = ch.push(x[a:b], max_rows=(b - a) // nfft + 2)
            vectors_so_far = b // nfft
            assert produced + n == vectors_so_far // keep, (trial, a, b)
            assert len(rows) == n
            produced += n
            got.append(rows)
        got = np.concatenate(got)
        assert got.shape == want.shape
        assert relerr(got.astype(np.float64).mean(axis=0), want.mean(axis=0)) < RTOL
        assert relerr(ch.peak(), want.max(axis=0)) < RTOL
        check_single_rows(got[-3:], want[-3:])


def test_random_plan_shapes_tuned_route_against_the_coverage_kernel():
    """tools/fuzz_shapes.py: 300 random plans (size, nperseg = nfft, nfft / 2, nfft / 4, any overlap, four windows, three
    detrend modes, 1 ... 64 streams with padded strides, 1 ... 4000 segments, forced schedules and chunk sizes) - whatever
    build resolve_recipe() routes them to against the independent coverage kernel on the same samples, 3e-5 (2400 cases
    over four more seeds ran clean when this was added)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_shapes.py'), '300', '11'], capture_output=True, timeout=600)
    assert p.returncode == 0 and b'300 cases, 0 mismatches' in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_two_contexts_in_two_threads_agree_with_the_serial_run(hip):
    """include/ofdm_tools_hip.h: a context is used from one thread at a time, DIFFERENT contexts from different threads
    freely (no global mutable state in the library).  Two host threads, each with its own context, stream and plans
    (one- and two-channel Welch at several sizes, the fused chain, the decision stage), run the same work first one
    after the other and then at the same time: static schedules, so every result must be identical bit for bit."""
    import threading
    n = 1 << 22

    def work(seed, out, rounds):
        c = hip.Context(0)
        d = c.alloc((n + 5) * 8)
        try:
            c.synth_iq(d, n + 5, seed, R.TONES, R.DC)
            plans = []
            for nfft in (256, 1024, 4096, 16384):
                p = c.welch_plan(nfft, window=hann(nfft), fs=1.0, kernel=hip.KERNEL_TUNED)
                p.set_schedule(hip.SCHED_CONTIGUOUS)
                plans.append(p)
            ch = c.chain(2048, None, True, hip.EPI_MAG2, 16)
            x = c.d2h(d, (1 << 18,), np.complex64)
            for _ in range(rounds):
                res = [p.exec_device_src(d, n).copy() for p in plans]
                res.append(np.concatenate([np.asarray(v).view(np.float32).ravel() for v in plans[2].csd_device_src(d + 40, d, n)]))
                res.append(ch.push(x, 4)[0].ravel().copy())
                pw, ma = c.channel_power(res[2], 25.0, [0, 100, 2000], [50, 300, 4096], want_movavg=True)
                res.append(np.asarray(ma).ravel().copy())
                res.append(np.asarray(pw).ravel().copy())
                out.append(res)
            for p in plans:
                p.close()
        finally:
            c.free(d)
            c.close()

    serial = {7: [], 8: []}
    for seed in serial:
        work(seed, serial[seed], 1)
    both = {7: [], 8: []}
    threads = [threading.Thread(target=work, args=(seed, both[seed], 4)) for seed in both]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive()
    for seed in serial:
        assert len(both[seed]) == 4
        for res in both[seed]:
            for a, b in zip(res, serial[seed][0]):
                assert a.tobytes() == b.tobytes(), seed


def test_known_answers_independent_of_any_library(ctx, hip):
    """SURVEY 8c's checks that need no oracle, on the HIP path itself:
      * a pure tone on bin k0 through the rectangular |X|^2 / N^2 chain puts A^2 into bin k0 and nothing elsewhere;
      * unit-power white noise has a flat two-sided density 1 / fs (Parseval: the bins sum to the power);
      * x against itself has magnitude-squared coherence 1 on every bin;
      * two independent noises have E[Cxy] of the order 1 / (number of averages)."""
    N, k0, A = 4096, 517, 0.75
    tone = (A * np.exp(2j * np.pi * k0 * np.arange(8 * N) / N)).astype(np.complex64)
    ch = ctx.chain(N, None, True, hip.EPI_MAG2_OVER_N2, 1)
    rows, n = ch.push(tone, 8)
    assert n == 8
    for row in rows:
        assert abs(row[N // 2 + k0] - A * A) < 2e-6 * A * A and (row.astype(np.float64).sum() - row[N // 2 + k0]) < 1e-9
    rng = np.random.default_rng(77)
    fs, nseg = 2.0e6, 400
    n_ = N + (N // 2) * (nseg - 1)
    x = ((rng.standard_normal(n_) + 1j * rng.standard_normal(n_)) / np.sqrt(2)).astype(np.complex64)
    y = ((rng.standard_normal(n_) + 1j * rng.standard_normal(n_)) / np.sqrt(2)).astype(np.complex64)
    plan = ctx.welch_plan(N, window=hann(N), fs=fs, kernel=hip.KERNEL_TUNED)
    p = plan.exec(x).astype(np.float64)
    assert abs(p.sum() * fs / N - 1.0) < 0.01                        # Parseval
    flat = np.delete(p, [0, 1, N - 1]) * fs          # (the constant detrend takes the mean out of bins 0, +-1)
    assert abs(np.median(flat) - 1.0) < 0.05 and flat.max() < 1.4 and flat.min() > 0.7      # 400 averages, 4096 bins
    pxx, pyy, pxy, cxy = plan.csd(x, x)
    assert np.max(np.abs(cxy - 1.0)) < 1e-5 and np.max(np.abs(pxy.imag)) < 1e-6 * np.max(pxx)
    _, _, _, cxy = plan.csd(x, y)
    # 400 Hann segments at 50 % overlap are worth about 400 / 1.06 ... 400 / 1.9 independent averages
    assert 1.0 / nseg < cxy.mean() < 2.5 / nseg, cxy.mean() * nseg
    plan.close()


def test_one_launch_over_16_GiB_sample_offsets_beyond_2_31(ctx, hip):
    """A stream of 2^31 + 2^22 samples (16 GiB) in ONE launch: sample indices pass 2^31 and byte offsets 2^34 inside the
    kernels.  The last 2^22 samples carry a tone 30 dB above everything before them, so a segment fetched from a wrapped
    offset shows; the sums of a Welch average are additive over runs of segments, so
        nseg P(whole) = k_A P(first k_A segments) + k_B P(the rest)
    with the rest launched on its own from a pointer 16 GiB into the buffer.  Checked for the builds the BASELINE
    configurations route to: 4096 role-split, 256, 1024 role-split, 16384 one-exchange with and without overlap."""
    head, tail = 1 << 31, 1 << 22
    n = head + tail
    d = ctx.alloc(n * 8)
    try:
        ctx.synth_iq(d, head, 1002, R.TONES, R.DC)
        ctx.synth_iq(d + head * 8, tail, 77, ((30.0, 0.2003),), 0j)
        cases = [(4096, hann(4096), None, hip.DETREND_CONSTANT, 'welch4096:ws'),
                 (256, hann(256), None, hip.DETREND_CONSTANT, 'seg'),
                 (1024, hann(1024), None, hip.DETREND_CONSTANT, 'segws'),
                 (16384, hann(16384), None, hip.DETREND_CONSTANT, 'welch16k1x_half'),
                 (16384, None, 0, hip.DETREND_NONE, 'welch16k1x')]
        for nfft, win, noverlap, det, kern in cases:
            plan = ctx.welch_plan(nfft, window=win, noverlap=noverlap, detrend=det, fs=1.0, kernel=hip.KERNEL_TUNED)
            step = nfft if noverlap == 0 else nfft // 2
            whole = plan.exec_device_src(d, n).astype(np.float64)
            nseg = plan.last_nseg
            assert kern in plan.last_recipe(), plan.last_recipe()
            assert nseg == (n - (nfft - step)) // step
            k_a = head // step                                   # segments that start in front of the tail
            a = plan.exec_device_src(d, k_a * step + (nfft - step)).astype(np.float64)
            assert plan.last_nseg == k_a
            b = plan.exec_device_src(d + head * 8, tail).astype(np.float64)
            k_b = plan.last_nseg
            assert k_a + k_b == nseg
            both = (k_a * a + k_b * b) / nseg
            assert relerr(whole, both) < 2e-5, (nfft, noverlap)
            # and the tail's tone is where it belongs, at the power its share of the segments gives it
            k0 = int(round(0.2003 * nfft))
            assert np.argmax(b) == k0 and whole[k0] > 100 * np.median(whole)
            plan.close()
        # the two-channel kernel on the same buffer: y = x delayed by five samples, both streams 16 GiB long
        delay, N, step = 5, 4096, 2048
        plan = ctx.welch_plan(N, window=hann(N), fs=1.0, kernel=hip.KERNEL_TUNED)
        m = n - delay
        whole = [v.astype(np.complex128) for v in plan.csd_device_src(d + 8 * delay, d, m)[:3]]
        nseg = plan.last_nseg
        assert 'csd4096ws' in plan.last_recipe() and nseg == (m - step) // step
        k_a = head // step
        a = [v.astype(np.complex128) for v in plan.csd_device_src(d + 8 * delay, d, k_a * step + step)[:3]]
        assert plan.last_nseg == k_a
        b = [v.astype(np.complex128) for v in plan.csd_device_src(d + 8 * (delay + head), d + 8 * head, m - head)[:3]]
        k_b = plan.last_nseg
        assert k_a + k_b == nseg
        for w, pa, pb in zip(whole, a, b):
            both = (k_a * pa + k_b * pb) / nseg
            assert np.max(np.abs(w - both)) / np.max(np.abs(both)) < 2e-5
            assert np.max(np.abs(w - both) / np.maximum(np.abs(both), 1e-3 * np.median(np.abs(both)))) < 1e-3
        plan.close()
    finally:
        ctx.free(d)


def test_full_size_config3_csd_properties(ctx, hip):
    """BASELINE config 3 at its full size (2 x 2^26 samples on the device; the oracle cannot run there): y is x delayed by
    five samples (the same buffer, five samples in), so the truth is known in closed form -
      (1) the role-split kernel against the independent coverage kernel on Pxx, Pyy, Pxy;
      (2) Pxx of the pair equals the one-channel Welch PSD of x;
      (3) magnitude-squared coherence = 1 and the phase of Pxy = conj(X) Y is that of a five-sample delay,
          -2 pi 5 f0, at the bin of every tone f0;
      (4) csd(y, x) = conj(csd(x, y));
      (5) a 2^20-sample prefix against the float64 oracle."""
    n, N, delay = 1 << 26, 4096, 5
    d_in = ctx.alloc((n + delay) * 8)
    try:
        ctx.synth_iq(d_in, n + delay, 1003, R.TONES, R.DC)
        dx, dy = d_in + 8 * delay, d_in                      # y[i] = x[i - delay]
        tuned = ctx.welch_plan(N, window=hann(N), fs=1.0, kernel=hip.KERNEL_TUNED)
        pxx, pyy, pxy, cxy = tuned.csd_device_src(dx, dy, n)
        assert tuned.last_nseg == (n - 2048) // 2048
        gen = ctx.welch_plan(N, window=hann(N), fs=1.0, kernel=hip.KERNEL_GENERIC)
        gxx, gyy, gxy, gc = gen.csd_device_src(dx, dy, n)
        assert relerr(pxx, gxx) < 2e-5 and relerr(pyy, gyy) < 2e-5
        assert np.max(np.abs(pxy - gxy) / np.sqrt(gxx.astype(np.float64) * gyy)) < 2e-5
        assert relerr(pxx, tuned.exec_device_src(dx, n)) < 2e-5
        # a pure delay: |Pxy|^2 = Pxx Pyy up to the segment edges (5 of 4096 samples differ per segment)
        assert np.all(cxy < 1.0 + 1e-5) and cxy.min() > 0.99
        for amp, f0 in R.TONES:                              # at a tone's bin Pxy = conj(X) Y has the phase -2 pi d f0
            kc = int(round(f0 * N)) % N
            dphi = np.angle(pxy[kc] * np.exp(2j * np.pi * delay * f0))
            assert abs(dphi) < 1e-3, (f0, dphi)
        qxx, qyy, qxy, _ = tuned.csd_device_src(dy, dx, n)
        assert relerr(qxx, pyy) < 2e-6 and relerr(qyy, pxx) < 2e-6
        assert np.max(np.abs(qxy - np.conj(pxy)) / np.sqrt(pxx.astype(np.float64) * pyy)) < 2e-6
        m = 1 << 20
        buf = ctx.d2h(d_in, (m + delay,), np.complex64)
        _, rc, rxx, ryy, rxy = R.coherence_np(buf[delay:], buf[:m], nperseg=N, nfft=N)
        hxx, hyy, hxy, hc = tuned.csd(buf[delay:], buf[:m])
        assert relerr(hxx, rxx) < RTOL and relerr(hyy, ryy) < RTOL
        assert np.max(np.abs(hxy - rxy) / np.sqrt(rxx * ryy)) < RTOL and np.max(np.abs(hc - rc)) < RTOL
    finally:
        ctx.free(d_in)


def test_full_size_config5_scanner_properties(ctx, hip):
    """BASELINE config 5 at its full size (64 channel streams x 2^22 samples, 16384-point rectangular |X|^2 / N^2 mean
    + the device decision stage): (1) the pipelined one-exchange kernel against the coverage kernel on all 64 rows;
    (2) Parseval per row - with a rectangular window and no overlap the row sums to the stream's mean power exactly
    (here: to float32 summation); (3) identical streams give identical rows, a stream scaled by 2 a row scaled by 4 and
    the same mask; (4) the decision stage's noise floor / mask / channel sums against the restatement on the rows;
    (5) a prefix of one stream against the float64 oracle."""
    from ofdm_tools.scan_batch import BatchScanPlan
    N, ns, per = 16384, 64, 1 << 22
    d_in, d_out, d_gen = ctx.alloc(ns * per * 8), ctx.alloc(ns * N * 4), ctx.alloc(ns * N * 4)
    try:
        for i in range(ns):                                  # streams 0 and 1 identical
            ctx.synth_iq(d_in + i * per * 8, per, 4000 + max(i, 1), R.TONES, R.DC)
        bp = BatchScanPlan(ctx, N, 1000000, 15625.0, 10e3, thr_leveler=3)
        assert bp.psd_rows_dev(d_in, per, ns, per, d_out) == per // N
        rows = ctx.d2h(d_out, (ns, N), np.float32)
        gen = ctx.welch_plan(N, noverlap=0, window=None, detrend=hip.DETREND_NONE, scaling=hip.SCALE_OVER_N2,
                             fftshift=True, kernel=hip.KERNEL_GENERIC)
        assert gen.exec_dev(d_in, per, d_gen, nstreams=ns, stream_stride=per) == per // N
        assert relerr(rows, ctx.d2h(d_gen, (ns, N), np.float32)) < 2e-5
        assert np.array_equal(rows[0], rows[1])
        for i in (1, 17, 63):
            mean, var = ctx.iq_power(d_in + i * per * 8, per)
            power = var + abs(mean) ** 2                     # mean |x|^2
            assert abs(rows[i].astype(np.float64).sum() - power) / power < 1e-5, i
        mask, noise, plc = bp.decide_dev(d_out, ns)
        st = R.ScannerState(N, 1000000, 15625.0, 10e3, trunc_band=1000000)
        for i in (0, 1, 40):
            ref = rows[i].astype(np.float64)
            ma = R.movingaverage(ref, st.srch_bins)
            assert np.isclose(noise[i], ma.min(), rtol=1e-5)
            want = rows[i] > np.float32(3) * noise[i]
            assert np.array_equal(mask[i].astype(bool), want)
            assert np.allclose(plc[i], R.src_power(ref, N, st.Fr, 1000000, st.bb_freqs, st.srch_bins), rtol=1e-5)
        assert np.array_equal(mask[0], mask[1]) and 0 < mask[0].sum() < N
        x = ctx.d2h(d_in + 5 * per * 8, (N * 6,), np.complex64)
        x2 = (2 * x).astype(np.complex64)
        d_s = ctx.alloc(2 * x.nbytes)
        try:
            ctx.h2d(d_s, np.concatenate((x, x2)))
            assert bp.psd_rows_dev(d_s, len(x), 2, len(x), d_out) == 6
            pair = ctx.d2h(d_out, (2, N), np.float32)
            assert np.array_equal(pair[1], 4 * pair[0])      # powers of two scale exactly in float32
            m2, n2, _ = bp.decide_dev(d_out, 2)
            assert np.array_equal(m2[0], m2[1]) and np.isclose(n2[1], 4 * n2[0], rtol=1e-6)
        finally:
            ctx.free(d_s)
        assert relerr(pair[0], R.chain_sensor_v2(x, N).mean(axis=0)) < RTOL
    finally:
        for ptr in (d_in, d_out, d_gen):
            ctx.free(ptr)


def test_beyond_4GiB_offsets(ctx, hip):
    """2^29 + 5000 samples (4 GiB + a ragged tail): byte offsets pass 2^32.  The PSD of the whole stream equals
    the segment-weighted mean of the PSDs of its two halves cut with a 2048-sample halo (integer/pointer
    overflow anywhere in the schedule or the loads would break the identity), tuned and generic kernels."""
    nfft = 4096
    n = (1 << 29) + 5000
    nseg = (n - 2048) // 2048
    s1 = nseg // 2 + 3                       # segments in the first part
    n1 = 2048 * (s1 + 1)                     # its samples; the second part starts s1 hops in
    d_in = ctx.alloc(n * 8)
    try:
        ctx.synth_iq(d_in, n, 77, R.TONES, R.DC)
        for kern in (hip.KERNEL_TUNED, hip.KERNEL_GENERIC):
            plan = ctx.welch_plan(nfft, window=hann(nfft), fs=1.0, kernel=kern)
            whole = plan.exec_device_src(d_in, n).astype(np.float64)
            assert plan.last_nseg == nseg
            a = plan.exec_device_src(d_in, n1).astype(np.float64)
            assert plan.last_nseg == s1
            b = plan.exec_device_src(d_in + 8 * 2048 * s1, n - 2048 * s1).astype(np.float64)
            assert plan.last_nseg == nseg - s1
            mix = (a * s1 + b * (nseg - s1)) / nseg
            assert relerr(whole, mix) < 2e-5, kern
            plan.close()
    finally:
        ctx.free(d_in)


# ------------------------------------------------- randomized plan sweep ----

def test_randomized_welch_plans_vs_oracle(ctx, hip):
    """60 random plans (sizes, segment lengths, overlaps, windows, detrend, scaling, shift/trim/dB,
    kernel choice) against the float64 oracle - catches dispatch mistakes between the kernels."""
    rng = np.random.default_rng(20261004)
    x_all = R.synth_iq(300000, 99)
    for it in range(60):
        nfft = int(rng.choice([64, 256, 1024, 2048, 4096, 4096, 4096, 8192, 16384]))
        if rng.random() < 0.5:
            nperseg = nfft
        else:
            nperseg = int(rng.choice([nfft // 4, nfft // 2, nfft - 1, max(8, nfft // 3)]))
        noverlap = int(rng.choice([0, nperseg // 2, nperseg // 4, nperseg - 1, min(nperseg - 1, 7)]))
        detrend = bool(rng.integers(2))
        scaling = str(rng.choice(['density', 'spectrum']))
        fs = float(rng.choice([1.0, 2.5e6]))
        shift = bool(rng.integers(2))
        trim = int(rng.choice([0, 0, nfft // 16]))
        db = bool(rng.integers(2))
        wkind = str(rng.choice(['hann', 'flattop', 'blackmanharris', 'random']))
        w = (R.get_window(wkind, nperseg) if wkind != 'random' else 0.2 + rng.random(nperseg))
        n = int(rng.integers(nperseg, min(len(x_all), nperseg + 40 * max(1, nperseg - noverlap)) + 1))
        x = x_all[:n]
        _, ref = R.welch_np(x, fs=fs, window=w, nperseg=nperseg, noverlap=noverlap, nfft=nfft,
                            detrend='constant' if detrend else False, scaling=scaling)
        if shift:
            ref = np.fft.fftshift(ref)
        if trim:
            ref = ref[trim:-trim]
        kern = hip.KERNEL_GENERIC if rng.random() < 0.25 else hip.KERNEL_AUTO
        plan = ctx.welch_plan(nfft, nperseg=nperseg, noverlap=noverlap, window=w,
                              detrend=hip.DETREND_CONSTANT if detrend else hip.DETREND_NONE,
                              scaling=hip.SCALE_DENSITY if scaling == 'density' else hip.SCALE_SPECTRUM, fs=fs,
                              fftshift=shift, trim_bins=trim, db=db, kernel=kern)
        got = plan.exec(x).astype(np.float64)
        if db:
            got = 10 ** (got / 10)
        assert got.shape == ref.shape
        # a detrended rectangular-ish window can leave a bin near zero; judge against the spectrum's scale
        err = np.max(np.abs(got - ref) / np.maximum(ref, 1e-6 * ref.max()))
        assert err < RTOL, (it, nfft, nperseg, noverlap, detrend, scaling, wkind, shift, trim, db, kern, err)
        plan.close()


@pytest.mark.parametrize('variant', ['', 'ws', 'pipe', 'dpp'])
def test_tuned_vs_generic_on_awkward_segment_counts(ctx, hip, variant):
    """Chunked schedules at the edges: segment counts around multiples of the chunk size and of the
    resident workgroup count, one to three streams, all three schedules, default and tiny chunks
    (one- and two-segment chunks take their own paths in the wave-specialised kernel) - every build of
    the tuned kernel vs the independent generic kernel on device-resident data."""
    rng = np.random.default_rng(7)
    nmax = 4096 + 2048 * 9000
    d_in = ctx.alloc(3 * nmax * 8)
    d_a, d_b = ctx.alloc(3 * 4096 * 4), ctx.alloc(3 * 4096 * 4)
    try:
        ctx.synth_iq(d_in, 3 * nmax, 31, R.TONES, R.DC)
        tuned = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_TUNED)
        gen = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_GENERIC)
        counts = [1, 2, 3, 4, 6, 7, 8, 9, 15, 16, 17, 1023, 1024, 1025, 1026, 4095, 4096, 4097, 8191, 8192, 8193, 9000] + \
            [int(v) for v in rng.integers(1, 9000, 6)]
        for nseg in counts:
            n = 4096 + 2048 * (nseg - 1) + int(rng.integers(0, 2048))
            ns = int(rng.integers(1, 4))
            sched = int(rng.integers(0, 3))
            chunk = int(rng.integers(0, 6))
            tuned.set_tuning(variant or None, chunk=chunk)
            tuned.set_schedule(sched)
            assert tuned.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
            assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
            a = ctx.d2h(d_a, (ns, 4096), np.float32)
            b = ctx.d2h(d_b, (ns, 4096), np.float32)
            # few-segment periodograms have near-empty bins: judge those against the spectrum's typical level
            # (two fp32 FFTs differ by ~1e-6 of the typical amplitude, which is 1e-5 of a bin 10x below it)
            err = np.max(np.abs(a.astype(np.float64) - b) / np.maximum(b, 0.1 * np.median(b)))
            assert err < 5e-5, (variant, nseg, ns, sched, chunk, err)
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


@pytest.mark.parametrize('variant', ['', 'pipe', 'ws'])
def test_welch4096_large_dc_offset(ctx, hip, variant):
    """A DC offset 30x the noise level (uncalibrated SDR front end): the default build removes the mean in
    the frequency domain (X - mean * FFT(w)), the fallback in the time domain; both must hold 1e-4 on every
    bin, the ones under the removed DC line included."""
    rng = np.random.default_rng(11)
    n = 4096 + 2048 * 63
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n) + (30.0 - 18.0j)).astype(np.complex64)
    plan = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_TUNED)
    plan.set_tuning(variant or None)
    got = plan.exec(x)
    plan.close()
    _, want = R.welch_np(x, fs=1.0, window=hann(4096), nperseg=4096, noverlap=2048, nfft=4096)
    assert relerr(got, want) < RTOL


def _dc_stream(n, ratio, seed):
    """Unit-variance complex noise (sigma = 1 over both components) + a DC offset of |m| = ratio * sigma."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)
    return (x + ratio * np.exp(0.54j)).astype(np.complex64)


DC_BINS = lambda N: [0, 1, N - 1]      # noqa: E731 - the bins a Hann window spreads the removed DC line over


def _f32_mean_bound(ref64, m_abs, N):
    """What ANY float32 detrend may lose in bins k = 0, +-1, per bin: the segment mean m is known to a float32
    computation only to |dm| <= 2 * 2^-23 |m| (representation 2^-24 + the rounding of the sum), and
    FFT((x - m - dm) w)[k] = X[k] - dm W[k] turns that into a relative power error 2 |dm| |W[k]| / |X[k]|.
    ref64 is density-scaled with fs = 1: rms |X[k]| = sqrt(P[k] sum(w^2)).  Hann: W[0] = N/2, W[+-1] = N/4,
    sum(w^2) = 3N/8.  SciPy on complex64 input - the reference's arithmetic - is subject to the same bound (and
    measured beside it)."""
    dm = 2 * 2.0 ** -23 * m_abs
    wk = np.array([N / 2.0, N / 4.0, N / 4.0])
    return 2 * dm * wk / np.sqrt(ref64[DC_BINS(N)] * 3 * N / 8.0)


def _dc_bins_gate(got, ref64, ref32, N, m_abs):
    """-> (worst error outside k = 0, +-1; worst error in k = 0, +-1; the reference's own float32 error there;
    True when every DC bin is inside max(1e-4, float32-mean bound))."""
    dc = DC_BINS(N)
    e = np.abs(got - ref64) / ref64
    e32 = np.abs(ref32 - ref64) / ref64
    ok = bool(np.all(e[dc] <= np.maximum(RTOL, _f32_mean_bound(ref64, m_abs, N))))
    return float(np.delete(e, dc).max()), float(e[dc].max()), float(e32[dc].max()), ok


@pytest.mark.parametrize('N', [2048, 4096, 8192, 16384])
def test_detrend_forms_few_segments_and_large_dc(ctx, hip, N):
    """SciPy's default detrend='constant' (ofdm_cr_tools.py:214,322,342) where single precision is weakest: 1-9 segments
    and a DC line far above the noise, then many segments at 30 / 300 / 3000 sigma.  Compared with the float64 oracle
    on ALL bins, for the two detrend modes of the ABI and the forced forms:
      * OTH_DETREND_CONSTANT ('auto', the default): the PILOT builds - every kernel takes the stream's pilot (the average
        of eight segment means) off each sample as it is loaded, so neither detrend form handles the DC line in float32.  Flat
        1e-4 on ALL bins at every segment count and offset; with many segments every bin at the rounding of the average
        (2e-6) where the reference's own arithmetic (SciPy on GNU Radio's complex64 = float32 throughout,
        oracle.welch_c64, measured beside every case) stands at 3e-4 ... 2.6e-3 in k = 0, +-1.  'td' forces the
        time-domain pilot builds, same gates;
      * OTH_DETREND_CONSTANT_FAST ('fast'): the raw-sample builds.  Every bin but k = 0, +-1 at 1e-4; k = 0, +-1 at
        1e-4 - or, where no float32 detrend of the raw samples can hold that, the float32-mean bound (_f32_mean_bound):
        at one segment and |m| = 35 sigma the reference loses 1e-4 ... 5e-3 there itself.  Fewer than 8 segments per
        stream take the time-domain builds; the FORCED frequency-domain form ('fast-fd') is only recorded below 8
        segments (there it spreads the DC line's rounding over all bins - why the plan does not pick it).  With many
        segments: 1e-4 on all bins up to |m| = 300 sigma, < 1e-3 at 3000 sigma (the bound include/ofdm_tools_hip.h
        states)."""
    step = N // 2
    lines = []
    fmt = 'N=%d |m|=%g sigma nseg=%d %-8s: k=0,+-1 %.2e (reference float32: %.2e), other bins %.2e'
    forms = (('auto', hip.DETREND_CONSTANT, None), ('td', hip.DETREND_CONSTANT, 'td'),
             ('fast', hip.DETREND_CONSTANT_FAST, None), ('fast-fd', hip.DETREND_CONSTANT_FAST, 'fd'))
    for ratio in (3.6, 35.0):
        for nseg in (1, 2, 3, 7, 8, 9):
            x = _dc_stream(N + step * (nseg - 1) + 5, ratio, 1000 * nseg + N)
            _, ref = R.welch_np(x, nperseg=N, nfft=N)
            ref32 = R.welch_c64(x, nperseg=N, nfft=N)
            for name, det, force in forms:
                plan = ctx.welch_plan(N, window=hann(N), detrend=det, kernel=hip.KERNEL_TUNED)
                plan.set_tuning(force)
                got = plan.exec(x)
                assert plan.last_nseg == nseg
                plan.close()
                rest, dc, dc32, ok = _dc_bins_gate(got, ref, ref32, N, ratio)
                lines.append(fmt % (N, ratio, nseg, name, dc, dc32, rest))
                if not name.startswith('fast'):
                    assert rest < RTOL and dc < RTOL, lines[-1]
                elif name != 'fast-fd' or nseg >= 8:
                    assert rest < RTOL and ok, lines[-1]
    nseg = 2047 if N <= 4096 else 511
    measured = {}
    for ratio in (30.0, 300.0, 3000.0):
        x = _dc_stream(N + step * (nseg - 1), ratio, 7 + N)
        _, ref = R.welch_np(x, nperseg=N, nfft=N)
        ref32 = R.welch_c64(x, nperseg=N, nfft=N)
        for name, det, force in forms[:3]:
            plan = ctx.welch_plan(N, window=hann(N), detrend=det, kernel=hip.KERNEL_TUNED)
            plan.set_tuning(force)
            got = plan.exec(x)
            plan.close()
            measured[(name, ratio)] = m = _dc_bins_gate(got, ref, ref32, N, ratio)
            lines.append(fmt % (N, ratio, nseg, name, m[1], m[2], m[0]))
    print('\n'.join(['', 'detrend forms, N = %d' % N] + lines))
    for ratio in (30.0, 300.0, 3000.0):
        for name in ('auto', 'td'):
            # a DC line 70 dB above the signal's total power leaves no trace: every bin at the rounding of the average
            rest, dc, dc32, ok = measured[(name, ratio)]
            assert rest < 2e-6 and dc < 2e-5, (N, ratio, name, measured[(name, ratio)])
        assert measured[('fast', ratio)][3], (N, ratio, measured[('fast', ratio)])
    # the fast form on raw samples: all bins inside 1e-4 up to |m| = 300 sigma; at 3000 sigma the DC line's rounding
    # reaches the other bins (1.5e-4 ... 3.8e-4 measured) - the bound the header states
    assert measured[('fast', 30.0)][0] < RTOL and measured[('fast', 30.0)][1] < 3e-5
    assert measured[('fast', 300.0)][0] < RTOL             # (k = 0, +-1: inside the float32-mean bound, asserted above)
    assert measured[('fast', 3000.0)][0] < 1e-3


def test_csd_few_segments_with_dc(ctx, hip):
    """The two-channel kernels with 1-9 segment pairs under DC lines of 35 / 12 sigma, Pxx, Pyy, Pxy and Cxy against
    the float64 oracle.  OTH_DETREND_CONSTANT (the PILOT builds of the role-split kernel and, forced with 'csd1', of the
    one-role kernel): flat 1e-4 on all bins.  OTH_DETREND_CONSTANT_FAST: the role-split build detrends after the
    transform on raw samples, so 1-7 segment pairs go to the one-role kernel; bins k = 0, +-1 judged by the
    float32-mean bound as in test_detrend_forms_few_segments_and_large_dc."""
    N = 4096
    dc = DC_BINS(N)
    for nseg in (1, 2, 3, 7, 9):
        n = N + 2048 * (nseg - 1) + 3
        x = _dc_stream(n, 35.0, 50 + nseg)
        y = (0.7 * np.roll(x, 5) + _dc_stream(n, 10.0, 90 + nseg) * 0.5).astype(np.complex64)
        my = abs(0.7 * 35.0 + 0.5 * 10.0)
        _, cxy, pxx, pyy, pxy = R.coherence_np(x, y, nperseg=N, nfft=N)
        norm = np.sqrt(pxx * pyy)
        for force in (None, 'csd1'):
            plan = ctx.welch_plan(N, window=hann(N), kernel=hip.KERNEL_TUNED)
            plan.set_tuning(force)
            gxx, gyy, gxy, gc = plan.csd(x, y)
            plan.close()
            assert relerr(gxx, pxx) < RTOL and relerr(gyy, pyy) < RTOL, (force, nseg, relerr(gxx, pxx), relerr(gyy, pyy))
            assert (np.abs(gxy - pxy) / norm).max() < RTOL, (force, nseg)
            if nseg > 1:                                   # one segment: Cxy = 1 identically
                assert np.max(np.abs(gc - cxy)) < RTOL, (force, nseg)
        plan = ctx.welch_plan(N, window=hann(N), detrend=hip.DETREND_CONSTANT_FAST, kernel=hip.KERNEL_TUNED)
        gxx, gyy, gxy, gc = plan.csd(x, y)
        plan.close()
        for got, ref, sig, m_abs in ((gxx, pxx, x, 35.0), (gyy, pyy, y, my)):
            rest, e_dc, e32, ok = _dc_bins_gate(got, ref, R.welch_c64(sig, nperseg=N, nfft=N), N, m_abs)
            assert rest < RTOL and ok, (nseg, rest, e_dc, e32)
        e = np.abs(gxy - pxy) / norm
        bound = np.maximum(RTOL, _f32_mean_bound(pxx, 35.0, N) + _f32_mean_bound(pyy, my, N))
        assert np.delete(e, dc).max() < RTOL and np.all(e[dc] <= bound), (nseg, e.max(), e[dc], bound)
        if nseg > 1:                                   # (0 / 0 under the DC line)
            assert np.max(np.abs(np.delete(gc - cxy, dc))) < RTOL


def test_detrend_pilot_and_fast_build_of_every_kernel(ctx, hip):
    """OTH_DETREND_CONSTANT (the pilot builds; OTH_DETREND_CONSTANT_EXACT is the same mode) over every kernel family that detrends (sizes 64 ... 16384; 50 % overlap with a window
    whose spectrum is confined / is not; other steps; zero-padded segments; the coverage kernel): three device-resident
    streams in one launch with DC lines of 40 sigma, none, and 8 sigma - the pilot is per stream - against the float64
    oracle, flat 1e-4 on ALL bins, and streamed in ragged chunks through accumulate().  Then the same launch with
    OTH_DETREND_CONSTANT_FAST (the builds without the pilot): every bin outside the main lobe of the removed line at
    1e-4 on the 8-sigma and the offset-free stream."""
    import scipy.signal as sg
    ham = lambda n: sg.windows.hamming(n, sym=True).astype(np.float32)      # noqa: E731  1/k sidelobes: time-domain builds
    cases = [(64, 64, 32, hann), (128, 128, 64, hann), (256, 256, 128, hann), (256, 256, 0, hann), (512, 512, 256, ham),
             (1024, 1024, 512, hann), (1024, 1024, 100, hann), (1024, 256, 128, flattop), (1024, 512, 256, hann),
             (2048, 2048, 1024, hann), (2048, 2048, 1024, ham), (2048, 512, 256, flattop), (2048, 1024, 512, hann),
             (4096, 4096, 2048, hann), (4096, 4096, 2048, ham), (4096, 4096, 1000, hann), (4096, 1024, 512, flattop),
             (4096, 512, 256, hann), (4096, 256, 0, hann), (8192, 8192, 4096, hann), (8192, 8192, 4096, ham),
             (8192, 8192, 1000, hann), (8192, 2048, 1024, flattop), (16384, 16384, 8192, hann), (16384, 16384, 8192, ham),
             (16384, 16384, 3000, hann), (16384, 4096, 2048, flattop)]
    ratios = (40.0, 0.0, 8.0)
    for nfft, nperseg, nov, wf in cases:
        w = wf(nperseg)
        step = nperseg - nov
        nseg = 37 if nfft >= 4096 else 150
        n = nperseg + step * (nseg - 1) + 7
        xs = np.stack([_dc_stream(n, r, 31 * nfft + i) for i, r in enumerate(ratios)])
        refs = [R.welch_np(x, window=w.astype(np.float64), nperseg=nperseg, noverlap=nov, nfft=nfft)[1] for x in xs]
        d_in, d_out = ctx.alloc(xs.nbytes), ctx.alloc(3 * nfft * 4)
        try:
            ctx.h2d(d_in, xs)
            for kern in (hip.KERNEL_TUNED, hip.KERNEL_GENERIC):
                kern_eff = hip.KERNEL_AUTO if kern == hip.KERNEL_TUNED and nfft < 256 else kern
                plan = ctx.welch_plan(nfft, nperseg=nperseg, noverlap=nov, window=w, kernel=kern_eff,
                                      detrend=hip.DETREND_CONSTANT_EXACT if nfft == 512 else hip.DETREND_CONSTANT)
                assert plan.exec_dev(d_in, n, d_out, nstreams=3, stream_stride=n) == nseg
                got = ctx.d2h(d_out, (3, nfft), np.float32)
                for i in range(3):
                    assert relerr(got[i], refs[i]) < RTOL, (nfft, nperseg, nov, wf.__name__, kern, i, relerr(got[i], refs[i]))
                fast = ctx.welch_plan(nfft, nperseg=nperseg, noverlap=nov, window=w, kernel=kern_eff,
                                      detrend=hip.DETREND_CONSTANT_FAST)
                assert fast.exec_dev(d_in, n, d_out, nstreams=3, stream_stride=n) == nseg
                got = ctx.d2h(d_out, (3, nfft), np.float32)
                fast.close()
                hw = 5 * nfft // nperseg                 # the removed line's main lobe (a flat-top's: +-5 bins of nperseg)
                lobe = np.r_[0:hw + 1, nfft - hw:nfft]
                for i in (1, 2):
                    e = np.abs(got[i] - refs[i]) / refs[i]
                    assert np.delete(e, lobe).max() < RTOL, (nfft, nperseg, nov, 'fast', kern, i, e.max())
                if kern == hip.KERNEL_TUNED:      # the same stream in ragged chunks: every launch takes its own pilot
                    plan.reset()
                    pos = 0
                    for ln in (nperseg + 3, 5, 3 * step + 1, n):
                        plan.accumulate(xs[0][pos:pos + ln])
                        pos += ln
                    assert relerr(plan.finalize(), refs[0]) < RTOL, (nfft, nperseg, nov, 'chunks')
                plan.close()
        finally:
            ctx.free(d_in)
            ctx.free(d_out)


def test_pilot_under_a_transient_and_a_drifting_offset(ctx, hip):
    """An offset that MOVES within one launch (4096-point Hann, 50 % overlap, 2047 segments; the default plan = the
    frequency-domain build on x - pilot, and the exact time-domain builds beside it), eight noise seeds, offsets moving
    by up to 3000 sigma (an opening segment 70 dB above the signal's total power) and drifts of up to 1200 sigma.
      * The DEFAULT plan: every bin inside the contract's 1e-4, every seed, every case (round 5 held it to 2e-4: one
        pilot per launch left a 600-sigma line to the float32 transform at the ends of a drift.  Round 6: a launch this
        short runs contiguous runs of ~4 segments per workgroup, and each workgroup now spreads its pilot probes over
        its OWN run - welch4096ws.hip - so the pilot follows the drift: 1.05e-4 -> 1.9e-7; measured worst over the eight
        seeds 8.9e-5, on the 3000-sigma transient, where SciPy on complex64 reads up to 1.08e-4).
      * The TIME-DOMAIN builds ('td': what a window with a wide spectrum or a launch of fewer than eight segments
        takes) are gated too (round 5 printed them): the plain 1e-4 on every bin at or above the spectrum's median, and
        on every bin outside 1e-4 an amplitude error of at most 4 ulp of the spectrum's peak amplitude - the single-row criterion
        (check_single_rows), which is the regime here: the segment that holds the transient's edge puts bins +-1 a factor
        5e5 above the median while its own mean removal leaves bin 0 at 0.4 x the median, and ANY float32 transform
        leaves ~1 ulp of the largest amplitude in every bin (measured: 2.3e-4 of bin 0 = 0.85 ulp of the peak; SciPy on
        complex64, the reference's own arithmetic, 9.4e-5 of the same bin) - and the bound tied to that arithmetic,
        max(1e-4, 1.5 x relerr(welch_c64)), on every bin but the one the edge empties (bin 0).
    tools/moving_offset_probe.py prints the table (profiles/r06_moving_offset.txt).
    Few segments (1, 2, 3: below kFdMinSegments the plan takes the time-domain builds, advisor round 4) with an offset
    moving by 100 sigma: the rows are single float32 periodograms of a strong ramp - held to 4 ulp of the row's peak
    amplitude on every bin and to 1.5 x what the reference's own arithmetic (SciPy on complex64: 5e-5 ... 7e-4 on these
    inputs) loses against float64."""
    N, nseg = 4096, 2047
    n = N + (N // 2) * (nseg - 1)
    opening = np.zeros(n)
    opening[:N] = 1.0
    ramp = np.linspace(0.0, 1.0, n)
    worst = {}
    for seed in (5150, 1, 2, 3, 4, 5, 6, 7):
        rng = np.random.default_rng(seed)
        noise = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)
        for name, dc in (('transient 1000', 1000.0 * opening), ('drift 400', 400.0 * ramp),
                         ('transient 3000', 3000.0 * opening), ('drift 1200', 1200.0 * ramp)):
            x = (noise + dc * np.exp(0.54j)).astype(np.complex64)
            _, ref = R.welch_np(x, nperseg=N, nfft=N)
            c64 = relerr(R.welch_c64(x, nperseg=N, nfft=N), ref)      # what the reference's own float32 arithmetic loses
            for force in (None, 'td'):
                plan = ctx.welch_plan(N, window=hann(N), kernel=hip.KERNEL_TUNED)
                plan.set_tuning(force)
                plan.set_schedule(hip.SCHED_CONTIGUOUS)      # fixed summation order: the same digits on every run
                got = plan.exec(x).astype(np.float64)
                plan.close()
                rel = np.abs(got - ref) / ref
                worst[(name, force or 'auto')] = max(worst.get((name, force or 'auto'), 0.0), float(rel.max()))
                if force is None:
                    assert rel.max() < RTOL, (seed, name, rel.max())
                    continue
                amp = np.abs(np.sqrt(got) - np.sqrt(ref)) / np.sqrt(ref.max())
                weak = rel >= RTOL      # bins outside the plain 1e-4: within 4 ulp of the peak amplitude, and below the median
                assert np.all(amp[weak] <= 4 * 2.0 ** -23), (seed, name, amp[weak].max() * 2.0 ** 23)
                assert rel[ref >= np.median(ref)].max() < RTOL, (seed, name)
                assert rel[1:].max() < max(RTOL, 1.5 * c64), (seed, name, float(rel[1:].max()), c64)
    print('moving offsets, worst of eight seeds: ' + ', '.join('%s %s %.1e' % (k[0], k[1], v) for k, v in sorted(worst.items())))
    # one, two, three segments under an offset that moves by 100 sigma within the launch
    rng = np.random.default_rng(77)
    for nfft in (2048, 4096, 16384):
        for ns in (1, 2, 3):
            m = nfft + (nfft // 2) * (ns - 1)
            x = ((rng.standard_normal(m) + 1j * rng.standard_normal(m)) * np.sqrt(0.5)
                 + np.linspace(0.0, 100.0, m) * np.exp(-1.1j)).astype(np.complex64)
            _, ref = R.welch_np(x, nperseg=nfft, nfft=nfft)
            c64 = R.welch_c64(x, nperseg=nfft, nfft=nfft)      # the reference's own arithmetic: 5e-5 ... 7e-4 here
            plan = ctx.welch_plan(nfft, window=hann(nfft))
            got = plan.exec(x).astype(np.float64)
            plan.close()
            # one to three rows that hold a +-50-sigma ramp are SINGLE float32 periodograms of a strong component: every
            # bin carries about an ulp of the row's largest amplitude whatever the detrend form (check_single_rows)
            amp = np.abs(np.sqrt(got) - np.sqrt(ref)) / np.sqrt(ref.max())
            assert amp.max() <= 4 * 2.0 ** -23, (nfft, ns, amp.max() * 2.0 ** 23)
            assert relerr(got, ref) < max(RTOL, 1.5 * relerr(c64, ref)), (nfft, ns, relerr(got, ref), relerr(c64, ref))


def test_welch4096_window_with_wide_spectrum_takes_the_time_domain_detrend(ctx, hip):
    """A symmetric Hamming window has 1/k sidelobes: its spectrum is not confined to the bins the
    frequency-domain detrend corrects, so the plan must fall back to the kernel that subtracts the
    mean before windowing - visible as parity with a strong DC offset in the input."""
    rng = np.random.default_rng(12)
    n = 4096 + 2048 * 40
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n) + (5.0 + 2.0j)).astype(np.complex64)
    w = (0.54 - 0.46 * np.cos(2 * np.pi * np.arange(4096) / 4095.0)).astype(np.float32)
    plan = ctx.welch_plan(4096, window=w, kernel=hip.KERNEL_TUNED)
    got = plan.exec(x)
    plan.close()
    _, want = R.welch_np(x, fs=1.0, window=w, nperseg=4096, noverlap=2048, nfft=4096)
    assert relerr(got, want) < RTOL


def test_tuned_kernels_repeat_without_drift(ctx, hip):
    """300 launches of the tuned builds on one input, schedules and chunk sizes drawn at random: every result
    must stay within rounding of the generic kernel's (a missed barrier or a stale ticket shows up as an
    occasional outlier, not as a steady error)."""
    rng = np.random.default_rng(99)
    n = 4096 + 2048 * 2046 + 777
    d_in = ctx.alloc(2 * n * 8)
    d_a, d_b = ctx.alloc(2 * 4096 * 4), ctx.alloc(2 * 4096 * 4)
    try:
        ctx.synth_iq(d_in, 2 * n, 5, R.TONES, R.DC)
        tuned = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_TUNED)
        gen = ctx.welch_plan(4096, window=hann(4096), kernel=hip.KERNEL_GENERIC)
        gen.exec_dev(d_in, n, d_b, nstreams=2, stream_stride=n)
        want = ctx.d2h(d_b, (2, 4096), np.float32).astype(np.float64)
        worst = 0.0
        for it in range(300):
            tuned.set_tuning(('ws', 'pipe', 'dpp')[it % 3], chunk=int(rng.choice([1, 2, 3, 5, 8, 16, 32, 64])))
            tuned.set_schedule(int(rng.integers(0, 3)))
            assert tuned.exec_dev(d_in, n, d_a, nstreams=2, stream_stride=n) == 2047
            got = ctx.d2h(d_a, (2, 4096), np.float32)
            err = float(np.max(np.abs(got - want) / want))
            worst = max(worst, err)
            assert err < 2e-5, (it, err)
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


# ------------------------------ reference-tagged fixtures (tests/golden/ref_*.npz) ----
# Outputs of the reference's OWN function bodies (tests/golden/ref_extract.py, build container); the HIP
# path is called through the product's ofdm_cr_tools API, which has the reference's names and arguments.

def test_ref_a6_welch_plot_db_and_power_estimate(ctx, golden):
    """ofdm_cr_tools.py:321-326, :341-345, :149-153 on the device vs the reference's own output."""
    from ofdm_tools import ofdm_cr_tools as T
    g = golden('ref_welch_hann_4096.npz')
    x = golden(str(g['input_from']))['x']
    Sf, fc, nfft = int(g['Sf']), float(g['fc']), int(g['nfft'])
    axis, db = T.welch_plot_dB(x, Sf, fc, nfft, ctx=ctx)
    assert np.allclose(axis, g['expected_axis'], rtol=0, atol=1e-3)
    assert relerr(10 ** (np.array(db) / 10), 10 ** (g['expected_db'] / 10)) < RTOL
    for fs, key in ((Sf, 'expected_power'), (1.0, 'expected_power_fs1')):
        assert abs(T.welch_power_estimate(x, nfft, fs, ctx=ctx) - float(g[key])) < RTOL * float(g[key])
    want = float(g['expected_clc_power_freq'])
    assert abs(T.clc_power_freq(x[:4096], 4096, Sf, ctx=ctx) - want) < RTOL * want


def test_ref_a6_a14_src_power_welch_and_fast_spectrum_scan(ctx, golden):
    """ofdm_cr_tools.py:213-230 and :471-537 (method 'welch'), noise estimate carried over three scans."""
    from ofdm_tools import ofdm_cr_tools as T
    g = golden('ref_src_power_welch_2048.npz')
    x = golden(str(g['input_from']))['x']
    Sf, N, cs, sbw = int(g['Sf']), int(g['nfft']), float(g['channel_rate']), float(g['srch_bw'])
    Fr = float(Sf) / N
    bb = T.frange(-Sf // 2, Sf // 2, cs)
    assert np.array_equal(bb, g['bb_freqs'])
    psd, ax, plc = T.src_power_welch(x, len(x), N, Fr, Sf, bb, sbw / Fr, ctx=ctx)
    assert relerr(psd, g['expected_psd']) < RTOL and np.allclose(ax, g['expected_axis'])
    assert relerr(plc, g['expected_plc']) < RTOL
    ne = float(g['scan_noise0'])
    for i, (lo, hi) in enumerate(g['scan_ranges']):
        thr, plc, ne, occ = T.fast_spectrum_scan(x[lo:hi], float(g['scan_fc']), cs, sbw, N, Sf, 'welch',
                                                 int(g['scan_thr_leveler']), ne, float(g['scan_alpha']), ctx=ctx)
        assert abs(thr - g['scan_thr'][i]) < RTOL * g['scan_thr'][i]
        assert abs(ne - g['scan_noise'][i]) < RTOL * g['scan_noise'][i]
        assert relerr(plc, g['scan_plc'][i]) < RTOL
        assert [1.0 if a in occ else 0.0 for a in g['ax_ch']] == list(g['scan_occupied'][i])


def test_ref_a4_sweeper_src_power(ctx, golden):
    """spectrum_sweeper.py:260-276 through the product's spectrum_sweeper block (flattop, nperseg = nfft/4
    zero-padded, shift, trim, dB in one plan) vs the reference's own output."""
    import importlib
    SW = importlib.import_module('ofdm_tools.spectrum_sweeper')      # the package re-exports the class under this name
    g = golden('ref_sweeper_src_power.npz')
    x = golden(str(g['input_from']))['x']
    nfft, ex, fs = int(g['nfft']), int(g['excess_bins']), float(g['fs'])

    class Rx(object):
        def set_center_freq(self, f, chan):
            pass
    # trunc_sample_rate chosen so that floor((Sf - trunc)/2 / (Sf/nfft)) = excess_bins (:69-70)
    blk = SW.spectrum_sweeper(Rx(), 'osmosdr', nfft, fs, fs - 2 * ex * fs / nfft, 100e6, 110e6, 10, 0.0, 1.0, 0,
                              1472, ctx=ctx, threaded=False)
    assert blk.excess_bins == ex
    db = blk._src_power(x)
    assert db.shape == g['expected_db'].shape
    assert relerr(10 ** (db.astype(np.float64) / 10), 10 ** (g['expected_db'] / 10)) < RTOL
    assert np.max(np.abs(db - g['expected_db'])) < 1e-3
    assert np.array_equal(SW.frange(88.0e6 + 1.0e6, 108.0e6, 2.0e6), g['frange_le_a'])
    assert np.array_equal(SW.frange(0.0, 1.0, 0.1), g['frange_le_c'])
    nt = ctx.welch_plan(1024, nperseg=256, window=flattop(256), fs=250000.0, fftshift=True, db=True).exec(x[:9000])
    assert relerr(10 ** (nt.astype(np.float64) / 10), 10 ** (g['expected_db_notrim'] / 10)) < RTOL


def test_ref_a7_src_power_movingaverage(ctx, golden):
    """ofdm_cr_tools.py:168-170, :232-249 on the device vs the reference's own output on float32 PSD rows
    (what the watcher threads hand to src_power, spectrum_sensor_v2.py:414)."""
    from ofdm_tools import ofdm_cr_tools as T
    g = golden('ref_src_power_cases.npz')
    c = golden(str(g['input_from']))
    assert np.array_equal(T.frange(-500000.0, 500000.0, 25e3), g['frange_c'])
    for i in range(int(g['n'])):
        Sf, N = int(c['Sf_%d' % i]), int(c['N_%d' % i])
        cs, sbw = float(c['cs_%d' % i]), float(c['sbw_%d' % i])
        Fr = float(Sf) / N
        psd = c['psd_%d' % i].astype(np.float32)
        bb = T.frange(-Sf // 2, Sf // 2, cs)
        assert np.array_equal(bb, g['bb_%d' % i])
        got = T.src_power(psd, N, Fr, Sf, bb, sbw / Fr, ctx=ctx)
        assert len(got) == len(g['plc_f32_%d' % i]) and np.allclose(got, g['plc_f32_%d' % i], rtol=1e-5)
        assert np.allclose(T.movingaverage(psd, sbw / Fr, ctx=ctx), g['ma_%d' % i], rtol=1e-5)


def test_ref_a8_a9_a11_host_state_machines(ctx, golden):
    """The block-side state machines (ChannelScanner EMA + top-4, coherence_detector decision) fed by device
    channel powers vs the reference's own methods (spectrum_sensor_v2.py:533-544, :228-237;
    multichannel_scanner.py:214-239; coherence_detector.py:254-278)."""
    from ofdm_tools import scanner as S
    g = golden('ref_scanner_seq.npz')
    c = golden(str(g['input_from']))
    st = S.ChannelScanner(1024, 1000000, 25e3, 12.5e3, tune_freq=100000000, trunc_band=800000, thr_leveler=4,
                          alpha_avg=0.5, ctx=ctx)
    for i, r in enumerate(c['rows']):
        st.basic_scan(r.astype(np.float32))
        assert np.allclose(st.plc, g['plc_seq'][i], rtol=1e-5)
    subj = [float(v) for v in c['subject_channels']]
    pwr, top = S.top4(st.plc, st.subject_index(subj), subj)
    assert np.allclose(pwr, g['subject_pwr'], atol=1e-4) and top == list(g['top4'])
    # a8 in full against the reference's own stats_watcher.spectrum_scanner (:445-479, ref_threads.npz): EMA, max
    # hold, noise estimate, threshold decision per row
    t = golden('ref_threads.npz')
    st = S.ChannelScanner(1024, 1000000, 25e3, 12.5e3, tune_freq=100000000, trunc_band=800000, thr_leveler=4,
                          alpha_avg=0.5, ctx=ctx)
    for i, r in enumerate(c['rows']):
        occ = st.scan(r.astype(np.float32))
        assert np.allclose(st.plc, t['stats_plc_seq'][i], rtol=1e-5)
        assert np.isclose(st.noise_estimate, t['stats_noise_seq'][i], rtol=1e-5)
        assert [1.0 if a in occ else 0.0 for a in st.ax_ch] == list(t['stats_occupied_seq'][i])
    assert np.allclose(st.cumulative_max_power, t['stats_cumulative_max'], rtol=1e-5)
    assert np.allclose(st.periodic_max_power, t['stats_periodic_max'], rtol=1e-5)


# ------------------------------------------- device / partial forms added in ABI 2 ----

def test_csd_partial_sums_of_time_chunks_add_up_and_device_form(ctx, hip, golden):
    """raw sums (sum|X|^2, sum|Y|^2, sum conj(X)Y) of halo-overlapped time chunks add to the
    whole; oth_csd_scale_dev turns them into Pxx, Pyy, Pxy, Cxy; oth_csd_exec_dev equals oth_csd_exec."""
    from ofdm_tools import sweep
    g = golden('coherence_csd_4096.npz')
    x, y = g['x'], g['y']
    nfft, step, n = 4096, 2048, len(g['x'])
    plan = ctx.welch_plan(nfft, window=hann(nfft), fs=1.0, fftshift=True, trim_bins=64)
    want = plan.csd(x, y)
    dx, dy = ctx.alloc(x.nbytes), ctx.alloc(y.nbytes)
    d_sum, d_o = ctx.alloc(4 * nfft * 4), ctx.alloc(5 * nfft * 4)
    m = plan.out_len
    try:
        ctx.h2d(dx, x)
        ctx.h2d(dy, y)
        tot = np.zeros(4 * nfft, np.float64)
        nseg = 0
        for r in range(3):                               # three "ranks"
            first, cnt, s0, k = sweep.time_shard(n, nfft, step, r, 3)
            assert plan.csd_partial_dev(dx + 8 * first, dy + 8 * first, cnt, d_sum) == k
            tot += ctx.d2h(d_sum, (4 * nfft,), np.float32)
            nseg += k
        assert nseg == 31
        ctx.h2d(d_sum, tot.astype(np.float32))
        plan.csd_scale_dev(d_sum, nseg, d_o, d_o + 4 * nfft, d_o + 8 * nfft, d_o + 16 * nfft)
        got = [ctx.d2h(d_o, (m,), np.float32), ctx.d2h(d_o + 4 * nfft, (m,), np.float32),
               ctx.d2h(d_o + 8 * nfft, (2 * m,), np.float32).view(np.complex64), ctx.d2h(d_o + 16 * nfft, (m,), np.float32)]
        assert relerr(got[0], want[0]) < 2e-6 and relerr(got[1], want[1]) < 2e-6
        assert np.max(np.abs(got[2] - want[2]) / np.sqrt(want[0] * want[1])) < 2e-6
        assert np.max(np.abs(got[3] - want[3])) < 1e-5
        # asynchronous device form, only Cxy requested
        assert plan.csd_exec_dev(dx, dy, n, cxy=d_o) == 31
        assert np.max(np.abs(ctx.d2h(d_o, (m,), np.float32) - want[3])) < 1e-6
        e = np.fft.fftshift(g['expected_cxy'])[64:-64]
        assert np.max(np.abs(want[3] - e)) < RTOL
    finally:
        for p in (dx, dy, d_sum, d_o):
            ctx.free(p)


def test_chain_push_dev_matches_host_push_across_ragged_chunks(ctx, hip, golden):
    """oth_chain_push_dev (device in, device rows out, asynchronous): same rows as the host form when the stream
    arrives in chunks that split vectors, keep_one_in_n = 3, IIR + log epilogue."""
    g = golden('gr_chain_bh_iir_log_2048.npz')
    N, x = 2048, g['x']
    k = -10 * np.log10(N) - 10 * np.log10(int(g['sample_rate']))
    a = ctx.chain(N, g['window'], True, hip.EPI_MAG2, 3)
    b = ctx.chain(N, g['window'], True, hip.EPI_MAG2, 3)
    for ch in (a, b):
        ch.set_iir_log(float(g['average']), k)
    d_x, d_rows = ctx.alloc(x.nbytes), ctx.alloc(16 * N * 4)
    try:
        ctx.h2d(d_x, x)
        for lo, hi_ in ((0, 3000), (3000, 3001), (3001, 20000), (20000, 20480), (20480, len(x))):
            rows, n = a.push(x[lo:hi_])
            nd = b.push_dev(d_x + 8 * lo, hi_ - lo, d_rows, 16)
            assert nd == n
            if n:
                got = ctx.d2h(d_rows, (n, N), np.float32)
                assert np.array_equal(got, rows)
        assert np.array_equal(a.iir(), b.iir())
    finally:
        ctx.free(d_x)
        ctx.free(d_rows)
    ref = R.chain_local_worker(x, N, int(g['sample_rate']), float(g['average']), decim=3)[0]
    assert relerr(a.iir(), ref[-1]) < RTOL


def test_scan_decide_dev_on_device_rows(ctx, hip):
    """BASELINE config 5 decision stage on the rows oth_welch_exec_dev left in HBM: mask / noise floor equal
    oth_bin_threshold, channel sums equal src_power (the oracle's, ofdm_cr_tools.py:232-249) per stream."""
    from ofdm_tools.scan_batch import BatchScanPlan
    N, ns, n = 16384, 5, 16384 * 6
    Sf = 1000000
    bp = BatchScanPlan(ctx, N, Sf, 15625.0, 10e3, thr_leveler=3.0)
    xs = [R.synth_iq(n, 3000 + i) for i in range(ns)]
    d_in, d_rows = ctx.alloc(ns * n * 8), ctx.alloc(ns * N * 4)
    try:
        ctx.h2d(d_in, np.concatenate(xs))
        assert bp.psd_rows_dev(d_in, n, ns, n, d_rows) == 6
        mask, noise, plc = bp.decide_dev(d_rows, ns)
        rows = ctx.d2h(d_rows, (ns, N), np.float32)
    finally:
        ctx.free(d_in)
        ctx.free(d_rows)
    mask2, noise2 = ctx.bin_threshold(rows, bp.scanner.srch_bins, 3.0)
    assert np.array_equal(mask, mask2) and np.array_equal(noise, noise2)
    st = R.ScannerState(N, Sf, 15625.0, 10e3, trunc_band=Sf)
    for i in range(ns):
        want = R.src_power(rows[i], N, st.Fr, Sf, st.bb_freqs, st.srch_bins)
        assert plc.shape == (ns, len(want)) and np.allclose(plc[i], want, rtol=1e-5)
        ma = R.movingaverage(rows[i], st.srch_bins)
        assert np.isclose(noise[i], ma.min(), rtol=1e-5)
        assert np.array_equal(mask[i], (rows[i] > np.float32(3.0) * noise[i]).astype(np.uint8))
    # host-row convenience form goes through the same entry point
    m3, n3, p3 = bp.decide(rows)
    assert np.array_equal(m3, mask) and np.array_equal(p3, plc)
    # rows with carriers 100 / 130 / 150 dB above a floor that sits right next to them (a float32 PSD row spans up to
    # ~140 dB): the sliding sum of the moving average must not carry the carrier's rounding into the floor's minimum
    # (advisor, round 4) - noise floor and channel sums against the oracle's direct np.convolve
    rng = np.random.default_rng(9)
    hard = np.empty((3, N), np.float32)
    for i, db in enumerate((100.0, 130.0, 150.0)):
        row = (1e-12 * (1.0 + 0.1 * rng.random(N))).astype(np.float32)
        for k in (700, 701, 702, 5000, 9000, 9163, 16000):      # single bins and a pair exactly one window apart
            row[k] = np.float32(1e-12 * 10.0 ** (db / 10.0))
        hard[i] = row
    m4, n4, p4 = bp.decide(hard)
    for i in range(3):
        ma = R.movingaverage(hard[i], st.srch_bins)
        assert np.isclose(n4[i], ma.min(), rtol=2e-6), (i, n4[i], ma.min())
        want = R.src_power(hard[i], N, st.Fr, Sf, st.bb_freqs, st.srch_bins)
        assert np.allclose(p4[i], want, rtol=1e-5)
        assert np.array_equal(m4[i], (hard[i] > np.float32(3.0) * n4[i]).astype(np.uint8))


# ------------------------------------------- segfft.hip: 1024 / 2048 Welch, fused chain ----

@pytest.mark.parametrize('nfft', [256, 512, 1024, 2048])
@pytest.mark.parametrize('build', ['segws', 'seg3', 'seg4'])
def test_seg_welch_vs_oracle_and_generic(ctx, hip, nfft, build):
    """The team-per-segment kernel (wave-per-segment at 1024) against the float64 oracle and the coverage kernel:
    50 % overlap (kept half in registers) and other steps, detrend on / off, a DC offset 30x the noise, segment
    counts around chunk and grid multiples, three streams, all schedules."""
    rng = np.random.default_rng(nfft)
    x = R.synth_iq(nfft * 40 + 77, 500 + nfft)
    for nov, det in ((nfft // 2, True), (nfft // 2, False), (0, True), (nfft - 1, True), (nfft // 4, False)):
        _, ref = R.welch_np(x[:nfft * 12 + 5], fs=3.0, nperseg=nfft, noverlap=nov, nfft=nfft,
                            detrend='constant' if det else False)
        plan = ctx.welch_plan(nfft, noverlap=nov, window=hann(nfft), fs=3.0,
                              detrend=hip.DETREND_CONSTANT if det else hip.DETREND_NONE, kernel=hip.KERNEL_TUNED)
        plan.set_tuning(build)
        assert relerr(plan.exec(x[:nfft * 12 + 5]), ref) < RTOL, (nov, det)
        plan.close()
    xdc = (rng.standard_normal(nfft * 30) + 1j * rng.standard_normal(nfft * 30) + (30.0 - 18.0j)).astype(np.complex64)
    _, ref = R.welch_np(xdc, nperseg=nfft, nfft=nfft)
    plan = ctx.welch_plan(nfft, window=hann(nfft), kernel=hip.KERNEL_TUNED)
    plan.set_tuning(build)
    assert relerr(plan.exec(xdc), ref) < RTOL
    # device-resident, many segments, 1-3 streams, schedules and chunk sizes vs the coverage kernel
    step = nfft // 2
    nmax = nfft + step * 20000
    d_in = ctx.alloc(3 * nmax * 8)
    d_a, d_b = ctx.alloc(3 * nfft * 4), ctx.alloc(3 * nfft * 4)
    try:
        ctx.synth_iq(d_in, 3 * nmax, 31, R.TONES, R.DC)
        gen = ctx.welch_plan(nfft, window=hann(nfft), kernel=hip.KERNEL_GENERIC)
        for nseg in [1, 2, 3, 15, 16, 17, 255, 4095, 4097, 16383, 20000] + [int(v) for v in rng.integers(1, 20000, 5)]:
            n = nfft + step * (nseg - 1) + int(rng.integers(0, step))
            ns = int(rng.integers(1, 4))
            plan.set_schedule(int(rng.integers(0, 3)))
            plan.set_tuning(build, chunk=int(rng.integers(0, 6)))
            assert plan.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
            assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
            a = ctx.d2h(d_a, (ns, nfft), np.float32)
            b = ctx.d2h(d_b, (ns, nfft), np.float32)
            err = np.max(np.abs(a.astype(np.float64) - b) / np.maximum(b, 0.1 * np.median(b)))
            assert err < 5e-5, (nseg, ns, err)
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


@pytest.mark.parametrize('nfft,frac', [(1024, 4), (1024, 2), (2048, 4), (2048, 2), (8192, 4), (16384, 4)])
def test_seg_zero_padded_segments_vs_oracle_and_generic(ctx, hip, nfft, frac):
    """nperseg = nfft / 4 zero-padded to nfft - the sweeper's `_src_power` call for the fft_len a flowgraph passes
    (spectrum_sweeper.py:263: welch(flattop, nperseg=nFFT/4.0, nfft=nFFT)) - and nperseg = nfft / 2, on the
    team-per-segment kernel (1024 / 2048) and the workgroup-per-segment kernel (8192 / 16384, nfft / 4 only) with
    compile-time zero rows: against the float64 oracle (flattop, detrend, 50 % overlap of
    nperseg, fftshift + dB as the sweeper does; other steps; a DC offset 30 x the noise), then device-resident against
    the coverage kernel over segment counts around chunk and grid multiples."""
    nps = nfft // frac
    rng = np.random.default_rng(nfft + frac)
    x = R.synth_iq(nps * 40 + 77, 600 + nfft)
    for nov, det in ((nps // 2, True), (nps // 2, False), (0, True), (nps - 1, True), (nps // 4, False)):
        _, ref = R.welch_np(x[:nps * 12 + 5], fs=2.0e6, window='flattop', nperseg=nps, noverlap=nov, nfft=nfft,
                            detrend='constant' if det else False)
        plan = ctx.welch_plan(nfft, nperseg=nps, noverlap=nov, window=flattop(nps), fs=2.0e6,
                              detrend=hip.DETREND_CONSTANT if det else hip.DETREND_NONE, kernel=hip.KERNEL_TUNED)
        assert relerr(plan.exec(x[:nps * 12 + 5]), ref) < RTOL, (nov, det)
        plan.close()
    # the sweeper's own form: shift, trim, dB
    ex = nfft // 16
    want = R.sweeper_src_power(x, nfft, 2.0e6, ex) if frac == 4 else None
    if want is not None:
        plan = ctx.welch_plan(nfft, nperseg=nps, window=flattop(nps), fs=2.0e6, fftshift=True, trim_bins=ex, db=True,
                              kernel=hip.KERNEL_TUNED)
        got = plan.exec(x)
        assert relerr(10 ** (got.astype(np.float64) / 10), 10 ** (np.asarray(want) / 10)) < RTOL
    xdc = (rng.standard_normal(nps * 30) + 1j * rng.standard_normal(nps * 30) + (30.0 - 18.0j)).astype(np.complex64)
    _, ref = R.welch_np(xdc, window='flattop', nperseg=nps, nfft=nfft)
    plan = ctx.welch_plan(nfft, nperseg=nps, window=flattop(nps), kernel=hip.KERNEL_TUNED)
    assert relerr(plan.exec(xdc), ref) < RTOL
    step = nps // 2
    nmax = nps + step * 20000
    d_in = ctx.alloc(3 * nmax * 8)
    d_a, d_b = ctx.alloc(3 * nfft * 4), ctx.alloc(3 * nfft * 4)
    try:
        ctx.synth_iq(d_in, 3 * nmax, 31, R.TONES, R.DC)
        gen = ctx.welch_plan(nfft, nperseg=nps, window=flattop(nps), kernel=hip.KERNEL_GENERIC)
        for nseg in [1, 2, 3, 31, 32, 33, 255, 4095, 4097, 16383, 20000] + [int(v) for v in rng.integers(1, 20000, 5)]:
            n = nps + step * (nseg - 1) + int(rng.integers(0, step))
            ns = int(rng.integers(1, 4))
            plan.set_schedule(int(rng.integers(0, 3)))
            plan.set_tuning(None, chunk=int(rng.integers(0, 6)))
            assert plan.exec_dev(d_in, n, d_a, nstreams=ns, stream_stride=nmax) == nseg
            assert gen.exec_dev(d_in, n, d_b, nstreams=ns, stream_stride=nmax) == nseg
            a = ctx.d2h(d_a, (ns, nfft), np.float32)
            b = ctx.d2h(d_b, (ns, nfft), np.float32)
            err = np.max(np.abs(a.astype(np.float64) - b) / np.maximum(b, 0.1 * np.median(b)))
            assert err < 5e-5, (nseg, ns, err)
    finally:
        for ptr in (d_in, d_a, d_b):
            ctx.free(ptr)


@pytest.mark.parametrize('nfft', [256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fused_chain_many_rows_iir_peak_and_plain(ctx, hip, nfft):
    """The fused periodogram chain over thousands of kept vectors in one launch (IIR as a weighted sum over the
    launch, peak hold as a max) against the sequential oracle, against the coverage kernels, across pushes that
    split vectors, with keep_one_in_n = 3.  256 ... 4096: segfft.hip's team-per-segment build; 8192 / 16384 (the
    scanner's size in BASELINE config 5): welch16k.hip's workgroup-per-segment build, rectangular (no window
    registers, prefetching) and windowed."""
    from ofdm_tools import windows
    rows_n, keep = (700, 3) if nfft <= 4096 else (400, 3)
    x = R.synth_iq(nfft * rows_n * keep + 123, 900 + nfft)
    w = windows.blackmanharris(nfft)
    k = -10 * np.log10(nfft) - 10 * np.log10(2.0e6)
    lin, db = R.chain_local_worker(x, nfft, 2000000, 0.05, decim=keep)          # slow IIR: hundreds of rows matter
    assert len(db) == rows_n
    for kern in (hip.KERNEL_AUTO, hip.KERNEL_GENERIC):
        ch = ctx.chain(nfft, w, True, hip.EPI_MAG2, keep)
        ch.set_kernel(kern)
        ch.set_iir_log(0.05, k)
        cut = nfft * 1000 + 17
        r1, n1 = ch.push(x[:cut], max_rows=2)
        r2, n2 = ch.push(x[cut:], max_rows=5)
        assert n1 + n2 == rows_n and len(r2) == 5
        assert relerr(10 ** ((r2.astype(np.float64) - k) / 10), lin[-5:]) < RTOL, kern
        assert relerr(10 ** ((r1.astype(np.float64) - k) / 10), lin[n1 - 2:n1]) < RTOL, kern
        assert relerr(ch.iir(), lin[-1]) < RTOL
    # alpha = 1: the filter forgets everything but the last row
    ch = ctx.chain(nfft, w, True, hip.EPI_MAG2, keep)
    ch.set_iir_log(1.0, 0.0)
    r, n = ch.push(x, max_rows=1)
    last = R.chain_local_worker(x, nfft, 2000000, 1.0, decim=keep)[0][-1]
    # through float32 dB: one ulp of a 66 dB value is 1.7e-6 of the power, 7 ulp of the amplitude
    check_single_rows(10 ** (r[0].astype(np.float64) / 10), last, ulps=16)
    # peak hold on |X| (psd_logger): natural order
    mag, peak = R.chain_psd_logger(x[:nfft * 300], nfft, decim=2)
    ch = ctx.chain(nfft, w, False, hip.EPI_MAG, 2)
    ch.set_peak_hold(True)
    r1, n1 = ch.push(x[:nfft * 100 + 5], max_rows=1)
    assert relerr(ch.peak(), peak[n1 - 1]) < RTOL
    check_single_rows(r1[0], mag[n1 - 1], power=False)
    r2, n2 = ch.push(x[nfft * 100 + 5:nfft * 300], max_rows=3)
    assert n1 + n2 == len(mag) and relerr(ch.peak(), peak[-1]) < RTOL
    check_single_rows(r2, mag[-3:], power=False)
    # plain rows (spectrum_sensor_v2): all rows handed back, and the latest-wins form
    ref = R.chain_sensor_v2(x[:nfft * 64], nfft)
    ch = ctx.chain(nfft, None, True, hip.EPI_MAG2_OVER_N2, 1)
    rows, n = ch.push(x[:nfft * 64])
    assert n == 64
    check_single_rows(rows, ref)
    ch.reset()
    rows, n = ch.push(x[:nfft * 64], max_rows=1)
    assert n == 64
    check_single_rows(rows[0], ref[-1])


@pytest.mark.parametrize('nfft', [256, 4096])
def test_fused_chain_full_size_push_iir_and_peak(ctx, hip, nfft):
    """One push of 2^26 device-resident samples (the size profiles/r03_chain*.txt are quoted on; the cross-team
    reduction runs in its two-launch form over ~12 000 / 768 team rows) in IIR and in peak mode.
    IIR (alpha = 0.8): the state after the push depends on the last rows only ((1-alpha)^48 = 3e-34), so the oracle
    runs on the last 48 vectors; a second chain that takes the same samples in two pushes must agree.
    Peak: the oracle on a 2^20-sample prefix pushed first, then the whole stream against the coverage kernels
    (transform + sequential row epilogue), which share nothing with the fused launch but the input."""
    from ofdm_tools import windows
    n = 1 << 26
    w = windows.blackmanharris(nfft)
    d = ctx.alloc(n * 8)
    try:
        ctx.synth_iq(d, n, 77, R.TONES, R.DC)
        tail = ctx.d2h(d + (n - 48 * nfft) * 8, (48 * nfft,), np.complex64)
        lin, _ = R.chain_local_worker(tail, nfft, 2000000, 0.8)
        ch = ctx.chain(nfft, w, True, hip.EPI_MAG2, 1)
        ch.set_iir_log(0.8, 0.0)
        assert ch.push_dev(d, n) == n // nfft
        assert relerr(ch.iir(), lin[-1]) < RTOL
        ch2 = ctx.chain(nfft, w, True, hip.EPI_MAG2, 1)
        ch2.set_iir_log(0.8, 0.0)
        cut = (n // 3) + 5
        assert ch2.push_dev(d, cut) + ch2.push_dev(d + cut * 8, n - cut) == n // nfft
        assert relerr(ch2.iir(), ch.iir()) < 1e-5
        # peak hold on |X|
        npre = 1 << 20
        mag, peak = R.chain_psd_logger(ctx.d2h(d, (npre,), np.complex64), nfft)
        pk = ctx.chain(nfft, w, False, hip.EPI_MAG, 1)
        pk.set_peak_hold(True)
        assert pk.push_dev(d, npre) == npre // nfft
        assert relerr(pk.peak(), peak[-1]) < RTOL
        assert pk.push_dev(d + npre * 8, n - npre) == (n - npre) // nfft
        gen = ctx.chain(nfft, w, False, hip.EPI_MAG, 1)
        gen.set_kernel(hip.KERNEL_GENERIC)
        gen.set_peak_hold(True)
        assert gen.push_dev(d, n) == n // nfft
        a, b = pk.peak().astype(np.float64), gen.peak().astype(np.float64)
        assert np.all(a >= peak[-1] * (1 - 1e-6)) and relerr(a, b) < 2e-5
    finally:
        ctx.free(d)


@pytest.mark.parametrize('build', ['', 'csd1'])
def test_csd_tuned_vs_generic_on_awkward_segment_counts(ctx, hip, build):
    """The two-channel kernels (wave-specialised pairs with the lane-half spectrum exchange; the one-role build)
    against the coverage kernel on device-resident streams: segment counts around chunk and grid multiples,
    one- and two-segment chunks, all schedules, detrend on and off."""
    rng = np.random.default_rng(17)
    nmax = 4096 + 2048 * 5000
    dx, dy = ctx.alloc(nmax * 8), ctx.alloc(nmax * 8)
    outs = [ctx.alloc(5 * 4096 * 4) for _ in range(2)]
    try:
        ctx.synth_iq(dx, nmax, 61, R.TONES, R.DC)
        ctx.synth_iq(dy, nmax, 62, ((0.5, 0.1234), (1.0, 0.2)), 0.3 - 0.2j)
        for det in (hip.DETREND_CONSTANT, hip.DETREND_NONE):
            tuned = ctx.welch_plan(4096, window=hann(4096), detrend=det, kernel=hip.KERNEL_TUNED)
            gen = ctx.welch_plan(4096, window=hann(4096), detrend=det, kernel=hip.KERNEL_GENERIC)
            for nseg in [1, 2, 3, 7, 8, 9, 255, 256, 257, 511, 2047, 2049, 5000] + [int(v) for v in rng.integers(1, 5000, 4)]:
                n = 4096 + 2048 * (nseg - 1) + int(rng.integers(0, 2048))
                tuned.set_tuning(build or None, sched=int(rng.integers(-1, 3)), chunk=int(rng.integers(0, 6)))
                res = []
                for plan, o in ((tuned, outs[0]), (gen, outs[1])):
                    assert plan.csd_exec_dev(dx, dy, n, o, o + 4 * 4096, o + 8 * 4096, o + 16 * 4096) == nseg
                    res.append(ctx.d2h(o, (5, 4096), np.float32).astype(np.float64))
                a, b = res
                lvl = np.sqrt(np.maximum(b[0], 0.1 * np.median(b[0])) * np.maximum(b[1], 0.1 * np.median(b[1])))
                assert np.max(np.abs(a[0] - b[0]) / np.maximum(b[0], 0.1 * np.median(b[0]))) < 5e-5, (det, nseg)
                assert np.max(np.abs(a[1] - b[1]) / np.maximum(b[1], 0.1 * np.median(b[1]))) < 5e-5, (det, nseg)
                pa, pb = (r[2:4].reshape(-1, 2) for r in (a, b))           # Pxy interleaved re, im over rows 2-3
                assert np.max(np.abs(pa - pb) / lvl[:, None]) < 5e-5, (det, nseg)
                if nseg >= 8:
                    assert np.max(np.abs(a[4] - b[4])) < 1e-4, (det, nseg)        # coherence
    finally:
        for ptr in (dx, dy) + tuple(outs):
            ctx.free(ptr)


# ---- round 5: host-output ring (oth_welch_exec polls a completion word), oth_welch_exec_async / _poll / _wait, and the
# pilot formed inside the launch (WelchArgs.pilot_inline) -------------------------------------------------------------

def test_welch_exec_async_poll_wait_equal_exec(ctx, hip):
    """The work()-hosted Welch scan (python/spectrum_sensor.py:71-75,105-117 -> ofdm_cr_tools.py:471-537 needs a call that
    does not block its scheduler thread): exec_async returns while the GPU is still busy, poll() says so, wait()
    delivers exactly what the blocking exec delivers (static schedule: bit for bit), the host buffer may be overwritten
    as soon as exec_async returns, and the ring keeps the last four launches."""
    import time
    from ofdm_tools import windows
    n = 1 << 22
    x = R.synth_iq(n, 1002)
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0)
    plan.set_schedule(hip.SCHED_CONTIGUOUS)
    want = plan.exec(x)
    _, ref = R.welch_np(x, fs=1.0, nperseg=4096, nfft=4096)
    assert relerr(want, ref) < RTOL and plan.last_nseg == 2047
    # 1. a long device-resident queue in front: the async call returns long before the GPU gets to it
    big = 1 << 27
    d_big, d_out = ctx.alloc(big * 8), ctx.alloc(4096 * 4)
    try:
        ctx.synth_iq(d_big, big, 7, ((0.5, 0.1234),), 0.1 + 0.05j)
        ctx.sync()
        small = x[:1 << 16].copy()               # a work()-sized buffer (512 KiB): through the pinned ring
        want_small = plan.exec(small)
        assert np.array_equal(plan.wait(plan.exec_async(small)), want_small)      # (allocates the pinned slot)
        for _ in range(12):                      # ~12 x 0.3 ms of queued work
            plan.exec_dev(d_big, big, d_out)
        t0 = time.perf_counter()
        ticket = plan.exec_async(small)
        dt = time.perf_counter() - t0
        small[:] = 0                             # the caller's buffer dies when work() returns
        first_look = plan.poll(ticket)
        got = plan.wait(ticket)
        assert first_look is None, 'the GPU had ~3.6 ms of work queued in front of this ticket'
        assert dt < 2e-3, dt                     # enqueue only
        assert np.array_equal(got, want_small) and plan.last_nseg == 31
        assert np.array_equal(plan.poll(ticket), want_small)          # a finished ticket can be read again
        # a large pageable buffer: the runtime stages it (the call may wait for the stream); same result
        scratch = x.copy()
        ticket = plan.exec_async(scratch)
        scratch[:] = 0
        assert np.array_equal(plan.wait(ticket), want) and plan.last_nseg == 2047
        # 2. device source, several tickets in flight, collected out of order
        d_x = ctx.alloc(n * 8)
        ctx.h2d(d_x, x)
        ts = [plan.exec_async(d_x, n) for _ in range(4)]
        for t in reversed(ts):
            assert np.array_equal(plan.wait(t), want)
        # 3. the ring keeps four launches: the first of six is gone, the last four are there
        ts = [plan.exec_async(d_x, n) for _ in range(6)]
        with pytest.raises(hip.HipError) as e:
            plan.wait(ts[0])
        assert e.value.code == -5
        for t in ts[2:]:
            assert np.array_equal(plan.wait(t), want)
        with pytest.raises(hip.HipError):
            plan.poll(0)
        ctx.free(d_x)
    finally:
        ctx.free(d_big)
        ctx.free(d_out)
    plan.close()


@pytest.mark.parametrize('dc', [0.1 + 0.05j, 70 + 35j, 3000 - 4000j])
def test_pilot_formed_inside_the_launch_equals_the_pilot_launch(ctx, hip, dc):
    """WelchArgs.pilot_inline (round 5): the role-split 4096-point kernels - one- and two-channel - form the pilot of
    the constant detrend from eight 2 KiB probes in their own prologue instead of reading pilot_mean_kernel's result.
    Any constant near the mean comes off exactly, so both forms must agree with the float64 oracle at the default
    mode's flat 1e-4 at every offset, and with each other far below it; three streams with different offsets in ONE
    launch check that every stream takes its own."""
    from ofdm_tools import windows
    hann = windows.get_window('hann', 4096)
    n = 1 << 18
    x = (R.synth_iq(n, 41) + np.complex64(dc)).astype(np.complex64)
    y = (R.synth_iq(n, 42) - np.complex64(dc) * np.complex64(0.5)).astype(np.complex64)
    _, ref = R.welch_np(x, fs=1.0, nperseg=4096, nfft=4096)
    inl = ctx.welch_plan(4096, window=hann, fs=1.0)
    lau = ctx.welch_plan(4096, window=hann, fs=1.0)
    lau.set_tuning('plaunch')
    a, b = inl.exec(x), lau.exec(x)
    assert relerr(a, ref) < RTOL and relerr(b, ref) < RTOL
    assert relerr(a, b) < 5e-6
    # one and two segments: every probe falls into the same segment
    for m in (4096, 6144):
        _, r1 = R.welch_np(x[:m], fs=1.0, nperseg=4096, nfft=4096)
        assert relerr(inl.exec(x[:m]), r1) < RTOL
    # streams with their own offsets in one launch
    xs = np.stack([x, (x - np.complex64(dc)).astype(np.complex64), y])
    d_in, d_out = ctx.alloc(xs.nbytes), ctx.alloc(3 * 4096 * 4)
    try:
        ctx.h2d(d_in, xs.reshape(-1))
        inl.exec_dev(d_in, n, d_out, nstreams=3, stream_stride=n)
        rows = ctx.d2h(d_out, (3, 4096), np.float32)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    for row, s in zip(rows, xs):
        _, r = R.welch_np(s, fs=1.0, nperseg=4096, nfft=4096)
        assert relerr(row, r) < RTOL
    # the two-channel kernel
    _, rc, rxx, ryy, rxy = R.coherence_np(x, y, nperseg=4096, nfft=4096)
    for plan in (inl, lau):
        gxx, gyy, gxy, gc = plan.csd(x, y)
        assert relerr(gxx, rxx) < RTOL and relerr(gyy, ryy) < RTOL
        assert np.max(np.abs(gxy - rxy) / np.sqrt(rxx * ryy)) < RTOL and np.max(np.abs(gc - rc)) < RTOL
    inl.close()
    lau.close()


def test_recipe_occupancy_table_matches_the_runtime(ctx, hip):
    """tests/test_abi_cpu.py::test_launch_recipes_table enumerates the launch recipes without a GPU, with the resident
    workgroups per CU taken from a built-in table.  Here, on the device: that table equals what the occupancy calculator
    says for every tuned build, and the recipe a real launch recorded equals the table's recipe for the same shape."""
    import recipes
    lib = hip.load()
    from ofdm_tools import windows
    shapes = [dict(nfft=4096, nperseg=4096, noverlap=2048), dict(nfft=4096, nperseg=4096, noverlap=2048, window=2),
              dict(nfft=4096, nperseg=4096, noverlap=1024), dict(nfft=4096, nperseg=1024, noverlap=512),
              dict(nfft=4096, nperseg=4096, noverlap=2048, two_channel=1),
              dict(nfft=4096, nperseg=4096, noverlap=2048, two_channel=1, variant='csd1'),
              dict(nfft=16384, nperseg=16384, noverlap=0, window=0, detrend=0), dict(nfft=16384, nperseg=16384, noverlap=8192),
              dict(nfft=8192, nperseg=8192, noverlap=4096), dict(nfft=8192, nperseg=8192, noverlap=0, window=0, detrend=0)]
    for nfft in (256, 512, 1024, 2048):
        shapes += [dict(nfft=nfft, nperseg=nfft, noverlap=nfft // 2), dict(nfft=nfft, nperseg=nfft, noverlap=0),
                   dict(nfft=nfft, nperseg=nfft, noverlap=nfft // 2, variant='seg3'),
                   dict(nfft=nfft, nperseg=nfft, noverlap=nfft // 2, variant='seg4')]
    for nfft in (1024, 2048):
        shapes += [dict(nfft=nfft, nperseg=nfft // f, noverlap=nov) for f in (4, 2) for nov in (nfft // f // 2, 0)]
    wrong = []
    for sh in shapes:
        a = recipes.recipe(lib, nseg=20000, runtime=0, **sh)
        b = recipes.recipe(lib, nseg=20000, runtime=1, **sh)
        if a != b:
            wrong.append((sh, a.split('nbig')[1], b.split('nbig')[1]))
    assert not wrong, wrong
    # a real launch records the recipe it ran
    n = 4096 + 2048 * 599
    d, o = ctx.alloc(n * 8), ctx.alloc(4096 * 4)
    try:
        ctx.synth_iq(d, n, 5, (), 0.1 + 0.05j)
        plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096))
        assert plan.last_recipe() == ''
        assert plan.exec_dev(d, n, o) == 600
        assert plan.last_recipe() == recipes.recipe(lib, nfft=4096, nperseg=4096, noverlap=2048, nseg=600)
        plan.set_tuning('plaunch')
        plan.exec_dev(d, n, o)
        assert 'pilot=launch' in plan.last_recipe() and 'kernel=welch4096:ws' in plan.last_recipe()
        plan.close()
    finally:
        ctx.free(d)
        ctx.free(o)


def test_welch_wait_from_another_thread_and_the_synchronising_fallback(ctx, hip, tmp_path):
    """oth_welch_wait holds no context lock while it polls: a watcher thread waits for tickets that the stream-side thread
    keeps issuing on the same context (the work() / watcher split of python/spectrum_sensor_v2.py:138-155 applied to the
    Welch scan).  And OTH_HOSTWAIT=sync - hipStreamSynchronize instead of the polled completion word, the path the
    200 ms fallback also takes - gives the same PSD (own process: the switch is read once)."""
    import queue
    import subprocess
    import sys
    import threading
    from ofdm_tools import windows
    x = R.synth_iq(1 << 16, 77)
    plan = ctx.welch_plan(4096, window=windows.get_window('hann', 4096), fs=1.0)
    plan.set_schedule(hip.SCHED_CONTIGUOUS)
    want = plan.exec(x)
    tickets, got, errors = queue.Queue(), [], []

    def watcher():
        try:
            while True:
                t = tickets.get()
                if t is None:
                    return
                got.append(plan.wait(t))
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    th = threading.Thread(target=watcher)
    th.start()
    for i in range(24):
        tickets.put(plan.exec_async(x))
        while tickets.qsize() > 2:      # the ring keeps four launches: stay inside it
            pass
    tickets.put(None)
    th.join(30)
    assert not th.is_alive() and not errors, errors
    assert len(got) == 24 and all(np.array_equal(g, want) for g in got)
    plan.close()
    np.save(str(tmp_path / 'x.npy'), x)
    np.save(str(tmp_path / 'want.npy'), want)
    code = ('import sys, numpy as np\n'
            'sys.path.insert(0, %r)\n'
            'from ofdm_tools import _hip, windows\n'
            'ctx = _hip.Context(0)\n'
            'plan = ctx.welch_plan(4096, window=windows.get_window("hann", 4096), fs=1.0)\n'
            'plan.set_schedule(_hip.SCHED_CONTIGUOUS)\n'
            'x, want = np.load(%r), np.load(%r)\n'
            'assert np.array_equal(plan.exec(x), want)\n'
            'assert np.array_equal(plan.wait(plan.exec_async(x)), want)\n'
            'print("sync ok")\n') % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gr-ofdm_tools_amd'),
                                     str(tmp_path / 'x.npy'), str(tmp_path / 'want.npy'))
    p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, OTH_HOSTWAIT='sync'), capture_output=True, timeout=300)
    assert p.returncode == 0 and b'sync ok' in p.stdout, p.stderr.decode()[-1500:]
