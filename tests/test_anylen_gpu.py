"""GPU parity at the transform lengths the reference accepts and the power-of-two kernels do not take (round 6,
csrc/fft_any.hip): lengths that are not a power of two (mixed radix 2-3-5-7, Bluestein for the rest) and powers of two
above 16384 (the four-step route) - Welch averages, two-channel sums, the GNU Radio chains, xcorr / fac.

Reference call sites with no length limit: fft.fft_vcc(self.fft_len, ...) (psd_logger.py:48, spectrum_sensor_v2.py:90,
local_worker.py:62-63), sg.welch(nperseg=nFFT, nfft=nFFT) (ofdm_cr_tools.py:214,322,342), np.fft.fft(., nFFT)
(ofdm_cr_tools.py:177,157-160), fast_spectrum_scan's own nFFT = 2^ceil(log2(npts)) (ofdm_cr_tools.py:474-475).

Tolerances as everywhere: 1e-4 relative on averaged linear power over all bins (Cxy 1e-4 absolute), the 4-ulp-of-the-peak
criterion on single periodogram rows (tests/test_hip_parity.py::check_single_rows).
"""
import numpy as np
import pytest

from oracle import ref_cpu as R
from test_hip_parity import RTOL, check_single_rows, relerr

pytestmark = pytest.mark.gpu

# direct (smooth) | Bluestein in one launch | Bluestein through the four-step route | four-step powers of two | tiny powers of two
SMOOTH = [96, 1000, 1536, 3000, 6000, 12000, 15000]
BLUESTEIN = [97, 1021, 4099, 8191]
BLUESTEIN2 = [10007, 20000]
BIG = [32768, 65536]
TINY = [8, 32]


@pytest.fixture(scope='module')
def hip():
    from ofdm_tools import _hip
    return _hip


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def _win(name, n):
    from ofdm_tools import windows
    return windows.get_window(name, n)


def _route(plan):
    return plan.last_recipe().split()[0]


@pytest.mark.parametrize('n', SMOOTH + BLUESTEIN + BLUESTEIN2 + BIG + TINY)
def test_welch_any_length_against_the_oracle(ctx, hip, n):
    nseg = 9 if n >= 8192 else 41
    x = R.synth_iq(n // 2 * (nseg + 1) + 3, 600 + n % 97)
    _, ref = R.welch_np(x, nperseg=n, nfft=n)
    plan = ctx.welch_plan(n, window=_win('hann', n))
    psd = plan.exec(x)
    assert plan.last_nseg == (len(x) - n // 2) // (n - n // 2)
    assert 'anyfft' in _route(plan), plan.last_recipe()
    err = relerr(psd, ref)
    print('welch n=%d %s nseg=%d: %.2e' % (n, _route(plan), plan.last_nseg, err))
    assert err < RTOL
    plan.close()


@pytest.mark.parametrize('n', BIG)
@pytest.mark.parametrize('variant', [None, 'r16', 'anycov'])
def test_two_level_routes_across_workspace_chunks(ctx, hip, n, variant):
    """32768 / 65536 points: the default route (full segments: welch32k.hip's one-workgroup kernel; otherwise fft_tl.hip's
    register radix-16 four-step kernels, which 'r16' forces for full segments too) and fft_any.hip's coverage kernels
    ('anycov') on a launch longer than one workspace chunk (64 MiB: 256 / 128 segments), ragged segment count, zero
    padding inside the last row block, with and without the detrend."""
    nseg = (64 << 20) // (8 * n) + 37
    x = R.synth_iq(n // 2 * (nseg + 1) + 11, 71, dc=3 + 2j)
    for detrend, nper in ((hip.DETREND_CONSTANT, n), (hip.DETREND_NONE, n - 3000)):
        kw = dict(detrend=False) if detrend == hip.DETREND_NONE else {}
        xs = x if nper == n else x[:nper * 9]
        _, ref = R.welch_np(xs, nperseg=nper, nfft=n, **kw)
        plan = ctx.welch_plan(n, nperseg=nper, window=_win('hann', nper), detrend=detrend)
        if variant:
            plan.set_tuning(variant)
        psd = plan.exec(xs)
        assert plan.last_nseg == (len(xs) - nper // 2) // (nper - nper // 2)
        assert relerr(psd, ref) < RTOL, (detrend, nper)
        want = 'twolevel' if variant == 'anycov' else ('onewg' if nper == n and not variant else 'twolevel:r16')
        assert _route(plan) == 'kernel=anyfft:' + want
        plan.close()


@pytest.mark.parametrize('n', BIG)
def test_welch_32768_65536_inside_one_workgroup(ctx, hip, n):
    """welch32k.hip: the whole 32768-point segment in one workgroup's registers (radix 2 + two 16384-point transforms), exact
    mean in double, partial rows in layout 7; 65536 points as a PAIR of workgroups behind one more radix-2 step (even / odd
    bins; the mean taken off z0 = x w +- x' w' afterwards; layout 8).  Against scipy's form in f64: one segment (the
    single-row criterion) to more segments than workgroups, 50 % / odd / no overlap, a workgroup count that is not a multiple
    of 8 / 16 (no XCD dealing), with and without the detrend under a DC line of 36 sigma, a first sample that is not 16-byte
    aligned, fftshift + trim + dB, several streams per launch, the streaming form (accumulate / finalize) - and the same
    plans on the four-step route."""
    w = _win('hann', n)
    for detrend in (True, False):
        for nseg, ov in ((1, n // 2), (5, n // 2), (9, 0), (37, 1001), (300, n // 2), (700, n - 4096)):
            step = n - ov
            x = R.synth_iq(n + step * (nseg - 1) + 17, 100 + nseg, dc=(30 + 20j) if detrend else R.DC)
            plan = ctx.welch_plan(n, noverlap=ov, window=w, detrend=detrend)
            got = plan.exec(x)
            assert _route(plan) == 'kernel=anyfft:onewg' and plan.last_nseg == nseg
            _, ref = R.welch_np(x, nperseg=n, nfft=n, noverlap=ov, detrend='constant' if detrend else False)
            if nseg == 1:
                check_single_rows(got[None, :], ref[None, :], ulps=4)
            else:
                assert relerr(got, ref) < RTOL, (detrend, nseg, ov)
            four = ctx.welch_plan(n, noverlap=ov, window=w, detrend=detrend)
            four.set_tuning('r16')
            other = four.exec(x)
            assert _route(four) == 'kernel=anyfft:twolevel:r16'
            if nseg > 1:
                assert relerr(got, other.astype(np.float64)) < RTOL
            four.close()
            plan.close()
    # device input at an odd sample offset (8-byte aligned only), two streams per launch, shifted + trimmed + dB
    nseg, trim = 21, 1000
    m = n // 2 * (nseg + 1)
    xs = [R.synth_iq(m, 7), R.synth_iq(m, 8, tones=((1.0, 0.2),))]
    d = ctx.alloc((2 * m + 1) * 8)
    try:
        ctx.h2d(d + 8, np.concatenate(xs))
        plan = ctx.welch_plan(n, window=w, fs=2.5e6, fftshift=True, trim_bins=trim, db=True)
        out = ctx.alloc(2 * (n - 2 * trim) * 4)
        assert plan.exec_dev(d + 8, m, out, nstreams=2, stream_stride=m) == nseg
        got = ctx.d2h(out, (2, n - 2 * trim), np.float32)
        ctx.free(out)
        for i in range(2):
            _, ref = R.welch_np(xs[i], nperseg=n, nfft=n, fs=2.5e6)
            assert relerr(10.0 ** (got[i].astype(np.float64) / 10.0), np.fft.fftshift(ref)[trim:n - trim]) < RTOL
        assert _route(plan) == 'kernel=anyfft:onewg'
        plan.close()
    finally:
        ctx.free(d)
    # streaming: ragged pushes carry the overlap across calls
    x = R.synth_iq(n // 2 * 41 + 123, 9)
    plan = ctx.welch_plan(n, window=w)
    for a, b in ((0, 50000), (50000, 50001), (50001, 400000 * (n // 32768)), (400000 * (n // 32768), len(x))):
        plan.accumulate(x[a:b])
    got = plan.finalize()
    _, ref = R.welch_np(x, nperseg=n, nfft=n)
    assert plan.last_nseg == 40 and relerr(got, ref) < RTOL
    plan.close()


def test_routes_are_the_documented_ones(ctx):
    want = {96: 'direct', 15000: 'direct', 97: 'bluestein', 8191: 'bluestein', 10007: 'bluestein2', 20000: 'bluestein2',
            32768: 'onewg', 65536: 'onewg', 131072: 'twolevel', 32: 'direct'}
    for n, kind in want.items():
        plan = ctx.welch_plan(n)
        plan.exec(R.synth_iq(2 * n, 1))
        assert _route(plan) == 'kernel=anyfft:' + kind, (n, plan.last_recipe())
        plan.close()


@pytest.mark.parametrize('n', [1000, 4099, 20000, 65536])
def test_welch_flattop_zero_padded_shifted_trimmed_db(ctx, hip, n):
    """The sweeper's call shape (spectrum_sweeper.py:263-276) at lengths outside the power-of-two kernels: flat-top of
    nfft / 4 points zero-padded to nfft, fftshift (odd lengths included), excess bins dropped, dB."""
    nper, trim = n // 4, n // 16
    x = R.synth_iq(nper * 12 + 5, 77)
    _, ref = R.welch_np(x, window='flattop', nperseg=nper, nfft=n, fs=2.5e6)
    ref = np.fft.fftshift(ref)[trim:n - trim]
    plan = ctx.welch_plan(n, nperseg=nper, window=_win('flattop', nper), fs=2.5e6, fftshift=True, trim_bins=trim, db=True)
    db = plan.exec(x)
    assert db.shape == (n - 2 * trim,)
    assert relerr(10.0 ** (db.astype(np.float64) / 10.0), ref) < RTOL
    plan.close()


@pytest.mark.parametrize('n', [1536, 1021, 20000, 32768])
def test_welch_detrend_under_a_dc_line_and_without(ctx, hip, n):
    x = R.synth_iq(n * 6, 5, dc=30 + 20j)            # DC 36 sigma
    plan = ctx.welch_plan(n, window=_win('hann', n))
    _, ref = R.welch_np(x, nperseg=n, nfft=n)
    assert relerr(plan.exec(x), ref) < RTOL
    plan.close()
    # without the detrend the line stays in the data: the SURVEY 8d offset (0.1 + 0.05j), as every other no-detrend case
    # (a float32 transform leaves ~1e-7 of a line's amplitude in every bin: 36 sigma would be 1e-4 of the noise bins)
    x = R.synth_iq(n * 6, 5)
    plan = ctx.welch_plan(n, window=_win('hann', n), detrend=hip.DETREND_NONE, scaling=hip.SCALE_SPECTRUM)
    _, ref = R.welch_np(x, nperseg=n, nfft=n, detrend=False, scaling='spectrum')
    assert relerr(plan.exec(x), ref) < RTOL
    plan.close()


@pytest.mark.parametrize('n', [96, 3000, 4099, 10007, 32768, 65536])
def test_csd_and_coherence_any_length(ctx, hip, n):
    nseg = 7 if n >= 8192 else 33
    x = R.synth_iq(n // 2 * (nseg + 1), 11)
    noise = R.synth_iq(len(x), 12, tones=(), dc=0)
    y = 0.7 * np.roll(x, 5) + 0.5 * noise
    plan = ctx.welch_plan(n, window=_win('hann', n))
    pxx, pyy, pxy, cxy = plan.csd(x, y)
    if n in (32768, 65536):                      # both channels on the register radix-16 kernels, four sums per bin
        assert _route(plan) == 'kernel=anyfft:twolevel:r16' and 'nch=4' in plan.last_recipe()
    _, c_ref, pxx_ref, pyy_ref, pxy_ref = R.coherence_np(x, y, nperseg=n, nfft=n)
    assert relerr(pxx, pxx_ref) < RTOL and relerr(pyy, pyy_ref) < RTOL
    assert np.max(np.abs(pxy - pxy_ref)) / np.max(np.abs(pxy_ref)) < RTOL
    assert np.max(np.abs(pxy - pxy_ref) / np.sqrt(pxx_ref * pyy_ref)) < RTOL
    assert np.max(np.abs(cxy - c_ref)) < RTOL
    plan.close()


@pytest.mark.parametrize('n,detrend', [(32768, True), (65536, True), (32768, False)])
def test_two_channel_fast_route_in_workspace_chunks_matches_the_coverage_kernels(ctx, hip, n, detrend, monkeypatch):
    """The 32768 / 65536 two-channel route with a workspace of 5 segments per channel (so 4 chunks of partial sums, the last
    short), segment length a multiple of 4096 (sub-block means) and not (per-segment means): against scipy's form in f64
    and against the same plan on the coverage kernels (tuning 'anycov')."""
    monkeypatch.setenv('OTH_ANY_WS_MB', str(5 * 2 * n * 8 >> 20))
    for nper in (n, n - 1000):
        step = nper - nper // 2
        x = R.synth_iq(nper // 2 + step * 18, 3, dc=2 - 1j)
        y = 0.6 * np.roll(x, 11) + 0.8 * R.synth_iq(len(x), 4, tones=(), dc=0.5)
        kw = dict(nperseg=nper, window=_win('hann', nper), detrend=detrend)
        plan = ctx.welch_plan(n, **kw)
        got = plan.csd(x, y)
        assert _route(plan) == 'kernel=anyfft:twolevel:r16'
        _, c_ref, pxx_ref, pyy_ref, pxy_ref = R.coherence_np(x, y, nperseg=nper, nfft=n, detrend='constant' if detrend else False)
        cov = ctx.welch_plan(n, **kw)
        cov.set_tuning('anycov')
        ref = cov.csd(x, y)
        assert _route(cov) == 'kernel=anyfft:twolevel'
        for g, r64, r32 in zip(got, (pxx_ref, pyy_ref, pxy_ref, c_ref), ref):
            scale = np.max(np.abs(r64))
            assert np.max(np.abs(g - r64)) / scale < RTOL, (nper, detrend)
            assert np.max(np.abs(g - r32)) / scale < RTOL
        plan.close()
        cov.close()


@pytest.mark.parametrize('n', [1000, 1536, 12000, 1021, 20000, 32768])
def test_chain_rows_any_length(ctx, hip, n):
    """a1: rectangular, shifted, |X|^2 / N^2, every 2nd vector kept; single rows by the 4-ulp criterion, the mean of all
    rows by the plain 1e-4."""
    nvec = 12
    x = R.synth_iq(n * nvec + n // 3, 31)
    ref = R.chain_sensor_v2(x, n, decim=2)
    ch = ctx.chain(n, window=None, fftshift=True, epilogue=hip.EPI_MAG2_OVER_N2, keep_one_in_n=2)
    rows, got = ch.push(x)
    assert got == nvec // 2 and rows.shape == ref.shape
    # Bluestein rows carry the rounding of two transforms of 2-4 x the length: 8 ulp of the peak
    check_single_rows(rows, ref, ulps=4 if R_is_smooth_or_pow2(n) else 8)
    assert relerr(rows.mean(axis=0), ref.mean(axis=0)) < RTOL
    ch.close()


def R_is_smooth_or_pow2(n):
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


@pytest.mark.parametrize('n', [1000, 4099])
def test_chain_psd_logger_and_local_worker_forms_any_length(ctx, hip, n):
    x = R.synth_iq(n * 9, 32)
    bh = R.gr_blackmanharris(n)
    mag, peak = R.chain_psd_logger(x, n)
    ch = ctx.chain(n, window=bh, fftshift=False, epilogue=hip.EPI_MAG)
    ch.set_peak_hold(True)
    rows, got = ch.push(x)
    assert got == 9
    check_single_rows(rows, mag, power=False, ulps=4 if R_is_smooth_or_pow2(n) else 8)
    assert relerr(ch.peak(), peak[-1]) < RTOL
    ch.close()
    lin, db = R.chain_local_worker(x, n, 1e6, 0.3)
    k = -10 * np.log10(n) - 10 * np.log10(1e6)
    ch = ctx.chain(n, window=bh, fftshift=True, epilogue=hip.EPI_MAG2)
    ch.set_iir_log(0.3, k)
    rows, got = ch.push(x)
    assert relerr(10 ** ((rows[-1].astype(np.float64) - k) / 10), lin[-1]) < RTOL
    assert relerr(ch.iir(), lin[-1]) < RTOL
    ch.close()


def test_chain_ragged_pushes_keep_their_state_any_length(ctx, hip):
    n = 1536
    x = R.synth_iq(n * 20, 33)
    ref = R.chain_sensor_v2(x, n, decim=3)
    ch = ctx.chain(n, window=None, fftshift=True, epilogue=hip.EPI_MAG2_OVER_N2, keep_one_in_n=3)
    rng = np.random.default_rng(4)
    got, pos = [], 0
    while pos < len(x):
        m = int(rng.integers(1, 3 * n))
        rows, k = ch.push(x[pos:pos + m])
        got.extend(rows[:k])
        pos += m
    got = np.array(got)
    assert got.shape == ref.shape
    check_single_rows(got, ref)
    ch.close()


@pytest.mark.parametrize('L', [100, 1000, 1021, 4099, 20000, 32768, 65536])
def test_xcorr_fac_any_length(ctx, L):
    rng = np.random.default_rng(L)
    na, nb = L - L // 5, L // 2 + 1
    a = (rng.normal(size=na) + 1j * rng.normal(size=na)).astype(np.complex64)
    b = np.roll(a, 3)[:nb].copy()
    ref = R.xcorr(a, b, L)
    out = ctx.xcorr(a, b, L)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) / np.max(ref) < RTOL
    ref = R.fac(a, L)
    out = ctx.fac(a, L)
    assert np.max(np.abs(out - ref)) / np.max(ref) < RTOL


def test_xcorr_truncates_longer_inputs_like_numpy(ctx):
    rng = np.random.default_rng(9)
    a = (rng.normal(size=300) + 1j * rng.normal(size=300)).astype(np.complex64)
    for L in (256, 200):      # the power-of-two kernel and the any-length route
        ref = R.xcorr(a, a[::-1].copy(), L)
        out = ctx.xcorr(a, a[::-1].copy(), L)
        assert np.max(np.abs(out - ref)) / np.max(ref) < RTOL


def test_streaming_accumulate_any_length(ctx, hip):
    n = 3000
    x = R.synth_iq(n * 15 + 77, 40)
    _, ref = R.welch_np(x, nperseg=n, nfft=n)
    plan = ctx.welch_plan(n, window=_win('hann', n))
    rng = np.random.default_rng(1)
    pos = 0
    while pos < len(x):
        m = int(rng.integers(100, 2 * n))
        plan.accumulate(x[pos:pos + m])
        pos += m
    assert relerr(plan.finalize(), ref) < RTOL
    assert plan.last_nseg == (len(x) - n // 2) // (n - n // 2)
    plan.close()


def test_many_streams_and_partials_any_length(ctx, hip):
    n, ns = 1000, 5
    per = n * 20
    x = np.concatenate([R.synth_iq(per, 50 + i) for i in range(ns)])
    plan = ctx.welch_plan(n, window=_win('hann', n))
    d = ctx.alloc(x.nbytes)
    o = ctx.alloc(4 * n * ns)
    ctx.h2d(d, x.astype(np.complex64))
    plan.exec_dev(d, per, o, nstreams=ns, stream_stride=per)
    rows = ctx.d2h(o, (ns, n), np.float32)
    for i in range(ns):
        _, ref = R.welch_np(x[i * per:(i + 1) * per], nperseg=n, nfft=n)
        assert relerr(rows[i], ref) < RTOL
    # raw partial sums of two halves add up to the whole (time-sharded form)
    s1, s2 = ctx.alloc(4 * n), ctx.alloc(4 * n)
    half = (per // 2 // (n // 2)) * (n // 2)
    k1 = plan.partial_dev(d, half + n // 2, s1)
    k2 = plan.partial_dev(d + 8 * half, per - half, s2)
    a, b = ctx.d2h(s1, (n,), np.float32), ctx.d2h(s2, (n,), np.float32)
    _, ref = R.welch_np(x[:per], nperseg=n, nfft=n, scaling='raw')
    assert k1 + k2 == (per - n // 2) // (n // 2)
    assert relerr((a.astype(np.float64) + b) / (k1 + k2), ref) < RTOL
    for p in (d, o, s1, s2):
        ctx.free(p)
    plan.close()


def test_tuned_request_and_oversize_are_refused_with_a_reason(ctx, hip):
    with pytest.raises(hip.HipError) as e:
        ctx.welch_plan(1000, kernel=hip.KERNEL_TUNED).exec(R.synth_iq(4000, 1))
    assert e.value.code == -3
    with pytest.raises(hip.HipError) as e:
        ctx.welch_plan((1 << 20) + 2)
    assert e.value.code == -3 and '1048576' in str(e.value)
    with pytest.raises(hip.HipError) as e:
        ctx.chain(600000)          # Bluestein M = 2^21
    assert e.value.code == -3


def test_ref_any_length_scans_plots_and_xcorr_on_the_gpu(ctx, golden):
    """ref_anylen.npz - outputs of the reference's OWN fast_spectrum_scan / src_power_welch / src_power_fft /
    welch_plot_dB / welch_power_estimate / xcorr / fac bodies (tests/golden/make_golden.py --reference) at lengths the
    library refused before round 6 - against the drop-in helpers of ofdm_tools.ofdm_cr_tools on the GPU:
    fast_spectrum_scan(n_fft=0) on 20 000 / 100 000 samples (32768 / 131072 points), n_fft 1000 / 3000, both methods."""
    from ofdm_tools import ofdm_cr_tools as T
    from test_oracle_golden import anylen_cases
    g = golden('ref_anylen.npz')
    Sf, cs, sbw, fc = int(g['Sf']), float(g['channel_rate']), float(g['srch_bw']), float(g['fc'])
    for i, x, n_fft, method, nfft, stride in anylen_cases(g):
        Fr = float(Sf) / nfft
        bb = T.frange(-Sf / 2, Sf / 2, cs)
        fn = T.src_power_welch if method == 'welch' else T.src_power_fft
        psd, _, plc = fn(x, len(x), nfft, Fr, Sf, bb, sbw / Fr, ctx=ctx)
        assert len(psd) == nfft
        assert relerr(psd[::stride], g['psd_%d' % i]) < RTOL, (i, method, nfft)
        assert relerr(plc, g['plc_%d' % i]) < RTOL
        thr, plc_s, ne, occ = T.fast_spectrum_scan(x, fc, cs, sbw, n_fft, Sf, method, int(g['thr_leveler']),
                                                   float(g['noise0']), float(g['alpha']), ctx=ctx)
        assert relerr(plc_s, g['plc_%d' % i]) < RTOL
        assert abs(thr - float(g['thr_%d' % i])) < RTOL * float(g['thr_%d' % i])
        assert abs(ne - float(g['noise_%d' % i])) < RTOL * float(g['noise_%d' % i])
        assert [1.0 if a in occ else 0.0 for a in g['ax_ch']] == list(g['occupied_%d' % i])
    x = R.synth_iq(int(g['plot_n']), int(g['plot_seed']))
    _, db = T.welch_plot_dB(x, Sf, fc, 1000, ctx=ctx)
    assert relerr(10 ** (np.array(db) / 10), 10 ** (g['plot_db_1000'] / 10)) < RTOL
    for key, nfft in (('power_6000', 6000), ('power_1021', 1021), ('power_short_40000', 40000)):
        assert abs(T.welch_power_estimate(x, nfft, Sf, ctx=ctx) - float(g[key])) < RTOL * float(g[key]), key
    a = R.synth_iq(int(g['xcorr_lens'][0]), int(g['xcorr_seeds'][0]), tones=(), dc=0)
    b = R.synth_iq(int(g['xcorr_lens'][1]), int(g['xcorr_seeds'][1]), tones=(), dc=0)
    for L in (1000, 1001, 20000):
        ref = g['xcorr_%d' % L]
        assert np.max(np.abs(T.xcorr(a, b, L, ctx=ctx) - ref)) / np.max(ref) < RTOL
        ref = g['fac_%d' % L]
        assert np.max(np.abs(T.fac(a, L, ctx=ctx) - ref)) / np.max(ref) < RTOL


# ---- the blocks at fft_len values that are not powers of two (the reference passes fft_len straight to fft.fft_vcc) --------

def test_blocks_accept_any_fft_len(ctx, hip, tmp_path):
    import ofdm_tools
    from ofdm_tools import packets
    # spectrum_sensor_v2 (spectrum_sensor_v2.py:85-93): 1000 points, keep_one_in_n(5), channel powers against the oracle
    fft_len, Sf = 1000, 1000 * 100
    blk = ofdm_tools.spectrum_sensor_v2(fft_len, 20, Sf, channel_space=5000, search_bw=2500, trunc_band=Sf - 10000, stats=True,
                                        ctx=ctx, threaded=False)
    assert blk.decimation == 5
    x = R.synth_iq(fft_len * 40 + 13, 17)
    assert blk.feed(x, max_items=3777) == len(x)
    st = R.ScannerState(fft_len, Sf, 5000, 2500, trunc_band=Sf - 10000)
    rows = R.chain_sensor_v2(x, fft_len, decim=5)
    for r in rows:
        st.scan(r.astype(np.float32))
    assert blk._scanner.n_measurements == len(rows) == 8
    assert np.allclose(blk.power_level_ch, st.plc, rtol=1e-4)
    # multichannel_scanner (multichannel_scanner.py:78-86) at 12000 points
    fft_len, Sf = 12000, 1200000
    st = R.ScannerState(fft_len, Sf, 25e3, 12.5e3, tune_freq=0, trunc_band=Sf)
    subj = [st.ax_ch[i] for i in (3, 9, 20, 31, 40)]
    blk = ofdm_tools.multichannel_scanner(fft_len, 1000, Sf, channel_space=25e3, search_bw=12.5e3, tune_freq=0, trunc_band=Sf,
                                          subject_channels=subj, ctx=ctx, threaded=False)
    x = R.synth_iq(fft_len * 3, 3000)
    blk.feed(x, max_items=fft_len)
    for r in R.chain_sensor_v2(x, fft_len, decim=blk.decimation):
        st.scan(r.astype(np.float32))
    assert np.allclose(blk.power_level_ch, st.plc, rtol=1e-4)
    assert blk.top4 == R.publish_top4(st.plc, st.ax_ch, subj)[1]
    # psd_logger (psd_logger.py:43-56,85) at a prime length: Blackman-Harris, |X|, running peak, np.save per vector
    n = 1021
    path = str(tmp_path / 'psd_log.npy')
    blk = ofdm_tools.psd_logger(n, 1000, n * 1000, ctx=ctx, threaded=False, mat_file=path)
    x = R.synth_iq(n * 7, 9)
    blk.feed(x, max_items=n)
    _, peak = R.chain_psd_logger(x, n)
    assert relerr(blk.peak_vals, peak[-1]) < RTOL and relerr(np.load(path), peak[-1]) < RTOL
    # local_worker (local_worker.py:58-71,147-172) at 3000 points: IIR + log row, fragments of the float32 payload
    N, Sf, alpha = 3000, 3000000, 0.4
    blk = ofdm_tools.local_worker(N, Sf, alpha, Sf / N, 1472, True, ctx=ctx, threaded=False)
    frames = []
    blk.msg_connect('pdus', lambda m: frames.append(m[1]))
    x = R.synth_iq(N * 6, 10)
    blk.feed(x, max_items=N)
    lin, _ = R.chain_local_worker(x, N, Sf, alpha)
    k = -10 * np.log10(N) - 10 * np.log10(Sf)
    assert relerr(10 ** ((blk.last_db.astype(np.float64) - k) / 10), lin[-1]) < RTOL
    assert frames[-9:] == R.worker_fragments(blk.last_db, 1470, N, True)      # ceil(12000 / 1470) = 9
    assert np.array_equal(np.frombuffer(packets.reassemble(frames[-9:]), '<f4'), blk.last_db)
    # ascii_plot's chain (ascii_plot.py:57-70) at 1536 points
    N, Sf = 1536, 1536 * 30
    blk = ofdm_tools.ascii_plot(N, Sf, 433.0e6, 0.3, 10, 64, 20, ctx=ctx, threaded=False)
    x = R.synth_iq(N * 31 + 5, 77)
    blk.feed(x, max_items=4000)
    lin, db = R.chain_ascii_plot(x, N, Sf, 0.3, decim=3)
    assert relerr(blk._chain.iir(), lin[-1]) < RTOL and blk.last_plot


class _Rx(object):
    def __init__(self):
        self.tuned = []

    def set_center_freq(self, f, chan):
        self.tuned.append(f)


def test_sweeper_estimator_and_legacy_sensor_accept_any_fft_len(ctx, hip, tmp_path):
    import ofdm_tools
    # spectrum_sweeper (spectrum_sweeper.py:62-70,260-276): fft_len 3000 -> flat-top of 750 points zero-padded to 3000
    rx = _Rx()
    fft_len, Sf, tSf = 3000, 2000000, 1750000
    blk = ofdm_tools.spectrum_sweeper(rx, 'rtl', fft_len, Sf, tSf, 100e6, 107e6, 15, 0.0, 8, 0, 1472, ctx=ctx, threaded=False)
    pts, tune, excess = R.sweeper_geometry(fft_len, Sf, tSf, 100e6, 107e6, 8)
    assert (blk.vector_probe_pts, blk.tune_frequencies, blk.excess_bins) == (pts, tune, excess)
    vectors = [R.synth_iq(pts, 2000 + i) for i in range(len(tune))]
    it = iter(vectors)
    blk.get_samples = lambda: next(it)
    psd = blk.sweep_once(sleep=lambda s: None)
    ref = R.sweeper_stitch(vectors, fft_len, Sf, excess, 0.0)
    assert psd.shape == ref.shape and relerr(10 ** (psd / 10), 10 ** (ref / 10)) < RTOL
    # coherence_estimator -> coherence_detector (coherence_detector.py:184-202,254-274) at N = 1000
    N, Sf, tune_f = 1000, 2000000, 433000000
    x = R.synth_iq(N * 40, 11)
    y = (0.7 * np.roll(x, 5) + 0.5 * R.synth_iq(len(x), 12, tones=(), dc=0)).astype(np.complex64)
    est = ofdm_tools.coherence_estimator(N, Sf, block_len=len(x), ctx=ctx)
    est.work([x, y], [])
    _, cref, _, _, _ = R.coherence_np(x, y, fs=Sf, nperseg=N, nfft=N)
    assert np.max(np.abs(est.cxy - np.fft.fftshift(cref))) < RTOL
    det = ofdm_tools.coherence_detector(N, Sf, threshold=1.2, threshold_mtm=0.2, tune_freq=tune_f,
                                        subject_channels=[tune_f + 0.1234 * Sf, tune_f - 0.31 * Sf])
    quiet = np.zeros(N, np.float32)
    det.work([est.cxy.reshape(1, N), quiet.reshape(1, N), quiet.reshape(1, N)], [])
    coh, outcome, _ = R.coherence_scanner(np.fft.fftshift(cref), quiet, quiet, det.idx_subject_channels, 1.2, 0.2)
    assert det.get_subject_channels_outcome() == outcome
    assert np.allclose(det.subject_channels_coherence, coh, atol=2e-4)


def test_eight_threads_block_in_exec_on_one_plan(ctx, hip):
    """Advisor, round 5: oth_welch_exec takes a ticket and collects it from a four-slot output ring outside the context
    lock; with five or more threads blocked in it on ONE plan the fifth launch rewrote the first caller's row.  Blocking
    calls of one plan now run one after the other (a plan mutex across enqueue + collect): eight threads, one cached plan,
    each its own input - every caller gets ITS spectrum, none an OTH_ERR_STATE."""
    import threading
    n = 4096
    plan = ctx.cached_plan(('t8', n), lambda: ctx.welch_plan(n, window=_win('hann', n)))
    assert ctx.cached_plan(('t8', n), lambda: None) is plan
    xs = [R.synth_iq(n * 40, 800 + i, tones=((1.0 + i, 0.05 * (i + 1)),)) for i in range(8)]
    refs = [R.welch_np(x, nperseg=n, nfft=n)[1] for x in xs]
    errs, bad = [None] * 8, []

    def run(i):
        try:
            worst = 0.0
            for _ in range(25):
                worst = max(worst, relerr(plan.exec(xs[i]), refs[i]))
            errs[i] = worst
        except Exception as e:      # noqa: BLE001
            bad.append((i, repr(e)))
    th = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not bad, bad
    assert all(e is not None and e < RTOL for e in errs), errs
    # the low-CPU wait mode gives the same answer
    plan.set_hostwait(True)
    assert relerr(plan.exec(xs[0]), refs[0]) < RTOL
    plan.set_hostwait(False)


def test_plan_cache_is_lru_and_keeps_plans_that_owe_a_ticket(ctx, hip):
    made = []

    def mk(n):
        def f():
            made.append(n)
            return ctx.welch_plan(n)
        return f
    x = R.synth_iq(4096, 3)
    p64 = ctx.cached_plan(('lru', 64), mk(64), limit=3)
    t = p64.exec_async(x)                       # owes a result from here on
    for n in (128, 256, 512, 1024):
        ctx.cached_plan(('lru', n), mk(n), limit=3)
    assert p64.h                                # survived four insertions past the limit
    assert p64.wait(t).shape == (64,) and p64.outstanding == 0
    ctx.cached_plan(('lru', 2048), mk(2048), limit=3)
    assert not p64.h                            # collected: now it is the oldest and goes
    assert made == [64, 128, 256, 512, 1024, 2048]


def test_fuzz_one_workgroup_route_against_the_oracle(ctx, hip):
    """40 random full-segment plans at 32768 / 65536 points (welch32k.hip): any overlap, 1 ... 600 segments (fewer and more than
    workgroups, counts that are not multiples of 8 / 16), detrend on / off under offsets of 0 ... 300 sigma, four window kinds
    and a random window, device input at odd sample offsets - each against the float64 oracle, criteria as in the fuzz below."""
    rng = np.random.default_rng(32768)
    worst = 0.0
    for i in range(40):
        n = (32768, 65536)[i & 1]
        nov = (n // 2, 0, int(rng.integers(0, n)), n - int(rng.integers(1, 4096)))[int(rng.integers(0, 4))]
        step = n - nov
        nseg = int(rng.integers(1, 40)) if step > 4096 else int(rng.integers(1, 600))
        detrend = bool(rng.random() < 0.6)
        kind = int(rng.integers(0, 5))
        w = _win(('hann', 'flattop', 'boxcar', 'blackmanharris')[kind], n) if kind < 4 else rng.random(n).astype(np.float32) + 0.1
        dc = complex(rng.normal(), rng.normal()) * (10.0 ** rng.uniform(-1, 2.5)) if detrend else R.DC
        lead = int(rng.integers(0, 3))
        x = R.synth_iq(lead + n + step * (nseg - 1) + int(rng.integers(0, step)), 7000 + i, dc=dc)
        _, ref = R.welch_np(x[lead:], window=w, nperseg=n, noverlap=nov, nfft=n, detrend='constant' if detrend else False)
        plan = ctx.welch_plan(n, noverlap=nov, window=w, detrend=hip.DETREND_CONSTANT if detrend else hip.DETREND_NONE)
        d = ctx.alloc(len(x) * 8)
        try:
            ctx.h2d(d, x)
            got = plan.exec_device_src(d + 8 * lead, len(x) - lead)
        finally:
            ctx.free(d)
        assert _route(plan) == 'kernel=anyfft:onewg' and plan.last_nseg == nseg, (n, nov, nseg, plan.last_recipe())
        rel = np.abs(got - ref) / np.maximum(ref, 1e-9 * ref.max())
        amp = np.abs(np.sqrt(np.maximum(got, 0)) - np.sqrt(ref)) / np.sqrt(ref.max())
        weak = rel >= RTOL
        worst = max(worst, float(rel[~weak].max()) if (~weak).any() else 0.0)
        assert np.all(amp[weak] <= 4 * 2.0 ** -23), (n, nov, nseg, detrend, kind, abs(dc), float(rel.max()), float(amp[weak].max() * 2.0 ** 23))
        assert not np.any(weak & (ref >= np.median(ref))), (n, nov, nseg, detrend, kind, abs(dc), float(rel.max()))
        assert nseg < 8 or not weak.any(), (n, nov, nseg, detrend, kind, abs(dc), float(rel.max()))
        plan.close()
    print('one-workgroup fuzz: worst relative error outside the few-segment regime %.1e' % worst)


def test_fuzz_any_length_shapes_against_the_oracle(ctx, hip):
    """120 random plans outside the power-of-two kernels - lengths 1 ... 40000 (small primes, prime powers, 2-3-5-7-smooth
    numbers, primes next to the route boundaries 16384 / 8192), nperseg <= nfft, any overlap, detrend on / off, window
    kinds, shift + trim - each against the float64 oracle on its own seeded input (the coverage routes have to be RIGHT)."""
    rng = np.random.default_rng(20260)
    special = [1, 2, 3, 5, 6, 7, 9, 10, 11, 15, 25, 27, 49, 63, 65, 100, 121, 127, 129, 243, 255, 257, 343, 625, 1023, 1025, 2187,
               4095, 4097, 8191, 8193, 16383, 16385, 16807, 15625, 14406, 13122, 16380, 32767, 32769, 40000]
    sizes = special + [int(v) for v in rng.integers(12, 20000, 79)]
    worst = {}
    for i, n in enumerate(sizes):
        if (n & (n - 1)) == 0 and 64 <= n <= 16384:
            n += 1                                     # (the power-of-two kernels have their own fuzz: tools/fuzz_shapes.py)
        nper = n if rng.random() < 0.5 else int(rng.integers(max(1, n // 4), n + 1))
        nov = int(rng.integers(0, nper)) if rng.random() < 0.7 else nper // 2
        detrend = bool(rng.random() < 0.6) and nper >= 8
        wname = ('hann', 'flattop', 'boxcar')[int(rng.integers(0, 3))]
        nseg = int(rng.integers(1, 12)) if n > 4096 else int(rng.integers(1, 60))
        step = nper - nov
        x = R.synth_iq(nper + step * (nseg - 1) + int(rng.integers(0, step)), 5000 + i)
        shift = bool(rng.random() < 0.5)
        trim = int(rng.integers(0, max(1, n // 3))) if rng.random() < 0.3 else 0
        w = _win(wname, nper)
        _, ref = R.welch_np(x, window=w, nperseg=nper, noverlap=nov, nfft=n, detrend='constant' if detrend else False)
        if shift:
            ref = np.fft.fftshift(ref)
        ref = ref[trim:n - trim]
        plan = ctx.welch_plan(n, nperseg=nper, noverlap=nov, window=w, detrend=hip.DETREND_CONSTANT if detrend else hip.DETREND_NONE,
                              fftshift=shift, trim_bins=trim)
        got = plan.exec(x)
        route = _route(plan).split(':')[1]
        assert plan.last_nseg == nseg, (n, nper, nov, plan.last_nseg, nseg)
        # Every bin: the plain 1e-4, or - the regime of few segments under the 2.0-amplitude tone, where any float32
        # transform leaves about an ulp of the LARGEST amplitude in every bin (check_single_rows) - an amplitude error of at
        # most 4 ulp of the spectrum's peak (Bluestein: 8, two transforms of 2-4 x the length); bins at or above the median
        # always the plain 1e-4.  (Bins the detrend / window empties to rounding level are measured against the peak.)
        rel = np.abs(got - ref) / np.maximum(ref, 1e-9 * ref.max())
        amp = np.abs(np.sqrt(np.maximum(got, 0)) - np.sqrt(ref)) / np.sqrt(ref.max())
        ulps = 8 if route.startswith('bluestein') else 4
        weak = rel >= RTOL
        worst[route] = max(worst.get(route, 0.0), float(rel[~weak].max()) if (~weak).any() else 0.0)
        assert np.all(amp[weak] <= ulps * 2.0 ** -23), (n, nper, nov, detrend, wname, shift, trim, route, float(rel.max()),
                                                       float(amp[weak].max() * 2.0 ** 23))
        assert not np.any(weak & (ref >= np.median(ref))), (n, nper, nov, detrend, wname, route, float(rel.max()))
        assert nseg < 8 or not weak.any(), (n, nper, nov, nseg, route, float(rel.max()))      # averages of 8+ segments: plain 1e-4 everywhere
        plan.close()
    print('fuzz worst by route: ' + ', '.join('%s %.1e' % kv for kv in sorted(worst.items())))
    assert set(worst) >= {'direct', 'bluestein', 'bluestein2'}


def test_any_length_results_do_not_depend_on_timing_under_a_bandwidth_hog(ctx, hip):
    """The dynamic side of the hazard question for the round-6 kernels (csrc/fft_any.hip, fft_tl.hip: workgroup barriers
    between Stockham passes, wave-level exchanges in tl_k2, the workspace hand-over between launches, read-modify-write of
    the partial rows across chunks): every route - all of them walk fixed runs of segments, so their sums are bit-reproducible
    - is run alone and then again and again while a second context streams 8 GiB reads over the same HBM; the results must be
    IDENTICAL bit for bit (tests/test_hip_parity.py does the same for the tuned builds)."""
    import threading
    n = 1 << 23
    d = ctx.alloc((n + 5) * 8)
    hog = hip.Context(0)
    hog_bytes = 8 << 30
    hog_buf = hog.alloc(hog_bytes)
    try:
        ctx.synth_iq(d, n + 5, 4343, R.TONES, R.DC)
        plans = []
        for nfft, variant in ((1000, None), (12000, None), (4099, None), (20000, None), (32768, None), (32768, 'r16'), (65536, None), (65536, 'r16'), (32768, 'anycov'),
                              (131072, None)):
            for det in (hip.DETREND_CONSTANT, hip.DETREND_NONE):
                plan = ctx.welch_plan(nfft, window=_win('hann', nfft), detrend=det)
                if variant:
                    plan.set_tuning(variant)
                m = n if nfft <= 65536 and variant != 'anycov' else n // 8      # (the coverage routes above 16384 are slow: fewer segments)
                plans.append(('welch %d %s det %d' % (nfft, variant or '', det), plan, lambda p=plan, m=m: p.exec_device_src(d, m)))
        csd = ctx.welch_plan(3000, window=_win('hann', 3000))
        plans.append(('csd 3000', csd, lambda: np.concatenate([np.asarray(v).view(np.float32).ravel()
                                                             for v in csd.csd_device_src(d + 40, d, n // 4)])))
        ch = ctx.chain(1536, None, True, hip.EPI_MAG2, 16)
        plans.append(('chain 1536', ch, lambda: ch.push(ctx.d2h(d, (1536 * 16 * 20,), np.complex64), 4)[0].ravel()))      # whole groups of 16 vectors: every push alike
        quiet = [run().copy() for _, _, run in plans]
        stop = threading.Event()

        def stream_reads():
            while not stop.is_set():
                hog.stream_read_probe(hog_buf, hog_bytes, 4)
        th = threading.Thread(target=stream_reads, daemon=True)
        th.start()
        try:
            for rep in range(6):
                for (name, _, run), want in zip(plans, quiet):
                    got = run()
                    assert got.tobytes() == want.tobytes(), (name, rep, float(np.max(np.abs(got - want) / np.abs(want))))
        finally:
            stop.set()
            th.join(60)
        for _, plan, _ in plans:
            plan.close()
    finally:
        ctx.free(d)
        hog.free(hog_buf)
        hog.close()
