"""Pins the CPU oracle (oracle/ref_cpu.py) to the committed golden vectors,
i.e. to SciPy / NumPy called with the reference's own argument patterns
(tests/golden/make_golden.py).  CPU only."""
import os
import struct

import numpy as np
import pytest

from oracle import ref_cpu as R

RTOL = 1e-9  # float64 restatement vs float64 SciPy


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / np.abs(b))


def test_welch_hann_default_call_pattern(golden):
    g = golden('welch_hann_4096_50.npz')
    f, p = R.welch_np(g['x'], fs=float(g['fs']), nperseg=4096, nfft=4096)
    assert relerr(p, g['expected_psd']) < RTOL
    assert np.allclose(f, g['expected_freqs'])
    # SciPy on complex64 input (what the reference really runs) stays within 1e-4
    assert relerr(g['scipy_c64_psd'], g['expected_psd']) < 1e-4


def test_welch_ragged_length_and_fs(golden):
    g = golden('welch_hann_1024_ragged.npz')
    _, p = R.welch_np(g['x'], fs=float(g['fs']), nperseg=1024, nfft=1024)
    assert relerr(p, g['expected_psd']) < RTOL


def test_src_power_welch_flattop(golden):
    g = golden('welch_flattop_2048.npz')
    N, Sf = int(g['nfft']), float(g['fs'])
    psd, axis, plc = R.src_power_welch(g['x'], len(g['x']), N, Sf / N, Sf,
                                       R.frange(-Sf / 2, Sf / 2, 50e3), 25e3 / (Sf / N))
    assert relerr(psd, np.fft.fftshift(g['expected_psd'])) < RTOL
    assert len(plc) == 20 and all(v > 0 for v in plc)


def test_sweeper_segment(golden):
    g = golden('welch_flattop_nperseg_quarter.npz')
    db = R.sweeper_src_power(g['x'], int(g['nfft']), float(g['fs']), int(g['excess_bins']))
    assert db.shape == g['expected_psd_db'].shape
    assert relerr(10 ** (db / 10), g['expected_psd_lin']) < 1e-9
    assert np.max(np.abs(db - g['expected_psd_db'])) < 1e-9


def test_csd_coherence(golden):
    g = golden('coherence_csd_4096.npz')
    _, cxy, pxx, pyy, pxy = R.coherence_np(g['x'], g['y'], fs=1.0, nperseg=4096, nfft=4096)
    assert relerr(pxx, g['expected_pxx']) < RTOL
    assert relerr(pyy, g['expected_pyy']) < RTOL
    assert np.max(np.abs(pxy - g['expected_pxy']) / np.abs(g['expected_pxy'])) < 1e-8
    assert np.max(np.abs(cxy - g['expected_cxy'])) < 1e-9


def test_gr_chain_rect(golden):
    g = golden('gr_chain_rect_1024.npz')
    rows = R.chain_sensor_v2(g['x'], 1024)
    assert relerr(rows, g['expected_rows']) < RTOL
    assert relerr(rows.reshape(-1, 8, 1024).mean(axis=1), g['expected_mean8']) < RTOL


def test_gr_chain_psd_logger(golden):
    g = golden('gr_chain_bh_mag_peak_4096.npz')
    assert np.allclose(R.gr_blackmanharris(4096), g['window'], rtol=0, atol=1e-15)
    mag, peak = R.chain_psd_logger(g['x'], 4096)
    assert relerr(mag, g['expected_mag']) < RTOL
    assert relerr(peak, g['expected_peak']) < RTOL


def test_gr_chain_local_worker(golden):
    g = golden('gr_chain_bh_iir_log_2048.npz')
    lin, db = R.chain_local_worker(g['x'], 2048, int(g['sample_rate']), float(g['average']))
    assert relerr(lin, g['expected_lin']) < RTOL
    assert np.max(np.abs(db - g['expected_db'])) < 1e-9


def test_keep_one_in_n_takes_last_of_group():
    x = np.arange(10 * 4).astype(np.complex64)
    v = R.gr_kept_vectors(x, 4, 3)
    assert v.shape == (3, 4)
    assert v[0, 0] == 8 and v[1, 0] == 20 and v[2, 0] == 32
    assert R.gr_decimation(1000000, 1024, 10) == 97  # int(1000000/1024/10) with py2 int division


def test_src_power_cases(golden):
    g = golden('src_power_cases.npz')
    for i in range(int(g['n'])):
        Sf, N = int(g['Sf_%d' % i]), int(g['N_%d' % i])
        cs, sbw = float(g['cs_%d' % i]), float(g['sbw_%d' % i])
        Fr = float(Sf) / N
        psd = g['psd_%d' % i]
        assert np.allclose(R.movingaverage(psd, sbw / Fr), g['ma_%d' % i], rtol=1e-12)
        plc = R.src_power(psd, N, Fr, Sf, R.frange(-Sf // 2, Sf // 2, cs), sbw / Fr)
        assert np.allclose(plc, g['plc_%d' % i], rtol=1e-12)


def test_frange_variants():
    assert R.frange(0, 1, 0.25) == [0, 0.25, 0.5, 0.75]
    assert R.frange_le(0, 1, 0.25) == [0, 0.25, 0.5, 0.75, 1.0]
    # float accumulation: 0.1 summed 10 times is < 1.0, so one extra element appears
    assert len(R.frange(0, 1, 0.1)) == 11


def test_scanner_state_sequence(golden):
    g = golden('scanner_state_seq.npz')
    st = R.ScannerState(1024, 1000000, 25e3, 12.5e3, tune_freq=100000000, trunc_band=800000,
                        thr_leveler=4, alpha_avg=0.5)
    assert np.allclose(st.ax_ch, g['ax_ch'])
    assert st.trunc_ch == 4
    for i, r in enumerate(g['rows']):
        _, occ = st.scan(r.astype(np.float32))
        assert np.allclose(st.plc, g['plc_seq'][i], rtol=1e-12)
        assert np.isclose(st.noise_estimate, g['noise_seq'][i], rtol=1e-12)
        assert [1.0 if a in occ else 0.0 for a in st.ax_ch] == list(g['occupied_seq'][i])
    pwr, top4 = R.publish_top4(st.plc, st.ax_ch, list(g['subject_channels']))
    assert np.allclose(pwr, g['subject_pwr']) and top4 == list(g['top4'])


def test_coherence_scanner(golden):
    g = golden('coherence_scanner.npz')
    ax = R.coherence_axis(int(g['N']), int(g['sample_rate']), int(g['tune_freq']))
    idx = [R.find_nearest_index(ax, c) for c in g['subject_channels']]
    assert idx == list(g['idx'])
    coh, outcome, valve = R.coherence_scanner(g['d0'], g['d1'], g['d2'], idx, 10, 0.2)
    assert np.allclose(coh, g['coherence']) and outcome == list(g['outcome']) and valve == list(g['valve'])
    assert 1 in outcome and 0.1 in outcome  # both branches exercised


def test_xcorr_fac(golden):
    g = golden('xcorr_fac.npz')
    L = int(g['L'])
    xc = R.xcorr(g['a'], g['b'], L)
    assert relerr(xc, g['expected_xcorr']) < RTOL
    assert int(np.argmax(xc)) == 37  # b is a delayed by 37 samples
    assert relerr(R.fac(g['a'], L), g['expected_fac']) < RTOL


def test_fragment_consumer_restatement_against_the_byte_fixture():
    """oracle.consumer_handler (remote_client_qt.py:100-164, sdr_webserver_ws.py:235-287) vs fragments_consumer.bin."""
    from test_host_logic_cpu import read_consumer_fixture
    for header, precision, frames, vectors in read_consumer_fixture():
        got, mx = R.consumer_handler(frames, '<f4' if precision else np.int8, header)
        assert len(got) == len(vectors) and all(np.array_equal(a, b) for a, b in zip(got, vectors))
        assert np.array_equal(mx, np.maximum.reduce(vectors))
    assert R.zmq_pdu_header(1472) == bytes([7, 6, 10, 0, 0, 0, 5, 0xc0, 1, 0]) and len(R.zmq_pdu_header(5)) == 10


def test_ref_psd_logger_watcher_peak_rule(golden):
    """a2, host side: oracle.chain_psd_logger's running peak against what the reference's own ``_queue_watcher.run``
    saved message by message (ref_psd_logger.npz; psd_logger.py:70-88) - and how that thread ends at its first
    two-vector message (``s`` used before assignment, :79-81)."""
    ref = golden('ref_psd_logger.npz')
    g = golden(str(ref['input_from']))
    _, peak = R.chain_psd_logger(g['x'], int(g['nfft']))
    assert peak.shape == ref['saved_peaks'].shape and np.allclose(peak, ref['saved_peaks'], rtol=1e-6, atol=0)
    assert str(ref['ended']) == 'UnboundLocalError' and int(ref['died_at_message']) == 16


def test_ref_fragment_consumer_restatement_against_the_reference_consumers(golden):
    """oracle.consumer_handler against what the reference's own consumers did with the same frames (ref_consumers.npz):
    vectors in completion order and the final peak, with the 10-byte header / cleared pending payload of the web server
    and the bare frames / kept pending payload of the Qt client - lost fragments and the error branch included."""
    g = golden('ref_consumers.npz')

    def frames_of(tag):
        raw, out, pos = bytes(g['frames_' + tag]), [], 0
        for n in g['frames_%s_len' % tag]:
            out.append(raw[pos:pos + int(n)])
            pos += int(n)
        return out
    for tag in ('f32', 'sweeper', 'lossy'):
        frames = [R.zmq_pdu_header(len(fr)) + fr for fr in frames_of(tag)]
        got, _ = R.consumer_handler(frames, '<f4', 10, clear_on_error=True)
        assert [v.tobytes() for v in got] == [bytes(g['web_%s_out_%d' % (tag, j)]) for j in range(len(g['web_%s_at' % tag]))]
    for tag, dt in (('f32', '<f4'), ('i8', np.int8), ('lossy', '<f4')):
        got, peak = R.consumer_handler(frames_of(tag), dt, 0, clear_on_error=False)
        n = len(g['qt_%s_at' % tag])
        assert len(got) == n
        assert np.array_equal(peak, g['qt_%s_peak_%d' % (tag, n - 1)], equal_nan=True)


def test_fragment_wire_format(golden):
    path = os.path.join(os.path.dirname(__file__), 'golden', 'fragments.bin')
    raw = open(path, 'rb').read()
    pos, groups = 0, []
    for _ in range(3):
        n = struct.unpack_from('<I', raw, pos)[0]
        pos += 4
        frames = []
        for _ in range(n):
            ln = struct.unpack_from('<I', raw, pos)[0]
            pos += 4
            frames.append(raw[pos:pos + ln])
            pos += ln
        groups.append(frames)
    db = (np.arange(4096, dtype=np.float32) * 0.01 - 90).astype('<f4')
    assert R.worker_fragments(db, 1470, 4096, True) == groups[0]
    assert R.worker_fragments(db, 1470, 4096, False) == groups[1]
    assert R.sweeper_fragments(db.tobytes(), 1470) == groups[2]
    # worker: ceil(16384/1470) = 12 frames; sweeper quirk: floor(16384/1470)+1 = 12 too
    assert len(groups[0]) == 12 and groups[0][0][0] == 12 and groups[0][5][1] == 5
    assert b''.join(f[2:] for f in groups[0]) == db.tobytes()
    assert len(groups[1]) == 3 and len(groups[2]) == 12


def test_known_answers():
    # pure tone at bin k0, rectangular a1 chain -> P[k0] = A^2 (after fftshift)
    N, k0, A = 1024, 100, 0.75
    x = (A * np.exp(2j * np.pi * k0 * np.arange(N) / N)).astype(np.complex64)
    row = R.chain_sensor_v2(x, N)[0]
    assert np.isclose(row[N // 2 + k0], A * A, rtol=1e-6) and np.sum(row) < A * A * (1 + 1e-6)
    # Parseval for Welch density on white noise
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(1 << 16) + 1j * rng.standard_normal(1 << 16)) / np.sqrt(2)
    _, p = R.welch_np(x, fs=1.0, nperseg=1024, nfft=1024)
    assert abs(p.sum() * 1.0 / 1024 - 1.0) < 0.02
    # x == y -> coherence 1
    _, c, *_ = R.coherence_np(x[:16384], x[:16384], nperseg=1024, nfft=1024)
    assert np.allclose(c, 1.0)


# ---------------------------------------------------------------------------------------------
# Fixtures tagged source='reference' (tests/golden/ref_*.npz): outputs of the reference's OWN function
# bodies, cut out of /root/reference/python/*.py and exec'd by tests/golden/ref_extract.py in the build
# container.  These pin the restatement to the reference itself, not to a re-reading of it.
# ---------------------------------------------------------------------------------------------

def test_reference_fixtures_are_tagged(golden):
    import glob
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'ref_*.npz')))
    assert names == ['ref_anylen.npz', 'ref_ascii_plot.npz', 'ref_coherence_scanner.npz', 'ref_consumers.npz', 'ref_fft_plot.npz', 'ref_flank.npz',
                     'ref_legacy_sensor.npz', 'ref_psd_logger.npz',
                     'ref_scanner_seq.npz', 'ref_sensing_log.npz',
                     'ref_src_power_cases.npz',
                     'ref_src_power_fft.npz', 'ref_src_power_welch_2048.npz', 'ref_sweeper_src_power.npz',
                     'ref_threads.npz', 'ref_welch_hann_4096.npz', 'ref_xcorr_fac.npz']
    for n in names:
        g = golden(n)
        assert str(g['source']) == 'reference'
        if n not in ('ref_anylen.npz', 'ref_ascii_plot.npz', 'ref_consumers.npz', 'ref_flank.npz', 'ref_sensing_log.npz'):      # (carry their own inputs / seeds)
            assert os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', str(g['input_from'])))


def test_ref_welch_plot_db_and_power_estimate(golden):
    """a6: ofdm_cr_tools.py:321-326 (welch_plot_dB) and :341-345 (welch_power_estimate)."""
    g = golden('ref_welch_hann_4096.npz')
    x = golden(str(g['input_from']))['x']
    Sf, fc, nfft = int(g['Sf']), float(g['fc']), int(g['nfft'])
    axis, db = R.welch_plot_dB(x, Sf, fc, nfft)
    assert np.allclose(axis, g['expected_axis'], rtol=0, atol=1e-6)
    assert relerr(10 ** (np.array(db) / 10), 10 ** (g['expected_db'] / 10)) < RTOL
    assert abs(R.welch_power_estimate(x, nfft, Sf) - float(g['expected_power'])) < RTOL * float(g['expected_power'])
    assert abs(R.welch_power_estimate(x, nfft, 1.0) - float(g['expected_power_fs1'])) < RTOL * float(g['expected_power_fs1'])
    assert abs(R.clc_power_freq(x[:4096].astype(np.complex128), 4096, Sf) - float(g['expected_clc_power_freq'])) < \
        RTOL * float(g['expected_clc_power_freq'])
    # the SciPy-tagged fixture of the same call agrees with what the reference's own lines returned
    lin = np.fft.fftshift(golden('welch_hann_4096_50.npz')['expected_psd']) / Sf       # fs = 1 there
    assert relerr(lin, 10 ** (g['expected_db'] / 10) - 1e-20) < 1e-9


def test_ref_src_power_welch_and_fast_spectrum_scan(golden):
    """a6/a14: ofdm_cr_tools.py:213-230 (src_power_welch), :471-537 (fast_spectrum_scan, method 'welch')."""
    g = golden('ref_src_power_welch_2048.npz')
    x = golden(str(g['input_from']))['x']
    Sf, N, cs, sbw = int(g['Sf']), int(g['nfft']), float(g['channel_rate']), float(g['srch_bw'])
    Fr = float(Sf) / N
    bb = R.frange(-Sf // 2, Sf // 2, cs)
    assert np.array_equal(bb, g['bb_freqs'])
    psd, ax, plc = R.src_power_welch(x, len(x), N, Fr, Sf, bb, sbw / Fr)
    assert relerr(psd, g['expected_psd']) < RTOL and np.allclose(ax, g['expected_axis'])
    assert relerr(plc, g['expected_plc']) < RTOL
    ne = float(g['scan_noise0'])
    for i, (lo, hi) in enumerate(g['scan_ranges']):
        thr, plc, ne, occ = R.fast_spectrum_scan(x[lo:hi], float(g['scan_fc']), cs, sbw, N, Sf, 'welch',
                                                 int(g['scan_thr_leveler']), ne, float(g['scan_alpha']))
        assert abs(thr - g['scan_thr'][i]) < RTOL * g['scan_thr'][i]
        assert abs(ne - g['scan_noise'][i]) < RTOL * g['scan_noise'][i]
        assert relerr(plc, g['scan_plc'][i]) < RTOL
        assert [1.0 if a in occ else 0.0 for a in g['ax_ch']] == list(g['scan_occupied'][i])
    assert 0 < g['scan_occupied'].sum() < g['scan_occupied'].size      # both outcomes present


def test_ref_sweeper_src_power_and_frange(golden):
    """a4: spectrum_sweeper.py:260-276 (_src_power), :37-42 (inclusive frange)."""
    g = golden('ref_sweeper_src_power.npz')
    x = golden(str(g['input_from']))['x']
    db = R.sweeper_src_power(x, int(g['nfft']), float(g['fs']), int(g['excess_bins']))
    assert db.shape == g['expected_db'].shape and np.max(np.abs(db - g['expected_db'])) < 1e-9
    db2 = R.sweeper_src_power(x[:9000], 1024, 250000.0, 0)
    assert np.max(np.abs(db2 - g['expected_db_notrim'])) < 1e-9
    assert np.array_equal(R.frange_le(88.0e6 + 1.0e6, 108.0e6, 2.0e6), g['frange_le_a'])
    assert np.array_equal(R.frange_le(0, 1, 0.25), g['frange_le_b'])
    assert np.array_equal(R.frange_le(0.0, 1.0, 0.1), g['frange_le_c'])


def test_ref_src_power_movingaverage_frange(golden):
    """a7: ofdm_cr_tools.py:136-141 (frange), :168-170 (movingaverage), :232-249 (src_power)."""
    g = golden('ref_src_power_cases.npz')
    c = golden(str(g['input_from']))
    assert np.array_equal(R.frange(0, 1, 0.25), g['frange_a']) and np.array_equal(R.frange(0, 1, 0.1), g['frange_b'])
    assert np.array_equal(R.frange(-500000.0, 500000.0, 25e3), g['frange_c'])
    for i in range(int(g['n'])):
        Sf, N = int(c['Sf_%d' % i]), int(c['N_%d' % i])
        cs, sbw = float(c['cs_%d' % i]), float(c['sbw_%d' % i])
        Fr = float(Sf) / N
        psd = c['psd_%d' % i]
        bb = R.frange(-Sf // 2, Sf // 2, cs)
        assert np.array_equal(bb, g['bb_%d' % i])
        assert np.allclose(R.movingaverage(psd, sbw / Fr), g['ma_%d' % i], rtol=1e-12, atol=0)
        assert np.allclose(R.src_power(psd, N, Fr, Sf, bb, sbw / Fr), g['plc_%d' % i], rtol=1e-12, atol=0)
        assert np.allclose(R.src_power(psd.astype(np.float32), N, Fr, Sf, bb, sbw / Fr), g['plc_f32_%d' % i],
                           rtol=1e-6, atol=0)      # float32 rows: the order of np.convolve's float32 sums


def test_ref_coherence_scanner(golden):
    """a11: coherence_detector.py:254-274 (watcher.scanner), :276-278 (find_nearest_index)."""
    g = golden('ref_coherence_scanner.npz')
    c = golden(str(g['input_from']))
    ax = R.coherence_axis(int(c['N']), int(c['sample_rate']), int(c['tune_freq']))
    idx = [R.find_nearest_index(ax, f) for f in c['subject_channels']]
    assert idx == list(g['idx'])
    coh, outcome, valve = R.coherence_scanner(c['d0'], c['d1'], c['d2'], idx, int(g['threshold']),
                                              float(g['threshold_mtm']))
    assert np.array_equal(np.array(coh, np.float64), g['coherence'])
    assert outcome == list(g['outcome']) and valve == list(g['valve'])


def test_ref_scanner_ema_and_top4(golden):
    """a8 (edge truncation + 0.6/0.4 EMA) and a9 (top-4): spectrum_sensor_v2.py:533-544, :228-237;
    multichannel_scanner.py:214-239."""
    g = golden('ref_scanner_seq.npz')
    c = golden(str(g['input_from']))
    st = R.ScannerState(1024, 1000000, 25e3, 12.5e3, tune_freq=100000000, trunc_band=800000,
                        thr_leveler=4, alpha_avg=0.5)
    for i, r in enumerate(c['rows']):
        st.scan(r.astype(np.float32))
        assert np.allclose(st.plc, g['plc_seq'][i], rtol=1e-12, atol=0)
    pwr, top4 = R.publish_top4(st.plc, st.ax_ch, list(c['subject_channels']))
    assert np.allclose(pwr, g['subject_pwr'], rtol=1e-12) and top4 == list(g['top4'])


def test_ref_xcorr_fac(golden):
    """a12: ofdm_cr_tools.py:155-166 - the reference's own `xcorr` / `fac` bodies, run in the namespace of their day
    (Python-2 `/` on `len()`, tests/golden/ref_extract.py)."""
    g = golden('ref_xcorr_fac.npz')
    x = golden(str(g['input_from']))
    L = int(g['L'])
    a, b = x['a'].astype(np.complex128), x['b'].astype(np.complex128)      # (NumPy 2 keeps complex64 input in single precision)
    assert relerr(R.xcorr(a, b, L), g['expected_xcorr']) < RTOL
    assert relerr(R.fac(a, L), g['expected_fac']) < RTOL
    assert relerr(R.xcorr(a[:700], b[:900], 1024), g['expected_xcorr_short']) < RTOL
    # the NumPy-made golden of round 1 (computed from the complex64 arrays, i.e. in single precision under NumPy 2)
    # agrees with the reference's float64 output to single precision
    assert relerr(x['expected_xcorr'], g['expected_xcorr']) < 1e-5 and relerr(x['expected_fac'], g['expected_fac']) < 1e-5


def test_ref_src_power_fft_and_fft_scan(golden):
    """a14: `src_power_fft` (ofdm_cr_tools.py:173-192, with `sg.flattop` as SciPy <= 1.12 exported it) and
    `fast_spectrum_scan(method='fft')` (:471-537) - the reference's own bodies."""
    g = golden('ref_src_power_fft.npz')
    x = golden(str(g['input_from']))['x']
    Sf, N = int(g['Sf']), int(g['nfft'])
    cs, sbw = float(g['channel_rate']), float(g['srch_bw'])
    Fr = float(Sf) / N
    bb = R.frange(-Sf / 2, Sf / 2, cs)
    assert np.allclose(bb, g['bb_freqs'])
    psd, ax, plc = R.src_power_fft(x[:N].astype(np.complex128), N, N, Fr, Sf, bb, sbw / Fr)
    assert relerr(psd, g['expected_psd']) < RTOL and np.allclose(ax, g['expected_axis'], rtol=0, atol=1e-6)
    assert np.allclose(plc, g['expected_plc'], rtol=1e-12)
    lo, hi = (int(v) for v in g['short_range'])
    psd, _, plc = R.src_power_fft(x[lo:hi].astype(np.complex128), hi - lo, N, Fr, Sf, bb, sbw / Fr)
    assert relerr(psd, g['expected_psd_short']) < RTOL and np.allclose(plc, g['expected_plc_short'], rtol=1e-12)
    thr, plc, ne, occ = R.fast_spectrum_scan(x[:N].astype(np.complex128), float(g['scan_fc']), cs, sbw, N, Sf, 'fft',
                                             int(g['scan_thr_leveler']), float(g['scan_noise0']), float(g['scan_alpha']))
    assert np.isclose(thr, float(g['scan_thr']), rtol=1e-12) and np.isclose(ne, float(g['scan_noise']), rtol=1e-12)
    assert np.allclose(plc, g['scan_plc'], rtol=1e-12)
    ax_ch = R.frange(float(g['scan_fc']) - Sf / 2, float(g['scan_fc']) + Sf / 2, cs)
    assert [1.0 if a in occ else 0.0 for a in ax_ch] == list(g['scan_occupied'])


def anylen_cases(g):
    """(index, samples, n_fft argument, method, nFFT the reference chose, stride of the stored PSD) of ref_anylen.npz"""
    for i, (npts, n_fft, seed, m, nfft, stride) in enumerate(g['cases']):
        yield i, R.synth_iq(int(npts), int(seed)), int(n_fft), ('welch', 'fft')[int(m)], int(nfft), int(stride)


def test_ref_any_length_scans_plots_and_xcorr(golden):
    """Round 6: the reference's own bodies at the lengths it accepts beyond a power of two in [64, 16384] -
    fast_spectrum_scan(n_fft=0) picking 32768 / 131072 points itself (ofdm_cr_tools.py:474-475; 'welch': SciPy's
    shortened nperseg, one zero-padded segment), free-integer n_fft (1000, 3000), welch_plot_dB / welch_power_estimate
    at 1000 / 6000 / 1021 points and with nFFT above the input length, xcorr / fac at 1000, 1001 and 20000 points."""
    g = golden('ref_anylen.npz')
    Sf, cs, sbw, fc = int(g['Sf']), float(g['channel_rate']), float(g['srch_bw']), float(g['fc'])
    seen = set()
    for i, x, n_fft, method, nfft, stride in anylen_cases(g):
        Fr = float(Sf) / nfft
        bb = R.frange(-Sf / 2, Sf / 2, cs)
        fn = R.src_power_welch if method == 'welch' else R.src_power_fft
        psd, _, plc = fn(x.astype(np.complex128), len(x), nfft, Fr, Sf, bb, sbw / Fr)
        assert len(psd) == nfft and relerr(psd[::stride], g['psd_%d' % i]) < 1e-9
        assert abs(np.sum(psd) - float(g['psd_sum_%d' % i])) < 1e-9 * float(g['psd_sum_%d' % i])
        thr, plc_s, ne, occ = R.fast_spectrum_scan(x.astype(np.complex128), fc, cs, sbw, n_fft, Sf, method,
                                                   int(g['thr_leveler']), float(g['noise0']), float(g['alpha']))
        assert relerr(plc_s, g['plc_%d' % i]) < 1e-9 and relerr(plc, g['plc_%d' % i]) < 1e-9
        assert np.isclose(thr, float(g['thr_%d' % i]), rtol=1e-9) and np.isclose(ne, float(g['noise_%d' % i]), rtol=1e-9)
        assert [1.0 if a in occ else 0.0 for a in g['ax_ch']] == list(g['occupied_%d' % i])
        seen.add(nfft)
    assert seen == {32768, 131072, 1000, 3000}
    x = R.synth_iq(int(g['plot_n']), int(g['plot_seed']))
    _, db = R.welch_plot_dB(x, Sf, fc, 1000)
    assert relerr(10 ** (np.array(db) / 10), 10 ** (g['plot_db_1000'] / 10)) < 1e-9
    for key, nfft in (('power_6000', 6000), ('power_1021', 1021), ('power_short_40000', 40000)):
        assert np.isclose(R.welch_power_estimate(x, nfft, Sf), float(g[key]), rtol=1e-9), key
    a = R.synth_iq(int(g['xcorr_lens'][0]), int(g['xcorr_seeds'][0]), tones=(), dc=0).astype(np.complex128)
    b = R.synth_iq(int(g['xcorr_lens'][1]), int(g['xcorr_seeds'][1]), tones=(), dc=0).astype(np.complex128)
    for L in (1000, 1001, 20000):
        assert g['xcorr_%d' % L].shape == (L - L // 2,)
        # (relative to the peak: beyond the inputs' support the correlation is rounding noise around zero)
        for got, ref in ((R.xcorr(a, b, L), g['xcorr_%d' % L]), (R.fac(a, L), g['fac_%d' % L])):
            assert np.max(np.abs(got - ref)) < 1e-9 * np.max(ref)


def _frames_of(blob):
    """u32 length + bytes records -> list of frames."""
    raw, pos, frames = bytes(blob), 0, []
    while pos < len(raw):
        ln = struct.unpack_from('<I', raw, pos)[0]
        frames.append(raw[pos + 4:pos + 4 + ln])
        pos += 4 + ln
    return frames


def test_ref_stats_scanner_noise_threshold_and_max_hold(golden):
    """a8 in full - EMA, cumulative / periodic max, noise estimate, threshold, occupied channels: the reference's own
    stats_watcher.spectrum_scanner (spectrum_sensor_v2.py:445-479) over the committed rows, against the restatement
    AND against the fixture the restatement wrote (scanner_state_seq.npz, tagged 'restated' - now pinned)."""
    g = golden('ref_threads.npz')
    c = golden(str(g['input_from']))
    st = R.ScannerState(1024, 1000000, 25e3, 12.5e3, tune_freq=100000000, trunc_band=800000,
                        thr_leveler=4, alpha_avg=0.5)
    for i, r in enumerate(c['rows']):
        _, occ = st.scan(r.astype(np.float32))
        assert np.allclose(st.plc, g['stats_plc_seq'][i], rtol=1e-12, atol=0)
        assert np.isclose(st.noise_estimate, g['stats_noise_seq'][i], rtol=1e-12, atol=0)
        assert [1.0 if a in occ else 0.0 for a in st.ax_ch] == list(g['stats_occupied_seq'][i])
    assert np.allclose(st.cumulative_max_power, g['stats_cumulative_max'], rtol=1e-12, atol=0)
    assert np.array_equal(g['stats_cumulative_max'], g['stats_periodic_max'])
    assert 0 < g['stats_occupied_seq'].sum() < g['stats_occupied_seq'].size          # the threshold decides something
    for ref_key, restated_key in (('stats_plc_seq', 'plc_seq'), ('stats_noise_seq', 'noise_seq'),
                                  ('stats_occupied_seq', 'occupied_seq'), ('stats_cumulative_max', 'cumulative_max')):
        assert np.allclose(g[ref_key], c[restated_key], rtol=1e-12, atol=0), ref_key


def test_ref_watchers_last_vector_peak_hold_and_waterfall(golden):
    """a10 / a15: psd_watcher.run, waterfall_watcher.run (spectrum_sensor_v2.py:304-354), main_thread.run
    (local_worker.py:126-139) and data_colector.run (spectrum_sweeper.py:161-172) on messages of 1-4 vectors - only the
    LAST vector of a message is used, the peak is a running bin-wise maximum that starts from the first vector."""
    g = golden('ref_threads.npz')
    c = golden(str(g['input_from']))
    rows = c['rows'].astype(np.float32)
    last = np.cumsum(g['msg_counts']) - 1
    assert np.array_equal(g['waterfall'], rows[last])
    assert np.array_equal(g['worker_vectors'], rows[last])
    assert np.array_equal(g['collector_vectors'], c['x'].astype(np.complex64).reshape(len(rows), -1)[last])
    peak = R.peak_hold(rows[last])[-1]
    assert np.array_equal(g['psd_cumulative'], peak) and np.array_equal(g['psd_periodic_peaks'], peak)
    # the row the GNU Radio chain hands over: keep_one_in_n in front of the sink keeps the LAST of n as well
    assert R.gr_kept_vectors(np.arange(12), 2, 3).tolist() == [[4, 5], [10, 11]]


def test_ref_packers_worker_and_sweeper(golden):
    """f1: the reference's own packet_source.send_packet (local_worker.py:147-172, float32 and int8;
    spectrum_sweeper.py:240-258) - the byte fixture fragments.bin, which the restatement wrote, is what they emit; and
    lengths around the fragment boundary (the worker's ceil, the sweeper's floor + 1 with its empty last frame)."""
    g = golden('ref_threads.npz')
    path = os.path.join(os.path.dirname(__file__), 'golden', 'fragments.bin')
    assert open(path, 'rb').read() == bytes(g['fragments_bin'])
    for n in g['frame_lengths']:
        v = (np.arange(n, dtype=np.float32) * 0.5 - 70).astype('<f4')
        assert R.worker_fragments(v, 1472, int(n), True) == _frames_of(g['worker_frames_%d' % n]), n
        assert R.sweeper_fragments(v.tobytes(), 1472) == _frames_of(g['sweeper_frames_%d' % n]), n
    assert len(_frames_of(g['sweeper_frames_368'])) == 2 and _frames_of(g['sweeper_frames_368'])[1][2:] == b''
    assert len(_frames_of(g['worker_frames_368'])) == 1


def test_ref_stitcher_run(golden):
    """a5: one pass of spectrum_stitcher.run (spectrum_sweeper.py:207-231) - 2 s start delay, retune in list order on
    channel 0 with the tune delay after each, concatenate, blend with the 1e-10 floor it re-creates every sweep, pack
    little-endian float32.  The reference's Welch runs on complex64 here (SciPy keeps single precision), the
    restatement in float64: the packed floats agree to float32 rounding."""
    g = golden('ref_threads.npz')
    assert g['stitch_sleeps'].tolist() == [2.0, 0.125, 0.125, 0.125]
    assert g['stitch_tuned'].tolist() == [[f, 0.0] for f in g['stitch_freqs']]
    nfft, fs, ex = int(g['stitch_nfft']), float(g['stitch_fs']), int(g['stitch_excess'])
    got = np.frombuffer(bytes(g['stitch_packed']), '<f4')
    assert len(got) == 3 * (nfft - 2 * ex)
    want = R.sweeper_stitch(list(g['stitch_captures']), nfft, fs, ex, float(g['stitch_average']))
    assert np.max(np.abs(got - want) / np.abs(want)) < 2e-5


def test_ref_legacy_sensor_session(golden):
    """f3: the reference's own spectrum_sensor methods through a scripted session (spectrum_sensor.py:73-206) - the
    restated fast_spectrum_scan and PAPR give the PDUs it published; the second scan starts from the first one's
    noise estimate."""
    g = golden('ref_legacy_sensor.npz')
    x = golden(str(g['input_from']))['x'].astype(np.complex128)
    Sf, N, cs, sbw = int(g['sample_rate']), int(g['fft_len']), float(g['channel_space']), float(g['search_bw'])
    for method in ('welch', 'fft'):
        L = int(g[method + '_block_length'])
        ne = 1e-11
        for k, (off, tune, lev) in enumerate(((0, float(g['tune_freq']), int(g['thr_leveler'])),
                                              (int(g['second_offset']), float(g['tune_freq2']), int(g['thr_leveler2'])))):
            thr, plc, ne, cons = R.fast_spectrum_scan(x[off:off + L], tune, cs, sbw, N, Sf, method, lev, ne,
                                                      float(g['alpha_avg']))
            assert np.isclose(thr, g[method + '_thre'][k], rtol=1e-9) and np.isclose(ne, g[method + '_nois'][k], rtol=1e-9)
            ax = R.frange(tune - Sf / 2, tune + Sf / 2, cs)
            assert [1.0 if a in cons else 0.0 for a in ax] == list(g[method + '_cons%d' % k])
        v = x[:L]
        papr = 10 * np.log10((max(v * np.conjugate(v)) / (np.vdot(v, v) / len(v))).real + 1e-20)
        assert np.isclose(papr, float(g[method + '_papr']), rtol=1e-12)
        log = bytes(g[method + '_log']).decode('ascii').splitlines()
        assert [ln.split(',')[2] for ln in log] == ['tune_freq[Hz]', 'threshold[dB]', 'noise[dB]', 'spectrum_constraint[Hz]',
                                                    'tune_freq', 'papr', 'received unknown request', 'set_tune_freq',
                                                    'set_thr_leveler', 'tune_freq[Hz]', 'threshold[dB]', 'noise[dB]',
                                                    'spectrum_constraint[Hz]']


def test_ref_flank_detector(golden):
    """f4: the reference's own _queue0_watcher.flank_detector (flanck_detector.py:345-399) row by row - clipped
    peak-tracking power, the AGC step to thr2, rising / falling edge flags, alpha reset, noise estimate, edge counts."""
    g = golden('ref_flank.npz')
    st = R.FlankState(int(g['fft_len']), int(g['sample_rate']), float(g['channel_space']), float(g['search_bw']),
                      [float(c) for c in g['subject_channels']], trunc_band=int(g['sample_rate']),
                      thr_leveler=int(g['thr_leveler']), alpha_avg=float(g['alpha_avg']), peak_alpha=float(g['peak_alpha']))
    for i, r in enumerate(g['rows']):
        st.detect(r)
        assert np.allclose(st.curr_power, g['curr_power_seq'][i], rtol=1e-12, atol=0)
        assert [1.0 if f else 0.0 for f in st.flag] == list(g['flag_seq'][i])
        assert np.array_equal(st.peak_alpha, g['peak_alpha_seq'][i])
        assert np.isclose(st.noise_estimate, g['noise_seq'][i], rtol=1e-12, atol=0)
    assert sorted(st.cumulative_statistics) == list(g['stat_channels'])
    assert [st.cumulative_statistics[k] for k in sorted(st.cumulative_statistics)] == list(g['stat_counts'])
    assert np.allclose(R.chain_sensor_v2(g['x'], int(g['fft_len'])), g['rows'], rtol=1e-6)


def test_ref_fft_plot_and_time_domain_power(golden):
    """clc_power_time (ofdm_cr_tools.py:144-146), td_power_estimate (:337-339), fft_plot_dB (:312-319), fft_plot_lin
    (:328-335): the reference's own bodies on a vector of nfft samples, a shorter one (zero-padded by fft()) and a longer
    one (truncated by fft(), normalised by its full length)."""
    g = golden('ref_fft_plot.npz')
    x = golden(str(g['input_from']))['x']
    Sf, fc, nfft = int(g['Sf']), float(g['fc']), int(g['nfft'])
    for tag in ('exact', 'short', 'long'):
        lo, hi = g[tag + '_range']
        v = x[lo:hi]
        ax, lin = R.fft_plot_lin(v, Sf, fc, nfft)
        ax2, db = R.fft_plot_dB(v, Sf, fc, nfft)
        assert ax == ax2 and np.array_equal(ax, g[tag + '_axis'])
        assert relerr(lin, g[tag + '_lin']) < RTOL and np.max(np.abs(np.array(db) - g[tag + '_db'])) < 1e-8
        assert np.isclose(R.clc_power_time(v), float(g[tag + '_power_time']), rtol=1e-12)
        assert np.isclose(R.td_power_estimate(v, Sf), float(g[tag + '_td_power']), rtol=1e-12)
