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
