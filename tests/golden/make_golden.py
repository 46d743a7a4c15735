#!/usr/bin/env python3
"""Generate the golden fixtures in this directory.

The reference (Python 2 + GNU Radio 3.7) cannot be imported, so the vectors are
made by calling the third-party libraries that hold its arithmetic - SciPy and
NumPy, installed in the build container (scipy 1.15.3 / numpy 2.2.6) - with the
exact argument patterns of the reference's call sites:

  sg.welch(x, fs=Sf, nperseg=nFFT, nfft=nFFT)                  ofdm_cr_tools.py:322,342
  sg.welch(x, window='flattop', fs=Sf, nperseg=nFFT, nfft=nFFT) ofdm_cr_tools.py:214
  sg.welch(x, window='flattop', fs=Sf, nperseg=nFFT/4.0, nfft=nFFT)  spectrum_sweeper.py:263
  np.fft.fft / fftshift / ifft / np.convolve                   ofdm_cr_tools.py:151-170,177

Inputs are complex64 as GNU Radio delivers them; expected outputs are computed
from the float64 promotion of the same samples (SciPy keeps complex64 in single
precision, which would put its own rounding into the expectation).  The oracle
restatement (oracle/ref_cpu.py) does NOT take part in producing ``expected_*``
arrays that are tagged scipy/numpy below; rows that have no library call behind
them (GNU Radio chains, scanner state machines, framing) are produced by the
restatement and tagged ``restated`` - they pin regressions, not the reference.

Rows tagged ``reference`` (files ``ref_*.npz``) are different: ``reference_fixtures()`` runs the
reference's OWN function bodies - cut out of /root/reference/python/{ofdm_cr_tools,spectrum_sweeper}.py at
generation time by ``ref_extract.py`` and exec'd against the real numpy / scipy.signal - on inputs for which
Python-2 and Python-3 ``/`` agree (float or even-int Sf).  They pin rows a4, a6, a7 and a14 to the reference
itself.  ``xcorr`` / ``fac`` (a12, ``len(h)/2`` index), ``src_power_fft`` (``sg.flattop``) and ``make_plot`` run
unmodified in the namespace of the reference's day (ref_extract.py); the thread bodies and packers that hold
Python-2 print statements (a5, a8 in full, a10, a15, f1) run after lib2to3's print fixer and nothing else
(``reference_thread_fixtures`` -> ref_threads.npz); the GNU Radio chains a1-a3 stay unpinned.  Only numbers are
written; no reference text is stored in any form.

Run from the repo root:  python tests/golden/make_golden.py            (everything)
                         python tests/golden/make_golden.py --reference  (only the ref_*.npz files)
                         python tests/golden/make_golden.py --consumer   (only fragments_consumer.bin)
"""
import os
import sys

import numpy as np
import scipy
import scipy.fft as sfft
import scipy.signal as sg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import ref_cpu as R  # noqa: E402

META = dict(scipy=scipy.__version__, numpy=np.__version__)


def save(name, **kw):
    kw['versions'] = np.array(repr(META))
    np.savez_compressed(os.path.join(HERE, name), **kw)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in kw.items()})


def reference_fixtures():
    """ref_*.npz: outputs of the reference's own functions (see the module docstring)."""
    import warnings
    warnings.simplefilter('ignore')
    sys.path.insert(0, HERE)
    import ref_extract as E
    if not E.available():
        raise SystemExit('/root/reference is not present: the reference-tagged fixtures can only be made in '
                         'the build container')
    T, S = E.cr_tools(), E.sweeper()
    c128 = lambda v: np.asarray(v).astype(np.complex128)      # noqa: E731

    # a6 - welch_plot_dB (:321-326) and welch_power_estimate (:341-345): default Hann, 50 %, nperseg = nfft
    x = np.load(os.path.join(HERE, 'welch_hann_4096_50.npz'))['x']
    Sf, fc, nfft = 2000000, 433.0e6, 4096
    axis, db = T['welch_plot_dB'](c128(x), Sf, fc, nfft)
    save('ref_welch_hann_4096.npz', source=np.array('reference'), input_from=np.array('welch_hann_4096_50.npz'),
         Sf=Sf, fc=fc, nfft=nfft, expected_axis=np.array(axis), expected_db=np.array(db),
         expected_power=T['welch_power_estimate'](c128(x), nfft, Sf),
         expected_power_fs1=T['welch_power_estimate'](c128(x), nfft, 1.0),
         expected_clc_power_freq=T['clc_power_freq'](c128(x[:4096]), 4096, Sf))

    # a6 / a14 - src_power_welch (:213-230) and fast_spectrum_scan(method='welch') (:471-537)
    x = np.load(os.path.join(HERE, 'welch_flattop_2048.npz'))['x']
    Sf, N, cs, sbw = 1000000, 2048, 50e3, 25e3
    Fr = float(Sf) / N
    bb = T['frange'](-Sf / 2, Sf / 2, cs)
    psd, ax, plc = T['src_power_welch'](c128(x), len(x), N, Fr, Sf, bb, sbw / Fr)
    scans, ne = [], 1e-11
    for lo, hi in ((0, 16384), (8192, 32768), (0, 32768)):      # the noise estimate carries over between scans
        thr, plc_s, ne, occ = T['fast_spectrum_scan'](c128(x[lo:hi]), 100.0e6, cs, sbw, N, Sf, 'welch', 4, ne,
                                                     0.5, False)
        scans.append((thr, plc_s, ne, occ))
    ax_ch = T['frange'](100.0e6 - Sf / 2, 100.0e6 + Sf / 2, cs)
    save('ref_src_power_welch_2048.npz', source=np.array('reference'),
         input_from=np.array('welch_flattop_2048.npz'), Sf=Sf, nfft=N, channel_rate=cs, srch_bw=sbw,
         bb_freqs=np.array(bb), expected_psd=np.array(psd), expected_axis=np.array(ax), expected_plc=np.array(plc),
         scan_ranges=np.array([(0, 16384), (8192, 32768), (0, 32768)]), scan_fc=100.0e6, scan_thr_leveler=4,
         scan_alpha=0.5, scan_noise0=1e-11, ax_ch=np.array(ax_ch),
         scan_thr=np.array([s[0] for s in scans]), scan_plc=np.array([s[1] for s in scans]),
         scan_noise=np.array([s[2] for s in scans]),
         scan_occupied=np.array([[1.0 if a in s[3] else 0.0 for a in ax_ch] for s in scans]))

    # a4 - spectrum_sweeper._src_power (:260-276) and the sweeper's inclusive frange (:37-42)
    g = np.load(os.path.join(HERE, 'welch_flattop_nperseg_quarter.npz'))
    x, nfft, ex, fs = g['x'], int(g['nfft']), int(g['excess_bins']), float(g['fs'])
    save('ref_sweeper_src_power.npz', source=np.array('reference'),
         input_from=np.array('welch_flattop_nperseg_quarter.npz'), nfft=nfft, excess_bins=ex, fs=fs,
         expected_db=T_db(S['_src_power'](c128(x), nfft, fs, ex)),
         expected_db_notrim=T_db(S['_src_power'](c128(x[:9000]), 1024, 250000.0, 0)),
         frange_le_a=np.array(S['frange'](88.0e6 + 1.0e6, 108.0e6, 2.0e6)),
         frange_le_b=np.array(S['frange'](0, 1, 0.25)), frange_le_c=np.array(S['frange'](0.0, 1.0, 0.1)))

    # a7 - frange (:136-141), movingaverage (:168-170), src_power (:232-249) on the committed PSD cases
    g = np.load(os.path.join(HERE, 'src_power_cases.npz'))
    out = {}
    for i in range(int(g['n'])):
        Sf, N = int(g['Sf_%d' % i]), int(g['N_%d' % i])
        cs, sbw = float(g['cs_%d' % i]), float(g['sbw_%d' % i])
        psd = g['psd_%d' % i]
        Fr = float(Sf) / N
        bb = T['frange'](-Sf / 2, Sf / 2, cs)          # even-int Sf: Python-2 and Python-3 '/' agree
        sb = sbw / Fr
        out['bb_%d' % i] = np.array(bb)
        out['ma_%d' % i] = T['movingaverage'](psd, 1 * sb)
        out['plc_%d' % i] = np.array(T['src_power'](psd, N, Fr, Sf, bb, sb))
        out['plc_f32_%d' % i] = np.array(T['src_power'](psd.astype(np.float32), N, Fr, Sf, bb, sb))
    save('ref_src_power_cases.npz', source=np.array('reference'), input_from=np.array('src_power_cases.npz'),
         n=int(g['n']), frange_a=np.array(T['frange'](0, 1, 0.25)), frange_b=np.array(T['frange'](0, 1, 0.1)),
         frange_c=np.array(T['frange'](-500000.0, 500000.0, 25e3)), **out)


    # a11 - coherence_detector.watcher.scanner (:254-274) + find_nearest_index (:276-278), run on a stand-in
    # object that carries the attributes the method reads (its __init__ needs gnuradio)
    from types import SimpleNamespace as NS
    g = np.load(os.path.join(HERE, 'coherence_scanner.npz'))
    scanner = E.load_method('coherence_detector.py', 'watcher', 'scanner')
    fni = E.load('coherence_detector.py', ['find_nearest_index'])['find_nearest_index']
    N, Sf, tune = int(g['N']), int(g['sample_rate']), int(g['tune_freq'])
    ax = np.array(range(-N // 2, N // 2)) * (float(Sf) / N) + tune          # coherence_detector.py:188
    idx = [int(fni(ax, c)) for c in g['subject_channels']]
    valve, outcome, sent = [], [], []
    me = NS(n_chans=len(idx), idx_subject_channels=idx, subject_channels_coherence=[0] * len(idx), threshold=10,
            threshold_mtm=0.2, valve_callback=valve.append, set_subject_channels_outcome=outcome.append,
            data_queue0=NS(put=lambda v: sent.append(list(v))))
    scanner(me, g['d0'], g['d1'], g['d2'])
    save('ref_coherence_scanner.npz', source=np.array('reference'), input_from=np.array('coherence_scanner.npz'),
         idx=np.array(idx), threshold=10, threshold_mtm=0.2, coherence=np.array(sent[0], np.float64),
         outcome=np.array(outcome[0]), valve=np.array(valve))

    # a8 (truncation + 0.6/0.4 EMA) and a9 (top-4): spectrum_sensor_v2.basic_spectrum_watcher.spectrum_scanner
    # (:533-544), spectrum_sensor_v2.output_data.publish (:228-237), multichannel_scanner
    # basic_spectrum_watcher.spectrum_scanner / publish (:214-239) over the 16 committed PSD rows.  The stand-in's
    # attributes follow the classes' __init__ lines (spectrum_sensor_v2.py:368-385) with Python-2 integer '/'.
    g = np.load(os.path.join(HERE, 'scanner_state_seq.npz'))
    fft_len, Sf, cs, sbw, tune, trunc_band = 1024, 1000000, 25e3, 12.5e3, 100000000, 800000
    Fr = float(Sf) / fft_len
    trunc = Sf - trunc_band
    trunc_ch = int(trunc / cs) // 2
    ax_ch = T['frange'](tune - Sf // 2, tune + Sf // 2, cs)[trunc_ch:-trunc_ch]
    assert np.allclose(ax_ch, g['ax_ch'])
    sent = []
    me = NS(fft_len=fft_len, Fr=Fr, sample_rate=Sf, bb_freqs=T['frange'](-Sf // 2, Sf // 2, cs), srch_bins=sbw / Fr,
            trunc=trunc, trunc_ch=trunc_ch, plc=np.array([0.0] * len(ax_ch)),
            data_queue=NS(put=lambda v: sent.append(np.array(v))))
    scan_v2 = E.load_method('spectrum_sensor_v2.py', 'basic_spectrum_watcher', 'spectrum_scanner',
                            {'src_power': T['src_power']})
    scan_mc = E.load_method('multichannel_scanner.py', 'basic_spectrum_watcher', 'spectrum_scanner',
                            {'src_power': T['src_power']})
    me_mc = NS(**{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in vars(me).items()})
    for r in g['rows']:
        scan_v2(me, r.astype(np.float32))
        scan_mc(me_mc, r.astype(np.float32))
    assert np.array_equal(me.plc, me_mc.plc)
    subj = [float(v) for v in g['subject_channels']]
    idx = [ax_ch.index(c) for c in subj]                                      # spectrum_sensor_v2.py:218-220
    freqs = []
    od = NS(data_queue=NS(get=lambda: sent[-1]), idx_subject_channels=idx, subject_channels=subj,
            subject_channels_pwr=np.array([1.0] * len(subj)), set_freqs=lambda *f: freqs.append(list(f)))
    E.load_method('spectrum_sensor_v2.py', 'output_data', 'publish')(od)
    me_mc.idx_subject_channels, me_mc.subject_channels = idx, subj
    me_mc.subject_channels_pwr = np.array([1.0] * len(subj))
    me_mc.set_freqs = lambda *f: freqs.append(list(f))
    me_mc.output_data = NS()
    E.load_method('multichannel_scanner.py', 'basic_spectrum_watcher', 'publish')(me_mc)
    assert freqs[0] == freqs[1] == me_mc.output_data.top4
    save('ref_scanner_seq.npz', source=np.array('reference'), input_from=np.array('scanner_state_seq.npz'),
         plc_seq=np.array(sent), subject_pwr=od.subject_channels_pwr, top4=np.array(freqs[0]))


    # a12 - xcorr / fac (ofdm_cr_tools.py:155-166) and a14 - src_power_fft (:173-192), fast_spectrum_scan('fft')
    # (:471-537): function text untouched, run in the namespace of the reference's day (ref_extract.py: Python-2 `/` on
    # len(), scipy.signal's old window re-exports)
    P = E.load('ofdm_cr_tools.py', ['xcorr', 'fac', 'src_power_fft'], E.py2_namespace())
    g = np.load(os.path.join(HERE, 'xcorr_fac.npz'))
    L = int(g['L'])
    save('ref_xcorr_fac.npz', source=np.array('reference'), input_from=np.array('xcorr_fac.npz'), L=L,
         expected_xcorr=P['xcorr'](c128(g['a']), c128(g['b']), L), expected_fac=P['fac'](c128(g['a']), L),
         expected_xcorr_short=P['xcorr'](c128(g['a'][:700]), c128(g['b'][:900]), 1024))
    x = np.load(os.path.join(HERE, 'welch_flattop_2048.npz'))['x']
    Sf, N, cs, sbw = 1000000, 2048, 50e3, 25e3
    Fr = float(Sf) / N
    bb = T['frange'](-Sf / 2, Sf / 2, cs)
    psd, ax, plc = P['src_power_fft'](c128(x[:N]), N, N, Fr, Sf, bb, sbw / Fr)
    psd2, ax2, plc2 = P['src_power_fft'](c128(x[3000:4500]), 1500, N, Fr, Sf, bb, sbw / Fr)      # npts < nFFT: zero-padded
    scan = E.load('ofdm_cr_tools.py', ['fast_spectrum_scan'],
                  dict(E.py2_namespace(), frange=T['frange'], src_power_fft=P['src_power_fft'],
                       src_power_welch=T['src_power_welch']))['fast_spectrum_scan']
    thr, plc_s, ne, occ = scan(c128(x[:N]), 100.0e6, cs, sbw, N, Sf, 'fft', 4, 1e-11, 0.5, False)
    ax_ch = T['frange'](100.0e6 - Sf / 2, 100.0e6 + Sf / 2, cs)
    save('ref_src_power_fft.npz', source=np.array('reference'), input_from=np.array('welch_flattop_2048.npz'),
         Sf=Sf, nfft=N, channel_rate=cs, srch_bw=sbw, bb_freqs=np.array(bb), expected_psd=np.array(psd),
         expected_axis=np.array(ax), expected_plc=np.array(plc), short_range=np.array([3000, 4500]),
         expected_psd_short=np.array(psd2), expected_plc_short=np.array(plc2), scan_fc=100.0e6, scan_thr_leveler=4,
         scan_alpha=0.5, scan_noise0=1e-11, scan_thr=thr, scan_plc=np.array(plc_s), scan_noise=ne,
         scan_occupied=np.array([1.0 if a in occ else 0.0 for a in ax_ch]))

    # f4 - ascii_plotter.make_plot (ascii_plot.py:169-228), the reference's own method body on a stand-in object.
    # It was written for Python 2: ``self.matrix[self.width/2]`` needs an integer quotient.  The method text is NOT
    # touched; the stand-in's ``width`` is an int whose ``/`` by an int floors, which is what Python 2 did (its only
    # other ``/ self.width`` has a float on the left).  ``height`` stays a plain int: every ``/ self.height`` has a
    # float numerator in Python 2 (math.floor returned a float), i.e. true division then as now.  The row goes in as
    # Python floats: the NumPy of the reference's day summed float32 scalars in double through Python's sum().
    Py2Int = E.Py2Int
    make_plot = E.load_method('ascii_plot.py', 'ascii_plotter', 'make_plot')
    rng = np.random.default_rng(3)
    out = {}
    cases = ((64, 20, 2048, 2000000, 100.0e6), (50, 15, 1024, 1000000, 433.0e6), (33, 12, 256, 250000, 2.4e9),
             (80, 25, 4096, 2000000, 0.0))
    for i, (W, H, N, Sf, tf) in enumerate(cases):
        row = (rng.standard_normal(N) * 6 - 80 + 20 * np.exp(-((np.arange(N) - N * 0.7) / 15.0) ** 2)).astype(np.float32)
        axis = Sf // 2 * np.linspace(-1, 1, N) + tf                                          # :157 (even-int Sf)
        me = NS(width=Py2Int(W), height=H, tune_freq=tf, sample_rate=Sf, fft_len=N, axis=axis,
                widthDens=len(axis) // W, matrix=[[' ' for x in range(H)] for y in range(W)])
        text = make_plot(me, [float(v) for v in row])
        out['row_%d' % i] = row
        out['text_%d' % i] = np.frombuffer(text.encode('ascii'), np.uint8)
        out['case_%d' % i] = np.array([W, H, N, Sf, tf], np.float64)
    save('ref_ascii_plot.npz', source=np.array('reference'), n=len(cases), **out)
    reference_anylen_fixture(E, T, P, scan)
    reference_thread_fixtures(E, T, S)
    reference_legacy_sensor_fixture(E, T, scan)
    reference_flank_fixture(E, T)
    reference_fft_plot_fixture(E)
    reference_logger_fixture(E)
    reference_consumer_fixture(E)
    reference_psd_logger_fixture(E)


def reference_anylen_fixture(E, T, P, scan):
    """ref_anylen.npz (round 6): the reference's own bodies at the transform lengths it accepts and rounds 1-5 refused -
    fast_spectrum_scan(n_fft=0) choosing nFFT = 2^ceil(log2(npts)) itself (ofdm_cr_tools.py:474-475) on 20 000 and
    100 000 samples (32768 / 131072 points; 'welch': SciPy shortens nperseg to the input, one zero-padded flat-top
    segment; 'fft': one flat-top periodogram), a free-integer n_fft (1000, 3000: the web gateway's --nfft,
    sdr_webserver/local_hw_gateway.py:284-285), welch_plot_dB / welch_power_estimate at lengths that are not powers of
    two, xcorr / fac at an even and an odd length (Python-2 `len(h)/2`).  Inputs are R.synth_iq seeds (not stored); PSDs
    are kept every `stride`-th bin (+ their sum) so that the file stays small."""
    c128 = lambda v: np.asarray(v).astype(np.complex128)      # noqa: E731
    out, cases = {}, []
    Sf, cs, sbw, fc = 2000000, 100e3, 50e3, 433.0e6
    ax_ch = T['frange'](fc - Sf / 2, fc + Sf / 2, cs)
    i = 0
    for npts, n_fft, seed in ((20000, 0, 901), (100000, 0, 902), (20000, 1000, 903), (9000, 3000, 904)):
        x = R.synth_iq(npts, seed)
        for method in ('welch', 'fft'):
            nfft = n_fft or int(2 ** np.ceil(np.log2(npts)))
            Fr = float(Sf) / nfft
            bb = T['frange'](-Sf / 2, Sf / 2, cs)
            fn = T['src_power_welch'] if method == 'welch' else P['src_power_fft']
            psd, _, plc = fn(c128(x), npts, nfft, Fr, Sf, bb, sbw / Fr)
            thr, plc_s, ne, occ = scan(c128(x), fc, cs, sbw, n_fft, Sf, method, 4, 1e-11, 0.5, False)
            assert np.array_equal(np.array(plc), np.array(plc_s))
            stride = max(1, nfft // 2048)
            cases.append((npts, n_fft, seed, 0 if method == 'welch' else 1, nfft, stride))
            out['psd_%d' % i] = np.array(psd)[::stride]
            out['psd_sum_%d' % i] = float(np.sum(psd))
            out['plc_%d' % i] = np.array(plc_s)
            out['thr_%d' % i] = thr
            out['noise_%d' % i] = ne
            out['occupied_%d' % i] = np.array([1.0 if a in occ else 0.0 for a in ax_ch])
            i += 1
    x = R.synth_iq(30000, 905)
    _, db = T['welch_plot_dB'](c128(x), Sf, fc, 1000)
    out['plot_db_1000'] = np.array(db)
    out['power_6000'] = T['welch_power_estimate'](c128(x), 6000, Sf)
    out['power_1021'] = T['welch_power_estimate'](c128(x), 1021, Sf)
    out['power_short_40000'] = T['welch_power_estimate'](c128(x), 40000, Sf)      # nFFT > len(x): SciPy's shortened nperseg
    a, b = R.synth_iq(900, 906, tones=(), dc=0), R.synth_iq(700, 907, tones=(), dc=0)
    for L in (1000, 1001, 20000):
        out['xcorr_%d' % L] = P['xcorr'](c128(a), c128(b), L)
        out['fac_%d' % L] = P['fac'](c128(a), L)
    save('ref_anylen.npz', source=np.array('reference'), Sf=Sf, channel_rate=cs, srch_bw=sbw, fc=fc, thr_leveler=4,
         alpha=0.5, noise0=1e-11, cases=np.array(cases), ax_ch=np.array(ax_ch), plot_seed=905, plot_n=30000,
         xcorr_seeds=np.array([906, 907]), xcorr_lens=np.array([900, 700]), **out)


def reference_thread_fixtures(E, T, S):
    """ref_threads.npz: the reference's own thread bodies and packers, run on stand-in queues / receivers / ports
    (ref_extract.load_method(..., py2_print=True): the only change to their text is lib2to3's print fixer).
      a8   stats_watcher.spectrum_scanner (spectrum_sensor_v2.py:445-479): EMA, cumulative / periodic max, noise
           estimate, threshold, occupied channels - over the 16 committed PSD rows;
      a10  psd_watcher.run (:336-354) and waterfall_watcher.run (:304-322): last vector of a message wins, running peak,
           waterfall rows;
      a15  main_thread.run (local_worker.py:126-139) and data_colector.run (spectrum_sweeper.py:161-172): the same
           last-vector rule in front of the packer / the sample store;
      a5   spectrum_stitcher.run (spectrum_sweeper.py:207-231): retune order, tune delay, concatenation, blend, pack;
      f1   packet_source.send_packet of local_worker.py:147-172 (float32 and int8) and spectrum_sweeper.py:240-258."""
    import io
    import math
    import struct
    from contextlib import redirect_stdout
    from types import SimpleNamespace as NS
    npd = E.NumpyOfItsDay()
    out = {}
    g = np.load(os.path.join(HERE, 'scanner_state_seq.npz'))
    rows = g['rows'].astype(np.float32)

    # ---- a8: the stats watcher's scanner over the committed rows, constructor values as in make_golden.main()
    fft_len, Sf, cs, sbw, tune, trunc_band = 1024, 1000000, 25e3, 12.5e3, 100000000, 800000
    Fr = float(Sf) / fft_len
    trunc = Sf - trunc_band
    trunc_ch = int(trunc / cs) // 2
    ax_ch = T['frange'](tune - Sf // 2, tune + Sf // 2, cs)[trunc_ch:-trunc_ch]
    sent = []

    class Logger(object):      # ofdm_cr_tools.py:1878-1894 initial values, :1932-1938 setters
        cumulative_max_power = None
        periodic_max_power = None

        def set_cumulative_max_power(self, v):
            self.cumulative_max_power = v

        def set_periodic_max_power(self, v):
            self.periodic_max_power = v

    me = NS(fft_len=fft_len, Fr=Fr, sample_rate=Sf, bb_freqs=T['frange'](-Sf // 2, Sf // 2, cs), srch_bins=sbw / Fr,
            trunc=trunc, trunc_ch=trunc_ch, plc=np.array([0.0] * len(ax_ch)), ax_ch=ax_ch, noise_estimate=1e-11,
            alpha_avg=0.5, thr_leveler=4, verbose=True, logger=Logger(),
            data_queue=NS(put=lambda v: sent.append(np.array(v))))
    scan = E.load_method('spectrum_sensor_v2.py', 'stats_watcher', 'spectrum_scanner',
                         {'np': npd, 'src_power': T['src_power']}, py2_print=True)
    occ, noise, said = [], [], io.StringIO()
    with redirect_stdout(said):
        for r in rows:
            hz = scan(me, r)
            occ.append(np.array([1.0 if a in hz else 0.0 for a in ax_ch]))
            noise.append(me.noise_estimate)
    assert said.getvalue().count('noise_estimate dB (channel)') == len(rows)
    out.update(stats_plc_seq=np.array(sent), stats_occupied_seq=np.array(occ), stats_noise_seq=np.array(noise),
               stats_cumulative_max=np.array(me.logger.cumulative_max_power),
               stats_periodic_max=np.array(me.logger.periodic_max_power))

    # ---- a10 / a15: messages of 1, 3, 1, 2, 1, 4, 1, 3 vectors; the queue ends the loop with its last message
    counts = (1, 3, 1, 2, 1, 4, 1, 3)
    assert sum(counts) == len(rows)

    def messages(vectors, dtype):
        msgs, k = [], 0
        for c in counts:
            body = np.ascontiguousarray(vectors[k:k + c], dtype).tobytes()
            itemsize = len(body) // c
            msgs.append(NS(arg1=lambda i=itemsize: float(i), arg2=lambda c=c: float(c), to_string=lambda b=body: b))
            k += c
        return msgs

    def queue_for(owner, msgs):
        pending = list(msgs)

        def delete_head():
            m = pending.pop(0)
            if not pending:
                owner.keep_running = False
            return m
        return NS(delete_head=delete_head)

    class PsdLogger(object):      # ofdm_cr_tools.py:1878-1906 initial values and setters
        cumulative_psd = None
        periodic_psd_peaks = None
        cumulative_waterfall = None

        def set_cumulative_psd(self, v):
            self.cumulative_psd = v

        def set_periodic_psd_peaks(self, v):
            self.periodic_psd_peaks = v

        def set_cumulative_waterfall(self, v):
            self.cumulative_waterfall = v

    lg = PsdLogger()
    lg.cumulative_waterfall = []
    w = NS(keep_running=True, logger=lg)
    w.rcvd_data = queue_for(w, messages(rows, np.float32))
    E.load_method('spectrum_sensor_v2.py', 'psd_watcher', 'run', {'np': npd}, py2_print=True)(w)
    w = NS(keep_running=True, logger=lg)
    w.rcvd_data = queue_for(w, messages(rows, np.float32))
    E.load_method('spectrum_sensor_v2.py', 'waterfall_watcher', 'run', {'np': npd}, py2_print=True)(w)
    out.update(msg_counts=np.array(counts), psd_cumulative=np.array(lg.cumulative_psd),
               psd_periodic_peaks=np.array(lg.periodic_psd_peaks), waterfall=np.array(lg.cumulative_waterfall))

    handed = []
    w = NS(keep_running=True, max_tu=1470, fft_len=fft_len, data_precision=True,
           packet_source=NS(send_packet=lambda data, max_tu, n, prec: handed.append(np.frombuffer(data, np.float32).copy())))
    w.rcvd_data = queue_for(w, messages(rows, np.float32))
    with redirect_stdout(io.StringIO()):
        E.load_method('local_worker.py', 'main_thread', 'run', {'np': npd}, py2_print=True)(w)
    out['worker_vectors'] = np.array(handed)
    iq = g['x'].astype(np.complex64).reshape(len(rows), -1)
    stored = []
    w = NS(keep_running=True, set_samples=lambda v: stored.append(np.array(v)))
    w.rcvd_data = queue_for(w, messages(iq, np.complex64))
    with redirect_stdout(io.StringIO()):
        E.load_method('spectrum_sweeper.py', 'data_colector', 'run', {'np': npd}, py2_print=True)(w)
    out['collector_vectors'] = np.array(stored)

    # ---- f1: both packers on the vector of fragments.bin; a stand-in port collects the frames
    class Port(object):
        def __init__(self):
            self.frames = []

        def message_port_pub(self, port, pdu):
            assert port == 'out' and pdu[0] is None
            self.frames.append(bytes(pdu[1]))

    pmt = NS(make_u8vector=lambda n, fill: bytearray([fill]) * n, PMT_NIL=None, intern=lambda s: s,
             u8vector_set=lambda v, i, val: v.__setitem__(i, val), cons=lambda a, b: (a, b))
    ns = {'np': npd, 'pmt': pmt, 'struct': struct, 'math': math, 'ord': E.py2_ord, 'len': lambda v: E.Py2Int(len(v))}
    worker_send = E.load_method('local_worker.py', 'packet_source', 'send_packet', ns)
    sweeper_send = E.load_method('spectrum_sweeper.py', 'packet_source', 'send_packet', ns)
    db = (np.arange(4096, dtype=np.float32) * 0.01 - 90).astype('<f4')
    groups = []
    for prec in (True, False):
        port = Port()
        worker_send(port, db.tobytes(), 1472 - 2, 4096, prec)
        groups.append(port.frames)
    port = Port()
    sweeper_send(port, db.tobytes(), 1472 - 2)
    groups.append(port.frames)
    blob = b''
    for group in groups:
        blob += np.uint32(len(group)).tobytes()
        for fr in group:
            blob += np.uint32(len(fr)).tobytes() + fr
    with open(os.path.join(HERE, 'fragments.bin'), 'rb') as fh:
        assert fh.read() == blob, 'fragments.bin (restated) differs from the reference packers'
    out['fragments_bin'] = np.frombuffer(blob, np.uint8)
    # lengths around the fragment boundaries: the worker's ceil and the sweeper's floor + 1 (one empty frame at multiples)
    for n in (1, 367, 368, 735, 736, 1104):
        v = (np.arange(n, dtype=np.float32) * 0.5 - 70).astype('<f4')
        port = Port()
        worker_send(port, v.tobytes(), 1472, n, True)
        out['worker_frames_%d' % n] = np.frombuffer(b''.join(np.uint32(len(f)).tobytes() + f for f in port.frames), np.uint8)
        port = Port()
        sweeper_send(port, v.tobytes(), 1472)
        out['sweeper_frames_%d' % n] = np.frombuffer(b''.join(np.uint32(len(f)).tobytes() + f for f in port.frames), np.uint8)
    out['frame_lengths'] = np.array([1, 367, 368, 735, 736, 1104])

    # ---- a5: one pass of the stitcher's loop over three tuned captures of the committed flattop input
    gq = np.load(os.path.join(HERE, 'welch_flattop_nperseg_quarter.npz'))
    xq = gq['x'].astype(np.complex64)
    nfft_s, fs_s, excess = int(gq['nfft']), float(gq['fs']), 96
    third = len(xq) // 3
    captures = [xq[i * third:(i + 1) * third] for i in range(3)]
    freqs = [100.0e6, 101.5e6, 103.0e6]
    tuned, slept, packed = [], [], []
    pending = list(captures)
    st = NS(keep_running=True, tune_frequencies=freqs, fft_len=nfft_s, sample_rate=fs_s, excess_bins=excess,
            tune_delay=0.125, average=0.25, max_tu=1472, get_samples=lambda: pending.pop(0),
            rf_receiver=NS(set_center_freq=lambda f, ch: tuned.append((f, ch))))

    def stitcher_send(data, max_tu):
        packed.append((data, max_tu))
        st.keep_running = False
    st.packet_source = NS(send_packet=stitcher_send)
    run = E.load_method('spectrum_sweeper.py', 'spectrum_stitcher', 'run',
                        {'np': npd, 'struct': struct, 'time': NS(sleep=slept.append), '_src_power': S['_src_power']},
                        py2_print=True)
    with redirect_stdout(io.StringIO()):
        run(st)
    assert len(packed) == 1 and packed[0][1] == 1472 and not pending
    out.update(stitch_captures=np.array(captures), stitch_nfft=nfft_s, stitch_fs=fs_s, stitch_excess=excess,
               stitch_average=0.25, stitch_freqs=np.array(freqs), stitch_tuned=np.array(tuned, np.float64),
               stitch_sleeps=np.array(slept), stitch_packed=np.frombuffer(packed[0][0], np.uint8))
    save('ref_threads.npz', source=np.array('reference'), input_from=np.array('scanner_state_seq.npz'), **out)


def reference_legacy_sensor_fixture(E, T, scan):
    """ref_legacy_sensor.npz (f3): the reference's own spectrum_sensor methods (spectrum_sensor.py:73-206: work,
    cogeng_rx, send_msg, set_papr, set_spectrum_constraint_hz, the logging setters) bound to a class without its
    GNU Radio base, driven through one scripted session with a fixed clock; `pmt` is a stand-in where a PDU is a
    (meta, data) tuple.  Recorded: every PDU the block publishes and the text of its request log."""
    import io
    from contextlib import redirect_stdout
    from types import SimpleNamespace as NS
    names = ['work', 'cogeng_rx', 'send_msg', 'set_spectrum_constraint_hz', 'get_spectrum_constraint_hz',
             'get_threshold', 'get_noise_estimate', 'get_power_level_ch', 'set_papr', 'get_papr', 'set_vector_sample',
             'get_vector_sample', 'set_sample_rate', 'set_tune_freq', 'get_tune_freq', 'set_channel_space',
             'set_search_bw', 'set_thr_leveler', 'get_alpha_avg']
    pmt = NS(car=lambda m: m[0], cdr=lambda m: m[1], to_python=lambda v: v, to_pmt=lambda v: v,
             cons=lambda a, b: (a, b), intern=lambda s: s)
    clock = NS(strftime=lambda fmt: {'%H%M%S': '120000', '%y%m%d': '140101'}[fmt])
    Ref = E.load_methods('spectrum_sensor.py', 'spectrum_sensor', names,
                         {'np': np, 'pmt': pmt, 'time': clock, 'fast_spectrum_scan': scan}, py2_print=True)
    x = np.load(os.path.join(HERE, 'welch_flattop_2048.npz'))['x']
    out = {}
    for method in ('welch', 'fft'):
        pdus = []
        me = Ref()
        me.block_length, me.sample_rate, me.fft_len = (8192 if method == 'welch' else 2048), 1000000, 2048
        me.channel_space, me.search_bw, me.method, me.thr_leveler = 50e3, 25e3, method, 4
        me.tune_freq, me.vector_sample, me.papr, me.spectrum_constraint_hz = 100.0e6, [0, 0], 1e-10, []
        me.threshold, me.power_level_ch, me.noise_estimate, me.alpha_avg = 0, [], 1e-11, 0.5
        me.log, me.log_file = True, io.StringIO()
        me.message_port_pub = lambda port, pdu, pdus=pdus: pdus.append((port,) + tuple(pdu))
        with redirect_stdout(io.StringIO()):
            assert me.work([x.astype(np.complex128)], []) == me.block_length
            me.cogeng_rx(({}, 'SC'))
            me.cogeng_rx(({}, 'PAPR'))
            me.cogeng_rx(({}, 'bogus'))
            me.set_tune_freq(101.0e6)
            me.set_thr_leveler(6)
            assert me.work([x[5000:].astype(np.complex128)], []) == me.block_length
            me.cogeng_rx(({}, 'SC'))                     # the noise estimate carries over (alpha_avg = 0.5)
            me.cogeng_rx('not a pdu'[:0])                # pmt.car of a non-PDU: "Message is not a valid PDU", no reply
        assert all(p[0] == 'PDU spect_msg' for p in pdus)
        metas = [p[1] for p in pdus]
        assert metas == ['thre', 'nois', 'cons', 'papr', 'unkn', 'thre', 'nois', 'cons']
        ax_ch = T['frange'](100.0e6 - 1000000 / 2, 100.0e6 + 1000000 / 2, 50e3)
        ax_ch2 = T['frange'](101.0e6 - 1000000 / 2, 101.0e6 + 1000000 / 2, 50e3)
        out[method + '_metas'] = np.array(metas)
        out[method + '_thre'] = np.array([pdus[0][2], pdus[5][2]])
        out[method + '_nois'] = np.array([pdus[1][2], pdus[6][2]])
        out[method + '_cons0'] = np.array([1.0 if a in pdus[2][2] else 0.0 for a in ax_ch])
        out[method + '_cons1'] = np.array([1.0 if a in pdus[7][2] else 0.0 for a in ax_ch2])
        assert len(pdus[2][2]) == int(out[method + '_cons0'].sum()) and len(pdus[7][2]) == int(out[method + '_cons1'].sum())
        out[method + '_papr'] = np.array(pdus[3][2])
        out[method + '_unkn'] = np.array(pdus[4][2])
        out[method + '_log'] = np.frombuffer(me.log_file.getvalue().encode('ascii'), np.uint8)
        out[method + '_block_length'] = np.array(me.block_length)
    save('ref_legacy_sensor.npz', source=np.array('reference'), input_from=np.array('welch_flattop_2048.npz'),
         second_offset=5000, sample_rate=1000000, fft_len=2048, channel_space=50e3, search_bw=25e3, thr_leveler=4,
         thr_leveler2=6, tune_freq=100.0e6, tune_freq2=101.0e6, alpha_avg=0.5, **out)


def reference_flank_fixture(E, T):
    """ref_flank.npz (f4): the reference's own _queue0_watcher.flank_detector (flanck_detector.py:345-399) over 30 PSD
    rows in which a tone switches on in one subject channel for ten rows - peak-tracking power, flags, alpha, noise
    estimate per row and the edge counts.  The rows are the rectangular |FFT|^2 / N^2 chain of the committed IQ."""
    import io
    from contextlib import redirect_stdout
    from types import SimpleNamespace as NS
    N, Sf, cs, sbw = 1024, 1024000, 32000.0, 16e3
    nvec = 30
    x = R.synth_iq(N * nvec, 71, tones=(), dc=0)
    t = np.arange(N * nvec)
    Fr = float(Sf) / N
    ax_ch = T['frange'](0 - Sf // 2, 0 + Sf // 2, cs)               # trunc_band = Sf: no truncation
    subj = [ax_ch[10], ax_ch[20]]
    gate = ((t // N >= 8) & (t // N < 18)).astype(np.float64)
    x = (x + 3.0 * gate * np.exp(2j * np.pi * (ax_ch[20] / Sf) * t)).astype(np.complex64)
    rows = R.chain_sensor_v2(x, N).astype(np.float32)

    class Logger(object):      # flanck_detector.py:38-128: the dictionaries and their setters
        def __init__(self):
            self.cumulative_statistics, self.periodic_statistic, self.settings = {}, {}, {'n_measurements': 0}
            self.n_measurements_period = 0

        def set_settings(self, v):
            self.settings = v

        def set_n_measurements_period(self, v):
            self.n_measurements_period = v

        def set_cumulative_statistics(self, v):
            self.cumulative_statistics = v

        def set_periodic_statistic(self, v):
            self.periodic_statistic = v

    n = len(subj)
    me = NS(fft_len=N, Fr=Fr, sample_rate=Sf, bb_freqs=T['frange'](-Sf // 2, Sf // 2, cs), srch_bins=sbw / Fr, trunc=0,
            trunc_ch=0, noise_estimate=1e-11, alpha_avg=0.2, thr_leveler=4, verbose=True, ax_ch=ax_ch,
            subject_channels=subj, idx_subject_channels=[ax_ch.index(c) for c in subj],
            prev_power=np.array([1.0] * n), curr_power=np.array([1.0] * n), flag=[True] * n,
            peak_alpha=np.array([0.0] * n), peak_alpha_original=0.5, logger=Logger())
    detect = E.load_method('flanck_detector.py', '_queue0_watcher', 'flank_detector',
                           {'np': np, 'src_power': T['src_power']}, py2_print=True)
    curr, flags, alphas, noise = [], [], [], []
    with redirect_stdout(io.StringIO()):
        for r in rows:
            detect(me, r)
            curr.append(me.curr_power.copy())
            flags.append([1.0 if f else 0.0 for f in me.flag])
            alphas.append(me.peak_alpha.copy())
            noise.append(me.noise_estimate)
    stats = me.logger.cumulative_statistics
    assert stats == me.logger.periodic_statistic and stats.get(subj[1], 0) >= 1
    save('ref_flank.npz', source=np.array('reference'), x=x, rows=rows, fft_len=N, sample_rate=Sf, channel_space=cs,
         search_bw=sbw, thr_leveler=4, alpha_avg=0.2, peak_alpha=0.5, subject_channels=np.array(subj),
         curr_power_seq=np.array(curr), flag_seq=np.array(flags), peak_alpha_seq=np.array(alphas),
         noise_seq=np.array(noise), stat_channels=np.array(sorted(stats)), stat_counts=np.array([stats[k] for k in sorted(stats)]))


def reference_fft_plot_fixture(E):
    """ref_fft_plot.npz: clc_power_time (ofdm_cr_tools.py:144-146), td_power_estimate (:337-339), fft_plot_dB (:312-319)
    and fft_plot_lin (:328-335) - the reference's own bodies on the committed flattop input: a vector of exactly nfft
    samples, a shorter one (fft() zero-pads) and a longer one (fft() truncates, the normalisation keeps the full length)."""
    F = E.load('ofdm_cr_tools.py', ['clc_power_time', 'td_power_estimate', 'fft_plot_dB', 'fft_plot_lin'])
    x = np.load(os.path.join(HERE, 'welch_flattop_2048.npz'))['x']
    Sf, fc, nfft = 1000000, 100.0e6, 2048
    out = {}
    for tag, lo, hi in (('exact', 0, 2048), ('short', 3000, 4500), ('long', 0, 5000)):
        v = x[lo:hi].astype(np.complex128)
        ax, db = F['fft_plot_dB'](v, Sf, fc, nfft)
        ax2, lin = F['fft_plot_lin'](v, Sf, fc, nfft)
        assert ax == ax2
        out[tag + '_range'] = np.array([lo, hi])
        out[tag + '_axis'] = np.array(ax)
        out[tag + '_db'] = np.array(db)
        out[tag + '_lin'] = np.array(lin)
        out[tag + '_power_time'] = np.array(F['clc_power_time'](v))
        out[tag + '_td_power'] = np.array(F['td_power_estimate'](v, Sf))
    save('ref_fft_plot.npz', source=np.array('reference'), input_from=np.array('welch_flattop_2048.npz'), Sf=Sf, fc=fc,
         nfft=nfft, **out)


LOGGER_SESSION = dict(
    # one scripted campaign of the stats / psd / waterfall watchers against the sensing log (f2): values the watchers
    # hand over before the first write, during the first sleep and during the second; the clock is frozen and ticks
    # one stamp per strftime pair; the test duration ends during the second sleep
    fft_len=8, periodicity=5, test_duration=7,
    stamps=[('240131', '235958', '2359'), ('240131', '235959'), ('240201', '000004')],
    settings={'date': '24-01-31', 'time': '23:59:58', 'fft_len': 8, 'sample_rate': 1000000},
    phases=[
        dict(cumulative_psd=[1e-9, 2e-9, 3e-9, 4e-9, 5e-7, 6e-9, 7e-9, 8e-9],
             periodic_psd_peaks=[1e-9, 2e-9, 3e-9, 4e-9, 5e-7, 6e-9, 7e-9, 8e-9],
             cumulative_statistics={100000000.0: 2, 100025000.0: 0}, periodic_statistic={100000000.0: 2},
             cumulative_max_power=[1.5e-6, 2.5e-8], periodic_max_power=[1.5e-6, 2.5e-8], n_measurements_period=2,
             cumulative_waterfall=[[1.234e-5, 6.5e-7, 1e-12], [2.0, 3.0, 4.0]]),
        dict(cumulative_psd=[1e-9, 2e-9, 3e-9, 4e-9, 5e-7, 6e-9, 7e-8, 8e-9],
             periodic_psd_peaks=[5e-10, 2e-10, 3e-10, 4e-10, 5e-10, 6e-10, 7e-8, 8e-10],
             cumulative_statistics={100000000.0: 3, 100025000.0: 1}, periodic_statistic={100000000.0: 1, 100025000.0: 1},
             cumulative_max_power=[1.5e-6, 7.5e-8], n_measurements_period=1,
             cumulative_waterfall=[[9.87e-3, 0.0, 5.55e-5]]),
        dict(cumulative_statistics={100000000.0: 4, 100025000.0: 1}, periodic_statistic={100000000.0: 1},
             periodic_max_power=[2.5e-7, 1.0e-9], cumulative_waterfall=[]),
    ])


def logger_session_apply(lg, phase):
    """The watchers' hand-over of one phase through the logger's setters (ofdm_cr_tools.py:1914-1956)."""
    for key, val in phase.items():
        if key in ('cumulative_psd', 'periodic_psd_peaks', 'cumulative_max_power', 'periodic_max_power'):
            val = np.array(val, np.float32 if 'psd' in key else np.float64)
        elif key == 'cumulative_waterfall':
            val = [np.array(r, np.float32) for r in val]
        getattr(lg, 'set_' + key)(val)


def reference_logger_fixture(E):
    """ref_sensing_log.npz (f2): the reference's OWN ``logger`` (ofdm_cr_tools.py:1850-1956) and ``file_logger``
    (:1958-2107) - constructor, setters, reset and the thread body ``run`` - executed on LOGGER_SESSION with a frozen
    clock, a scratch home directory and a thread base class that does not start; every file they leave behind goes into
    the fixture, name and bytes.  Environment of their day (ref_extract's docstring): ``open(path, 'w')`` gave a file
    that takes the byte strings ``np.save`` writes (Python-2 ``str``); here it is opened in binary mode and encodes text."""
    import io
    import shutil
    import tempfile
    from contextlib import redirect_stdout
    S = LOGGER_SESSION
    home = tempfile.mkdtemp(prefix='ref_logger_')
    stamps = [list(t) for t in S['stamps']]

    class Clock(object):               # time.strftime / time.sleep of the session
        def __init__(self):
            self.k, self.slept = 0, []

        def strftime(self, fmt):
            dat, tim = stamps[self.k][0], stamps[self.k][1]
            if fmt == '%y%m%d':
                return dat
            if fmt == '%H%M':          # the directory name, constructor only
                return stamps[0][2]
            assert fmt == '%H%M%S', fmt
            self.k += 1                # a pair (date, time) is one stamp; the time is asked for last
            return tim

        def sleep(self, seconds):
            self.slept.append(seconds)
            logger_session_apply(lg, S['phases'][len(self.slept)])

    clock = Clock()
    t0 = 1000.0

    class FakeDatetimeModule(object):  # datetime.datetime.now() + datetime.timedelta(seconds=..) on a counter
        class datetime(object):
            @staticmethod
            def now():
                return t0 + S['periodicity'] * len(clock.slept)

        @staticmethod
        def timedelta(seconds):
            return seconds

    opened = []

    class File2(object):               # the file object of its day: takes text and byte strings alike
        def __init__(self, path, mode='r'):
            self.fh = open(path, mode + 'b')
            opened.append(self.fh)

        def write(self, data):
            return self.fh.write(data.encode('latin-1') if isinstance(data, str) else data)

        def __getattr__(self, name):
            return getattr(self.fh, name)

        def __repr__(self):
            return '<open file>'

    class Thread(object):              # _threading.Thread that is never started: run() is called below
        def __init__(self):
            pass

        def setDaemon(self, flag):
            pass

        def start(self):
            pass

    ns = {'time': clock, 'datetime': FakeDatetimeModule, 'open': File2, 'os': os, 'np': np,
          'expanduser': lambda tilde: home, '_threading': type('M', (), {'Thread': Thread})}
    fl_methods = {n: E.load_method('ofdm_cr_tools.py', 'file_logger', n, ns, py2_print=True) for n in ('__init__', 'run')}
    ns['file_logger'] = type('Ref_file_logger', (Thread,), fl_methods)
    names = ['__init__', 'reset_periodic_vars'] + ['set_' + k for k in (
        'cumulative_psd', 'periodic_psd_peaks', 'settings', 'n_measurements_period', 'cumulative_statistics',
        'periodic_statistic', 'cumulative_max_power', 'periodic_max_power', 'cumulative_waterfall')]
    RefLogger = type('Ref_logger', (object,),
                     {n: E.load_method('ofdm_cr_tools.py', 'logger', n, ns, py2_print=True) for n in names})
    said = io.StringIO()
    with redirect_stdout(said):
        lg = RefLogger(S['fft_len'], S['periodicity'], S['test_duration'])
        lg.set_settings(S['settings'])
        logger_session_apply(lg, S['phases'][0])
        lg._file_logger.run()
    assert clock.slept == [S['periodicity']] * 2 and 'test expired' in said.getvalue()
    # The reference never closes a file: it re-binds ``self.psd_file = open(path, 'w')`` and lets CPython's reference
    # count close (and only then flush) the object it replaces, i.e. AFTER the new ``open`` has truncated the file; the
    # last objects are flushed when the interpreter exits.  File2 objects die the same way here; what is still open now
    # is closed in the order it was opened.  (Consequence, not exercised by this session and not part of the format: a
    # rewritten file that got SHORTER keeps the tail of its previous content.)
    for fh in opened:
        fh.close()
    files = {}
    for root, _, fns in os.walk(home):
        for fn in fns:
            with open(os.path.join(root, fn), 'rb') as fh:
                files[os.path.relpath(os.path.join(root, fn), home)] = np.frombuffer(fh.read(), np.uint8)
    shutil.rmtree(home)
    order = sorted(files)
    save('ref_sensing_log.npz', source=np.array('reference'), names=np.array(order), session=np.array(repr(S)),
         said=np.array(said.getvalue().replace(home, '~')),
         **{'file_%d' % i: files[n] for i, n in enumerate(order)})
    print('  ' + '\n  '.join('%s (%d B)' % (n, len(files[n])) for n in order))


def reference_psd_logger_fixture(E):
    """ref_psd_logger.npz (a2, the host side): the reference's own ``_queue_watcher.run`` (psd_logger.py:70-88) on
    stand-in messages - the |X| rows of the committed a2 input, one vector per message, then one message of two
    vectors.  It records what the watcher saved after every message (the peak starts from ``None``: the first vector IS
    the first peak) and how it ends: the two-vector message reaches ``s = s[start:start + itemsize]`` with ``s``
    never assigned (:79-81) and the thread dies with UnboundLocalError - which is why the product block states the
    last-vector rule of the other watchers instead (psd_logger.py header)."""
    from types import SimpleNamespace as NS
    g = np.load(os.path.join(HERE, 'gr_chain_bh_mag_peak_4096.npz'))
    mag = g['expected_mag'].astype(np.float32)
    saved = []
    npd = E.NumpyOfItsDay()
    npd.fromstring = lambda s, dt: np.frombuffer(bytes(s), dt)
    npd.save = lambda path, arr: saved.append((path, np.array(arr)))
    run = E.load_method('psd_logger.py', '_queue_watcher', 'run', {'np': npd})
    msgs = [(4 * mag.shape[1], 1, mag[i].tobytes()) for i in range(mag.shape[0])]
    msgs.append((4 * mag.shape[1], 2, mag[:2].tobytes()))
    me = NS(keep_running=True, mat_file='/tmp/psd_log-stand-in.mat', k=-1)

    def delete_head():
        me.k += 1
        size, n, payload = msgs[me.k]
        return NS(arg1=lambda: size, arg2=lambda: n, to_string=lambda: payload)
    me.rcvd_data = NS(delete_head=delete_head)
    try:
        run(me)
        ended = 'returned'
    except Exception as exc:      # noqa: BLE001 - the fixture records how the reference's thread ends
        ended = type(exc).__name__
    assert len(saved) == mag.shape[0] and all(p == me.mat_file for p, _ in saved)
    save('ref_psd_logger.npz', source=np.array('reference'), input_from=np.array('gr_chain_bh_mag_peak_4096.npz'),
         saved_peaks=np.array([a for _, a in saved]), ended=np.array(ended), died_at_message=np.array(me.k))
    print('  %d saves, then the %d-vector message: %s' % (len(saved), msgs[-1][1], ended))


def reference_consumer_fixture(E):
    """ref_consumers.npz (f1, receiving side): the reference's OWN consumers of the fragment format -
    ``data_processor.run`` of the web server (sdr_webserver/sdr_webserver_ws.py:235-287; strips the 10-byte ZMQ / PMT
    header, forwards the reassembled payload) and ``remote_client_qt.handler`` (python/remote_client_qt.py:100-164; decodes,
    peak hold, two curves) - fed frame by frame from a stand-in socket / as stand-in PDUs: complete vectors, a vector
    that fits one fragment, a lost middle fragment, a lost LAST fragment (two vectors glued), a payload of ragged
    length (the error branch: the web consumer drops what is pending, the Qt one keeps it).  Stored: every frame, and
    per consumer the index of each frame that completed something with what came out."""
    import io
    import struct
    from contextlib import redirect_stdout
    from types import SimpleNamespace as NS
    hdr = lambda fr: R.zmq_pdu_header(len(fr)) + fr      # noqa: E731
    rows = [(np.arange(2048, dtype=np.float32) * 0.02 - 95 + 3 * k + np.sin(np.arange(2048) * (k + 1))).astype('<f4')
            for k in range(5)]
    small = (np.arange(256, dtype=np.float32) - 120).astype('<f4')
    f32 = [R.worker_fragments(r, 1470, 2048, True) for r in rows]            # six fragments per vector
    i8 = [R.worker_fragments(r, 1470, 2048, False) for r in rows]            # two
    one = R.worker_fragments(small, 1470, 256, True)                          # one
    sw = [R.sweeper_fragments(r.tobytes(), 1470) for r in rows[:2]]
    lossy = (f32[0] + f32[1][:2] + f32[1][3:]            # middle fragment lost: a shorter vector
             + f32[2][:-1] + f32[3]                       # last fragment lost: glued to the next vector
             + f32[4][:-1] + [f32[4][-1][:-3]]            # ragged last payload: the error branch
             + one + f32[0])                              # what the next vectors look like after it
    streams = {'f32': sum(f32[:3], []) + one, 'i8': sum(i8, []), 'sweeper': sum(sw, []), 'lossy': lossy}
    npd = E.NumpyOfItsDay()
    npd.fromstring = lambda s, dt: np.frombuffer(s.encode('latin-1') if isinstance(s, str) else bytes(s), dt)
    out = {}

    # ---- the web server's thread body
    run = E.load_method('../sdr_webserver/sdr_webserver_ws.py', 'data_processor', 'run',
                        {'np': npd, 'struct': struct, 'sys': sys, 'getThreadId': lambda: 0}, py2_print=True,
                        fixers=('print', 'except'))
    for tag, dt in (('f32', np.float32), ('sweeper', np.float32), ('lossy', np.float32)):
        frames, got, said = [hdr(fr) for fr in streams[tag]], [], []
        me = NS(logger=NS(info=said.append), device='stand-in', keep_running=True, reasembled_frame='', data_type=dt,
                max_fft_data=np.array([]), strt=True, k=-1)

        def recv():
            me.k += 1
            me.keep_running = me.k + 1 < len(frames)
            return E.Str2(frames[me.k])
        me.zmq_sub = NS(recv=recv)
        me.shared_queue_data = NS(put=lambda v: got.append((me.k, bytes(v))))
        run(me)
        out['web_%s_at' % tag] = np.array([k for k, _ in got])
        for j, (_, b) in enumerate(got):
            out['web_%s_out_%d' % (tag, j)] = np.frombuffer(b, np.uint8)
        out['web_%s_errors' % tag] = np.array(sum('error' in line for line in said))

    # ---- the Qt client's message handler
    pmt = NS(cdr=lambda m: m[1], car=lambda m: m[0], u8vector_elements=lambda v: list(v), to_python=lambda v: v)
    struct2 = NS(unpack=lambda fmt, s: struct.unpack(fmt, s.encode('latin-1') if isinstance(s, str) else s))
    handler = E.load_method('remote_client_qt.py', 'remote_client_qt', 'handler',
                            {'np': npd, 'struct': struct2, 'pmt': pmt, 'QtCore': NS(SIGNAL=lambda s: s)}, py2_print=True)
    for tag, dt in (('f32', np.float32), ('i8', np.int8), ('lossy', np.float32)):
        got, said = [], io.StringIO()
        me = NS(data_type=dt, max_fft_data=np.array([]), strt=True, reasembled_frame='', hold_max=True,
                sample_rate=2.0e6, tune_freq=100.0e6, curve_data=[([], []), ([], [])], k=-1)

        def emit(signal, arg):
            # :127-129 / :154-156 - a one-fragment vector puts the data on curve 0 and the peak on 1, a reassembled
            # one the other way round
            a, b = me.curve_data[0][1], me.curve_data[1][1]
            got.append((me.k, np.array(me.max_fft_data), np.array(a), np.array(b), np.array(me.curve_data[0][0])))
        me.emit = emit
        with redirect_stdout(said):
            for k, fr in enumerate(streams[tag]):
                me.k = k
                handler(me, (None, bytearray(fr)))
        out['qt_%s_at' % tag] = np.array([k for k, *_ in got])
        for j, (_, peak, c0, c1, axis) in enumerate(got):
            out['qt_%s_peak_%d' % (tag, j)] = peak
            out['qt_%s_curve0_%d' % (tag, j)] = c0
            out['qt_%s_curve1_%d' % (tag, j)] = c1
        out['qt_%s_axis_mhz' % tag] = got[-1][4]
        out['qt_%s_errors' % tag] = np.array(said.getvalue().count('error reassembling'))
        out['qt_%s_pending' % tag] = np.array(len(me.reasembled_frame))
    for tag, frames in streams.items():
        out['frames_%s_len' % tag] = np.array([len(fr) for fr in frames])
        out['frames_%s' % tag] = np.frombuffer(b''.join(frames), np.uint8)
    save('ref_consumers.npz', source=np.array('reference'), **out)
    for k in sorted(out):
        if k.endswith('_at') or k.endswith('_errors') or k.endswith('_pending'):
            print('  %s = %s' % (k, out[k]))


def consumer_fixture():
    """fragments_consumer.bin (restated): what the two consumers receive and what they must decode.  Three streams of
    frames - local_worker float32 frames as they leave a ZMQ PUB sink (10-byte PMT header in front of each),
    local_worker int8 frames bare (the Qt client's msg port), sweeper frames with the header - each followed by the
    vectors the consumer decodes from it.  Layout: u32 nstreams; per stream u32 header_len, u32 itemsize (4 = '<f4',
    1 = int8), u32 nframes, frames as u32 length + bytes, u32 nvectors, vectors as u32 nbytes + raw bytes."""
    rows = [(np.arange(2048, dtype=np.float32) * 0.02 - 95 + 3 * k).astype('<f4') for k in range(3)]
    f32 = [R.zmq_pdu_header(len(fr)) + fr for row in rows for fr in R.worker_fragments(row, 1470, 2048, True)]
    i8 = [fr for row in rows for fr in R.worker_fragments(row, 1470, 2048, False)]
    sw = [R.zmq_pdu_header(len(fr)) + fr for row in rows[:2] for fr in R.sweeper_fragments(row.tobytes(), 1470)]
    with open(os.path.join(HERE, 'fragments_consumer.bin'), 'wb') as fh:
        fh.write(np.uint32(3).tobytes())
        for frames, header, dt in ((f32, 10, '<f4'), (i8, 0, np.int8), (sw, 10, '<f4')):
            vecs, _ = R.consumer_handler(frames, dt, header)
            fh.write(np.array([header, np.dtype(dt).itemsize, len(frames)], np.uint32).tobytes())
            for fr in frames:
                fh.write(np.uint32(len(fr)).tobytes())
                fh.write(fr)
            fh.write(np.uint32(len(vecs)).tobytes())
            for v in vecs:
                b = np.ascontiguousarray(v).tobytes()
                fh.write(np.uint32(len(b)).tobytes())
                fh.write(b)
    print('wrote fragments_consumer.bin')


def T_db(v):
    return np.asarray(v, np.float64)


def main():
    import warnings
    warnings.simplefilter('ignore')
    if '--consumer' in sys.argv:
        consumer_fixture()
        return
    if '--reference' in sys.argv:
        return reference_fixtures()

    # a6 - default Hann / 50 % overlap Welch, the BASELINE config-2 call pattern
    x = R.synth_iq(65536, 1002)
    f, p = sg.welch(x.astype(np.complex128), fs=1.0, nperseg=4096, nfft=4096)
    f32, p32 = sg.welch(x, fs=1.0, nperseg=4096, nfft=4096)
    save('welch_hann_4096_50.npz', source=np.array('scipy'), seed=1002, x=x, fs=1.0, nfft=4096,
         expected_psd=p, expected_freqs=f, scipy_c64_psd=p32)

    # a6 with fs != 1 and a length that is not a multiple of the step
    x = R.synth_iq(50000, 7)
    f, p = sg.welch(x.astype(np.complex128), fs=2.0e6, nperseg=1024, nfft=1024)
    save('welch_hann_1024_ragged.npz', source=np.array('scipy'), seed=7, x=x, fs=2.0e6, nfft=1024,
         expected_psd=p)

    # a6/a14 - src_power_welch: flattop, nperseg = nfft
    x = R.synth_iq(32768, 11)
    f, p = sg.welch(x.astype(np.complex128), window='flattop', fs=1.0e6, nperseg=2048, nfft=2048)
    save('welch_flattop_2048.npz', source=np.array('scipy'), seed=11, x=x, fs=1.0e6, nfft=2048,
         expected_psd=p)

    # a4 - sweeper segment: flattop, nperseg = nfft/4 zero-padded, fftshift, trim, dB
    x = R.synth_iq(32768, 2000)
    nfft, excess, fs = 4096, 256, 2.0e6
    f, p = sg.welch(x.astype(np.complex128), window='flattop', fs=fs, nperseg=nfft / 4.0, nfft=nfft)
    psd = np.fft.fftshift(p)[excess:-excess]
    save('welch_flattop_nperseg_quarter.npz', source=np.array('scipy'), seed=2000, x=x, fs=fs,
         nfft=nfft, excess_bins=excess, expected_psd_lin=psd, expected_psd_db=10 * np.log10(psd))

    # a13 - two-channel csd / coherence (hann, 4096, 50 %)
    x = R.synth_iq(65536, 1003)
    rng = np.random.default_rng(1004)
    noise = (rng.standard_normal(65536) + 1j * rng.standard_normal(65536)) / np.sqrt(2.0)
    y = (0.7 * np.roll(x.astype(np.complex128), 5) + 0.5 * noise).astype(np.complex64)
    x64, y64 = x.astype(np.complex128), y.astype(np.complex128)
    _, pxx = sg.welch(x64, fs=1.0, nperseg=4096, nfft=4096)
    _, pyy = sg.welch(y64, fs=1.0, nperseg=4096, nfft=4096)
    _, pxy = sg.csd(x64, y64, fs=1.0, nperseg=4096, nfft=4096)
    _, cxy = sg.coherence(x64, y64, fs=1.0, nperseg=4096, nfft=4096)
    save('coherence_csd_4096.npz', source=np.array('scipy'), x=x, y=y, fs=1.0, nfft=4096,
         expected_pxx=pxx, expected_pyy=pyy, expected_pxy=pxy, expected_cxy=cxy)

    # a1 - v2 / scanner chain: rect window, shifted FFT, |.|^2 / N^2, + 8-row mean (numpy)
    x = R.synth_iq(65536, 1001)
    N = 1024
    X = np.fft.fftshift(np.fft.fft(x.astype(np.complex128).reshape(-1, N), axis=1), axes=1)
    rows = np.abs(X) ** 2 / N ** 2
    # comparator, not expectation: the same chain on a single-precision CPU FFT (scipy.fft keeps complex64 in fp32 -
    # FFTW3f-class arithmetic, which is what fft_vcc runs), |X|^2 and the 1/N^2 product in float32 as the GR blocks do.
    # The GPU tests hold the HIP rows to at most 1.5 x this path's own error against the float64 expectation.
    Xc = sfft.fft(x.reshape(-1, N), axis=1)
    assert Xc.dtype == np.complex64
    Xc = np.fft.fftshift(Xc, axes=1)
    c64_rows = (Xc.real * Xc.real + Xc.imag * Xc.imag) * np.float32(1.0 / (N * N))
    save('gr_chain_rect_1024.npz', source=np.array('numpy'), seed=1001, x=x, nfft=N,
         expected_rows=rows, expected_mean8=rows.reshape(-1, 8, N).mean(axis=1), c64_rows=c64_rows)

    # a2 - psd_logger chain: BH window, natural order, |.|, running peak (numpy + restated window)
    x = R.synth_iq(65536, 5)
    N = 4096
    w = sg.windows.blackmanharris(N, sym=True)
    mag = np.abs(np.fft.fft(x.astype(np.complex128).reshape(-1, N) * w, axis=1))
    Xc = sfft.fft(x.reshape(-1, N) * w.astype(np.float32), axis=1)      # single-precision comparator (see a1)
    assert Xc.dtype == np.complex64
    c64_mag = np.abs(Xc)
    assert c64_mag.dtype == np.float32
    save('gr_chain_bh_mag_peak_4096.npz', source=np.array('numpy'), seed=5, x=x, nfft=N, window=w,
         expected_mag=mag, expected_peak=np.maximum.accumulate(mag, axis=0), c64_mag=c64_mag)

    # a3 - local_worker chain: BH, shifted, |.|^2, IIR(0.8), 10log10 + k (numpy)
    x = R.synth_iq(65536, 6)
    N, Sf, alpha = 2048, 2000000, 0.8
    w = sg.windows.blackmanharris(N, sym=True)
    p = np.abs(np.fft.fftshift(np.fft.fft(x.astype(np.complex128).reshape(-1, N) * w, axis=1), axes=1)) ** 2
    yv = np.zeros(N)
    lin = []
    for r in p:
        yv = alpha * r + (1 - alpha) * yv
        lin.append(yv)
    lin = np.array(lin)
    k = -10 * np.log10(N) - 10 * np.log10(Sf)
    Xc = np.fft.fftshift(sfft.fft(x.reshape(-1, N) * w.astype(np.float32), axis=1), axes=1)      # comparator (see a1)
    pc = Xc.real * Xc.real + Xc.imag * Xc.imag
    yc, c64_lin = np.zeros(N, np.float32), []
    for r in pc:
        yc = np.float32(alpha) * r + np.float32(1 - alpha) * yc
        c64_lin.append(yc)
    c64_lin = np.array(c64_lin)
    assert c64_lin.dtype == np.float32
    save('gr_chain_bh_iir_log_2048.npz', source=np.array('numpy'), seed=6, x=x, nfft=N, sample_rate=Sf,
         average=alpha, window=w, expected_lin=lin, expected_db=10 * np.log10(lin) + k, c64_lin=c64_lin)

    # a7 - src_power: np.convolve('same') moving average + channel sums (restated on numpy calls)
    rng = np.random.default_rng(21)
    cases = []
    for (Sf, N, cs, sbw) in [(1000000, 1024, 25e3, 12.5e3), (2000000, 4096, 200e3, 150e3),
                             (1000000, 16384, 15625.0, 10e3), (250000, 512, 12.5e3, 3e3)]:
        psd = rng.random(N) ** 4 + 1e-3
        Fr = float(Sf) / N
        bb = R.frange(-Sf // 2, Sf // 2, cs)
        sb = sbw / Fr
        ma = np.abs(np.convolve(psd, np.ones(int(sb)) / float(sb), 'same'))
        cases.append(dict(Sf=Sf, N=N, cs=cs, sbw=sbw, psd=psd, ma=ma,
                          plc=np.array(R.src_power(psd, N, Fr, Sf, bb, sb))))
    save('src_power_cases.npz', source=np.array('numpy+restated'), n=len(cases),
         **{'%s_%d' % (k, i): np.asarray(v) for i, c in enumerate(cases) for k, v in c.items()})

    # a8/a9/a10 - scanner state over 16 consecutive PSD rows (restated)
    x = R.synth_iq(16 * 1024, 33)
    rows = R.chain_sensor_v2(x, 1024)
    st = R.ScannerState(1024, 1000000, 25e3, 12.5e3, tune_freq=100000000, trunc_band=800000,
                        thr_leveler=4, alpha_avg=0.5)
    plcs, occ, noise = [], [], []
    for r in rows:
        plc, o = st.scan(r.astype(np.float32))
        plcs.append(st.plc.copy())
        occ.append(np.array([1.0 if a in o else 0.0 for a in st.ax_ch]))
        noise.append(st.noise_estimate)
    subj = [st.ax_ch[3], st.ax_ch[10], st.ax_ch[11], st.ax_ch[20], st.ax_ch[25], st.ax_ch[28]]
    pwr, top4 = R.publish_top4(st.plc, st.ax_ch, subj)
    save('scanner_state_seq.npz', source=np.array('restated'), x=x, rows=rows, ax_ch=np.array(st.ax_ch),
         plc_seq=np.array(plcs), occupied_seq=np.array(occ), noise_seq=np.array(noise),
         cumulative_max=st.cumulative_max_power, subject_channels=np.array(subj), subject_pwr=pwr,
         top4=np.array(top4), peak=R.peak_hold(rows)[-1])

    # a11 - coherence detector decision stage (restated)
    rng = np.random.default_rng(44)
    N, Sf, tune = 4096, 2000000, 433000000
    d0 = rng.random(N).astype(np.float32) * 8
    d1 = rng.random(N).astype(np.float32) * 0.25
    d2 = rng.random(N).astype(np.float32) * 0.25
    ax = R.coherence_axis(N, Sf, tune)
    subj = [tune - 600e3, tune - 100e3, tune + 3.3e3, tune + 250e3, tune + 900e3]
    idx = [R.find_nearest_index(ax, c) for c in subj]
    for ch in (idx[1], idx[3]):          # force both decision branches
        d0[ch - 1:ch + 1] = 6.5
        d1[ch - 1:ch + 1] = 0.05
        d2[ch - 1:ch + 1] = 0.05
    d1[idx[3]] = 0.3                     # coherent but MTM-L too high -> rejected
    coh, outcome, valve = R.coherence_scanner(d0, d1, d2, idx, 10, 0.2)
    save('coherence_scanner.npz', source=np.array('restated'), d0=d0, d1=d1, d2=d2, N=N, sample_rate=Sf,
         tune_freq=tune, subject_channels=np.array(subj), idx=np.array(idx), coherence=np.array(coh),
         outcome=np.array(outcome), valve=np.array(valve))

    # a12 - xcorr / fac (numpy)
    a = R.synth_iq(3000, 51)
    b = np.roll(a, 37) + R.synth_iq(3000, 52) * 0.3
    L = 4096
    e, f_ = np.fft.fft(a, L), np.fft.fft(b, L)
    h = np.fft.fftshift(np.fft.ifft(f_ * np.conj(e), L))
    bb = np.fft.fftshift(np.fft.fft(np.abs(np.fft.fft(a, L)), L))
    save('xcorr_fac.npz', source=np.array('numpy'), a=a, b=b.astype(np.complex64), L=L,
         expected_xcorr=np.abs(h[L // 2:]), expected_fac=np.abs(bb[L // 2:]))

    # f1 - fragment wire format (restated)
    db = (np.arange(4096, dtype=np.float32) * 0.01 - 90).astype('<f4')
    fw = R.worker_fragments(db, 1472 - 2, 4096, True)
    fw8 = R.worker_fragments(db, 1472 - 2, 4096, False)
    fs_ = R.sweeper_fragments(db.tobytes(), 1472 - 2)
    with open(os.path.join(HERE, 'fragments.bin'), 'wb') as fh:
        for group in (fw, fw8, fs_):
            fh.write(np.uint32(len(group)).tobytes())
            for fr in group:
                fh.write(np.uint32(len(fr)).tobytes())
                fh.write(fr)
    print('wrote fragments.bin')
    consumer_fixture()
    sys.path.insert(0, HERE)
    import ref_extract
    if ref_extract.available():
        reference_fixtures()


if __name__ == '__main__':
    main()
