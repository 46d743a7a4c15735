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

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy
import scipy.signal as sg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import ref_cpu as R  # noqa: E402

META = dict(scipy=scipy.__version__, numpy=np.__version__)


def save(name, **kw):
    kw['versions'] = np.array(repr(META))
    np.savez_compressed(os.path.join(HERE, name), **kw)
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in kw.items()})


def main():
    import warnings
    warnings.simplefilter('ignore')

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
    save('gr_chain_rect_1024.npz', source=np.array('numpy'), seed=1001, x=x, nfft=N,
         expected_rows=rows, expected_mean8=rows.reshape(-1, 8, N).mean(axis=1))

    # a2 - psd_logger chain: BH window, natural order, |.|, running peak (numpy + restated window)
    x = R.synth_iq(65536, 5)
    N = 4096
    w = sg.windows.blackmanharris(N, sym=True)
    mag = np.abs(np.fft.fft(x.astype(np.complex128).reshape(-1, N) * w, axis=1))
    save('gr_chain_bh_mag_peak_4096.npz', source=np.array('numpy'), seed=5, x=x, nfft=N, window=w,
         expected_mag=mag, expected_peak=np.maximum.accumulate(mag, axis=0))

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
    save('gr_chain_bh_iir_log_2048.npz', source=np.array('numpy'), seed=6, x=x, nfft=N, sample_rate=Sf,
         average=alpha, window=w, expected_lin=lin, expected_db=10 * np.log10(lin) + k)

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


if __name__ == '__main__':
    main()
