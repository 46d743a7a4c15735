#!/usr/bin/env python3
"""Single-row accuracy of the fused periodogram chain on the committed gr_chain_* fixtures, next to the
single-precision CPU FFT comparator rows stored in them (scipy.fft on complex64: FFTW3f-class arithmetic).
Prints, per fixture: the plain max relative error, the round-1 criterion (|d| / max(ref, 1e-3 median)), the worst
amplitude error in ulps of the row's peak, and the per-row ratio HIP error / CPU-fp32 error - the numbers behind
tests/test_hip_parity.py::check_single_rows.   usage: acc_rows.py   (OFDM_TOOLS_HIP_LIB selects another build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
ctx = _hip.Context(0)


def fft_radix2_c64(x):
    """Textbook iterative radix-2 DIT in complex64: a second, independent CPU fp32 FFT (rows along axis 1)."""
    x = x.astype(np.complex64)
    n = x.shape[1]
    lg = n.bit_length() - 1
    idx, rev = np.arange(n), np.zeros(n, int)
    for b in range(lg):
        rev |= ((idx >> b) & 1) << (lg - 1 - b)
    a, L = x[:, rev].copy(), 2
    while L <= n:
        w = np.exp(-2j * np.pi * np.arange(L // 2) / L).astype(np.complex64)
        a = a.reshape(x.shape[0], n // L, L)
        e, o = a[:, :, :L // 2], a[:, :, L // 2:] * w
        a = np.concatenate([e + o, e - o], axis=2).reshape(x.shape[0], n)
        L *= 2
    return a


def report(name, got, cpu, ref, power):
    got, cpu, ref = (np.asarray(v, np.float64) for v in (got, cpu, ref))
    for tag, v in (('HIP', got), ('c64', cpu)):
        amp, amp_ref = (np.sqrt(v), np.sqrt(ref)) if power else (v, ref)
        e = np.abs(amp - amp_ref) / amp_ref.max(axis=1, keepdims=True) * 2.0 ** 23
        rel = np.abs(v - ref) / ref
        old = np.abs(v - ref) / np.maximum(ref, 1e-3 * np.median(ref))
        print('%-28s %s  plain rel %.3e  old a1 criterion %.3e  amp err ulp-of-peak: worst %.3f median row %.3f  mean rel %.2e'
              % (name, tag, rel.max(), old.max(), e.max(), np.median(e.max(axis=1)), rel.mean()))
        if tag == 'HIP':
            eh = e.max(axis=1)
        else:
            ec = e.max(axis=1)
    print('%-28s per-row HIP/c64 amplitude-error ratio: max %.2f median %.2f; fixture worst HIP / worst c64 %.2f'
          % (name, np.max(eh / ec), np.median(eh / ec), eh.max() / ec.max()))


g = np.load(os.path.join(G, 'gr_chain_rect_1024.npz'))
ch = ctx.chain(1024, None, True, _hip.EPI_MAG2_OVER_N2, 1)
rows, n = ch.push(g['x'])
ch.close()
report('a1 rect 1024 |X|^2/N^2', rows, g['c64_rows'], g['expected_rows'], True)
# how far two CPU fp32 FFTs are apart row by row on the same fixture (why the per-row ratio is not a test criterion)
X2 = np.fft.fftshift(fft_radix2_c64(g['x'].reshape(-1, 1024)), axes=1)
r2 = (X2.real * X2.real + X2.imag * X2.imag) * np.float32(1.0 / (1024 * 1024))
print('--- CPU radix-2 complex64 in the role of "HIP" against the stored pocketfft rows:')
report('a1 radix-2 c64 (CPU)', r2, g['c64_rows'], g['expected_rows'], True)

g = np.load(os.path.join(G, 'gr_chain_bh_mag_peak_4096.npz'))
ch = ctx.chain(4096, g['window'], False, _hip.EPI_MAG, 1)
rows, n = ch.push(g['x'])
ch.close()
report('a2 BH 4096 |X|', rows, g['c64_mag'], g['expected_mag'], False)

g = np.load(os.path.join(G, 'gr_chain_bh_iir_log_2048.npz'))
ch = ctx.chain(2048, g['window'], True, _hip.EPI_MAG2, 1)
ch.set_iir_log(float(g['average']), 0.0)
rows, n = ch.push(g['x'])
ch.close()
report('a3 BH 2048 IIR rows (lin)', 10.0 ** (rows.astype(np.float64) / 10.0), g['c64_lin'], g['expected_lin'], True)
for nfft in (256, 512, 2048):      # sizes without a committed fixture: same construction as a1
    from oracle import ref_cpu as R
    import scipy.fft as sfft
    x = R.synth_iq(nfft * 32, 1001)
    X = np.fft.fftshift(np.fft.fft(x.astype(np.complex128).reshape(-1, nfft), axis=1), axes=1)
    ref = np.abs(X) ** 2 / nfft ** 2
    Xc = np.fft.fftshift(sfft.fft(x.reshape(-1, nfft), axis=1), axes=1)
    cpu = (Xc.real * Xc.real + Xc.imag * Xc.imag) * np.float32(1.0 / (nfft * nfft))
    ch = ctx.chain(nfft, None, True, _hip.EPI_MAG2_OVER_N2, 1)
    rows, n = ch.push(x)
    ch.close()
    report('a1 form at %d' % nfft, rows, cpu, ref, True)
