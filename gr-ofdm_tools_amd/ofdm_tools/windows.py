"""Window tables handed to the HIP plans (host side, float64 -> float32).

``get_window`` mirrors the periodic windows ``scipy.signal.welch`` builds from a
string (``window='hann'`` default at ofdm_cr_tools.py:322,342, ``'flattop'`` at
ofdm_cr_tools.py:214 and spectrum_sweeper.py:263); ``blackmanharris`` mirrors
``gnuradio.filter.window.blackmanharris`` (symmetric, psd_logger.py:47,
local_worker.py:62) and ``flattop`` the symmetric ``sg.flattop(npts)`` of
ofdm_cr_tools.py:175.
"""
import numpy as np

_FLATTOP = (0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368)
_BH92 = (0.35875, 0.48829, 0.14128, 0.01168)


def _cosine_sum(coeffs, n, denom):
    f = 2.0 * np.pi * np.arange(n) / float(denom)
    w = np.zeros(n)
    for i, a in enumerate(coeffs):
        w += ((-1) ** i) * a * np.cos(i * f)
    return w


def get_window(name, nperseg):
    """Periodic (DFT-even) window, as scipy.signal.get_window(name, nperseg)."""
    if name in ('hann', 'hanning'):
        return _cosine_sum((0.5, 0.5), nperseg, nperseg)
    if name == 'flattop':
        return _cosine_sum(_FLATTOP, nperseg, nperseg)
    if name == 'blackmanharris':
        return _cosine_sum(_BH92, nperseg, nperseg)
    if name in ('boxcar', 'rect', 'rectangular'):
        return np.ones(nperseg)
    raise ValueError('unknown window %r' % (name,))


def blackmanharris(ntaps):
    """gnuradio.filter.window.blackmanharris(ntaps): symmetric 4-term, 92 dB."""
    return _cosine_sum(_BH92, ntaps, max(ntaps - 1, 1))


def flattop(npts):
    """Symmetric flat-top, ``sg.flattop(npts)``."""
    if npts == 1:
        return np.ones(1)
    return _cosine_sum(_FLATTOP, npts, npts - 1)
