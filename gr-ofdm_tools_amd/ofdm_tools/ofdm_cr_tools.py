"""Spectrum-scanner helpers of ``ofdm_tools.ofdm_cr_tools`` with the reference's
names, argument order and return values (python/ofdm_cr_tools.py:136-250,
:321-345, :471-537), computing on the MI355X through libofdmtools_hip.so.

Host code here is limited to what the reference also does in Python around the
arithmetic: ``frange`` lists, slice bounds with ``int()`` truncation, threshold
bookkeeping.  The FFTs, the Welch averaging, the moving average and the channel
sums run in HIP kernels; nothing here falls back to NumPy/SciPy for them.
"""
import math

import numpy as np

from . import _hip
from . import windows


def _is_int(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, bool)


def _py2div(a, b):
    """The reference is Python 2: ``/`` between two ints floors."""
    if _is_int(a) and _is_int(b):
        return a // b
    return a / b


def _welch_plan(ctx, nfft, window_name, Sf, npts=None, use='exec'):
    """The reference's `sg.welch(x, Sf, window, nperseg=nfft, nfft=nfft)` + fftshift (ofdm_cr_tools.py:214,322,342).
    SciPy shortens nperseg to the input length when the vector is shorter than nfft ("nperseg = N is greater than input
    length", one zero-padded segment) - which fast_spectrum_scan(n_fft=0) always hits, its nFFT being the next power of
    two above len(vct_sample) (ofdm_cr_tools.py:474-475)."""
    nperseg = nfft if npts is None else min(int(nfft), int(npts))
    # `use`: the ticket callers (SpectrumScan: exec_async / poll from work()) keep plans of their own - a plan's output ring
    # holds the last four launches, and blocking helper calls on a shared plan could push an uncollected ticket out of it
    return ctx.cached_plan(('welch', use, nfft, nperseg, window_name, float(Sf)),
                 lambda: ctx.welch_plan(nfft, nperseg=nperseg, window=windows.get_window(window_name, nperseg), fs=float(Sf),
                                        fftshift=True))


def frange(x, y, jump):
    """ofdm_cr_tools.py:136-141."""
    out = []
    while x < y:
        out.append(x)
        x += jump
    return out


def _slice_bounds(n, Fr, Sf, bb_freqs, srch_bins):
    """Start/stop of every channel slice exactly as Python evaluates
    ``psd[0:int(b+sb/2)]`` / ``psd[int(b-sb/2):int(b+sb/2)]`` (ofdm_cr_tools.py:239-248)."""
    half = _py2div(Sf, 2)
    lo, hi = [], []
    for i, f in enumerate(bb_freqs):
        bin_n = (f + half) / Fr
        sl = slice(0, int(bin_n + srch_bins / 2)) if i == 0 else \
            slice(int(bin_n - srch_bins / 2), int(bin_n + srch_bins / 2))
        a, b, _ = sl.indices(n)
        lo.append(a)
        hi.append(max(a, b))
    return lo, hi


def movingaverage(interval, window_size, ctx=None):
    """ofdm_cr_tools.py:168-170 on the device."""
    ctx = ctx or _hip.default_context()
    _, ma = ctx.channel_power(interval, float(window_size), [0], [0], want_movavg=True)
    return ma


def src_power(psd, nFFT, Fr, Sf, bb_freqs, srch_bins, ctx=None):
    """ofdm_cr_tools.py:232-249: moving average, then per-channel sums."""
    ctx = ctx or _hip.default_context()
    lo, hi = _slice_bounds(len(psd), Fr, Sf, bb_freqs, srch_bins)
    return [float(v) for v in ctx.channel_power(psd, float(srch_bins), lo, hi)]


def _plain_channel_sums(psd, Fr, Sf, bb_freqs, srch_bins, ctx):
    # channel sums without the moving average (src_power_welch / src_power_fft): a 1-tap average
    lo, hi = _slice_bounds(len(psd), Fr, Sf, bb_freqs, srch_bins)
    return [float(v) for v in ctx.channel_power(psd, 1.0, lo, hi)]


def _enqueue_welch(vector, nFFT, Sf, ctx):
    """src_power_welch's PSD (flattop, nperseg = nfft, ofdm_cr_tools.py:213-216) as a ticket: -> (plan, ticket, post)."""
    plan = _welch_plan(ctx, nFFT, 'flattop', Sf, len(vector), use='async')
    return plan, plan.exec_async(vector), None


def _enqueue_fft(vector, nFFT, Sf, ctx):
    """src_power_fft's single flat-top periodogram |FFT(x w, nFFT)|^2 / nFFT (ofdm_cr_tools.py:173-178) as a ticket."""
    vector = np.asarray(vector)
    total = len(vector)                   # the reference windows ALL len(vector) samples, then np.fft.fft(., nFFT) keeps the first nFFT
    vector = vector[:nFFT]
    npts = len(vector)
    plan = ctx.cached_plan(('fft', nFFT, total, npts),
                 lambda: ctx.welch_plan(nFFT, nperseg=npts, noverlap=0, window=windows.flattop(total)[:nFFT],
                                        detrend=_hip.DETREND_NONE, scaling=_hip.SCALE_RAW, fftshift=True))
    return plan, plan.exec_async(vector), (lambda psd: psd / np.float32(nFFT))


def src_power_welch(vector, npts, nFFT, Fr, Sf, bb_freqs, srch_bins, ctx=None):
    """ofdm_cr_tools.py:213-230."""
    ctx = ctx or _hip.default_context()
    plan, ticket, _ = _enqueue_welch(vector, nFFT, Sf, ctx)
    psd = plan.wait(ticket)
    axis = np.fft.fftshift(np.fft.fftfreq(nFFT, 1.0 / Sf))
    return psd, axis, _plain_channel_sums(psd, Fr, Sf, bb_freqs, srch_bins, ctx)


def src_power_fft(vector, npts, nFFT, Fr, Sf, bb_freqs, srch_bins, ctx=None):
    """ofdm_cr_tools.py:173-192: one flat-top periodogram |FFT(x w, nFFT)|^2 / nFFT, shifted."""
    ctx = ctx or _hip.default_context()
    plan, ticket, post = _enqueue_fft(vector, nFFT, Sf, ctx)
    psd = post(plan.wait(ticket))
    axis = _py2div(Sf, 2) * np.linspace(-1, 1, nFFT)
    return psd, axis, _plain_channel_sums(psd, Fr, Sf, bb_freqs, srch_bins, ctx)


def clc_power_freq(vector, nFFT, Sf, ctx=None):
    """ofdm_cr_tools.py:149-153."""
    n = len(vector)                       # normalisation uses the full length even when fft() truncates
    return float((_raw_periodogram(vector, nFFT, ctx or _hip.default_context(), False) / n / Sf).sum())


def _raw_periodogram(vector, nFFT, ctx, fftshift):
    """|FFT(vector, nFFT)|^2 on the device - np.fft.fft(v, n) truncates a longer vector and zero-pads a shorter one."""
    vector = np.asarray(vector)[:nFFT]
    plan = ctx.welch_plan(nFFT, nperseg=len(vector), noverlap=0, window=None, detrend=_hip.DETREND_NONE,
                          scaling=_hip.SCALE_RAW, fftshift=fftshift)
    psd = plan.exec(vector).astype(np.float64)
    plan.close()
    return psd


def clc_power_time(vector, ctx=None):
    """ofdm_cr_tools.py:144-146: mean |x|^2 (one reduction on the device)."""
    ctx = ctx or _hip.default_context()
    v = np.ascontiguousarray(vector, np.complex64)
    d = ctx.alloc(v.nbytes)
    try:
        ctx.h2d(d, v)
        mean, var = ctx.iq_power(d, len(v))
    finally:
        ctx.free(d)
    return float(var + abs(mean) ** 2)


def td_power_estimate(vector, Sf, ctx=None):
    """ofdm_cr_tools.py:337-339: sum |x|^2 / Sf."""
    return clc_power_time(vector, ctx) * len(vector) / Sf


def fft_plot_dB(data, Sf, fc, nfft, ctx=None):
    """ofdm_cr_tools.py:312-319: one rectangular periodogram / (npts Sf), shifted, in dB over the shifted axis."""
    psd = _raw_periodogram(data, nfft, ctx or _hip.default_context(), True) / (len(data) * Sf)
    fft_axis = _py2div(Sf, 2) * np.linspace(-1, 1, nfft)
    return [item + fc for item in fft_axis], [10 * math.log10(item + 1e-20) for item in psd]


def fft_plot_lin(data, Sf, fc, nfft, ctx=None):
    """ofdm_cr_tools.py:328-335: the same periodogram, linear."""
    psd = _raw_periodogram(data, nfft, ctx or _hip.default_context(), True) / len(data) / Sf
    fft_axis = _py2div(Sf, 2) * np.linspace(-1, 1, nfft)
    return [item + fc for item in fft_axis], psd


def xcorr(a, b, length, ctx=None):
    """ofdm_cr_tools.py:155-161."""
    return (ctx or _hip.default_context()).xcorr(a, b, length)


def fac(data, length, ctx=None):
    """ofdm_cr_tools.py:163-166."""
    return (ctx or _hip.default_context()).fac(data, length)


def welch_plot_dB(data, Sf, fc, nfft, ctx=None):
    """ofdm_cr_tools.py:321-326 (default Hann window, 50 % overlap)."""
    ctx = ctx or _hip.default_context()
    psd = _welch_plan(ctx, nfft, 'hann', Sf, len(data)).exec(data)
    axis = np.fft.fftshift(np.fft.fftfreq(nfft, 1.0 / Sf))
    return [item + fc for item in axis], [10 * math.log10(item + 1e-20) for item in psd]


def welch_power_estimate(vector, nFFT, Sf, ctx=None):
    """ofdm_cr_tools.py:341-345."""
    ctx = ctx or _hip.default_context()
    return float(np.sum(_welch_plan(ctx, nFFT, 'hann', Sf, len(vector)).exec(vector), dtype=np.float64))


class SpectrumScan(object):
    """The legacy sensor's scan (reference: ofdm_cr_tools.py:471-537; its matplotlib branch is not carried over), split
    where the GPU works: the constructor enqueues the PSD of the chosen method ('welch': flat-top Welch, 'fft': one
    flat-top periodogram) through ``oth_welch_exec_async`` and returns at once - the sample buffer may be reused;
    ``poll(noise_estimate)`` returns None while the launch is running, ``wait(noise_estimate)`` blocks.  Both finish with
    the channel sums on the device, the noise estimate ``ne <- (1 - a) ne + a min(p)``, the threshold ``ne * thr_leveler``
    and the channel frequencies whose power exceeds it: -> (threshold, channel powers, noise estimate, occupied [Hz])."""
    _ENQUEUE = {'welch': _enqueue_welch, 'fft': _enqueue_fft}

    def __init__(self, vct_sample, fc, channel_rate, srch_bw, n_fft, samp_rate, method, thr_leveler, alpha_avg, ctx=None):
        try:
            enqueue = self._ENQUEUE[method]
        except KeyError:
            raise ValueError("method must be 'welch' or 'fft'")
        self.ctx = ctx or _hip.default_context()
        self.nfft = n_fft or int(2 ** math.ceil(math.log(len(vct_sample), 2)))
        self.samp_rate, self.thr_leveler, self.alpha_avg = samp_rate, thr_leveler, alpha_avg
        self.resolution = float(samp_rate) / float(self.nfft)
        half = _py2div(samp_rate, 2)
        self.bb_freqs = frange(_py2div(-samp_rate, 2), half, channel_rate)
        self.srch_bins = srch_bw / self.resolution
        self.channel_hz = frange(fc - half, fc + half, channel_rate)
        self._plan, self._ticket, self._post = enqueue(vct_sample, self.nfft, samp_rate, self.ctx)
        self._result = None

    def _finish(self, psd, noise_estimate):
        if self._post is not None:
            psd = self._post(psd)
        power = _plain_channel_sums(psd, self.resolution, self.samp_rate, self.bb_freqs, self.srch_bins, self.ctx)
        noise_estimate = (1 - self.alpha_avg) * noise_estimate + self.alpha_avg * np.amin(power)
        threshold = noise_estimate * self.thr_leveler
        occupied = [self.channel_hz[i] for i in np.flatnonzero(np.asarray(power) > threshold)]
        self._result = (threshold, power, noise_estimate, occupied)
        return self._result

    def poll(self, noise_estimate):
        if self._result is None:
            psd = self._plan.poll(self._ticket)
            if psd is None:
                return None
            self._finish(psd, noise_estimate)
        return self._result

    def wait(self, noise_estimate):
        if self._result is None:
            self._finish(self._plan.wait(self._ticket), noise_estimate)
        return self._result


def fast_spectrum_scan(vct_sample, fc, channel_rate, srch_bw, n_fft, samp_rate, method, thr_leveler,
                       noise_estimate, alpha_avg, show_plot=False, ctx=None):
    """ofdm_cr_tools.py:471-537 with the reference's arguments and return value: SpectrumScan, waited for."""
    return SpectrumScan(vct_sample, fc, channel_rate, srch_bw, n_fft, samp_rate, method, thr_leveler, alpha_avg,
                        ctx).wait(noise_estimate)


# the reference keeps its file logger next to the numeric helpers (ofdm_cr_tools.py:1850-2107): same import path here
from .sensing_log import logger  # noqa: E402,F401
