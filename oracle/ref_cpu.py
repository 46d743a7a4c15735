"""CPU oracle for the gr-ofdm_tools spectrum-sensing hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``gr-ofdm_tools_amd/ofdm_tools``) never does and
fails loudly when the HIP library is missing.

What it is
----------
A Python-3 / NumPy restatement of the reference's algorithm for the path in
SURVEY.md section 8, function by function, each citing the reference file:line
it follows (paths relative to the upstream tree, e.g. ``python/ofdm_cr_tools.py``).
The reference itself is Python 2 on GNU Radio 3.7 and can neither be imported
nor run here (ordinary SyntaxError / ModuleNotFoundError), so:

* SciPy / NumPy boundary (``scipy.signal.welch`` call sites
  ofdm_cr_tools.py:214,322,342 and spectrum_sweeper.py:263, ``numpy.fft``
  call sites ofdm_cr_tools.py:151,157-160,164-165,177): **pinned**.  The same
  libraries are installed here (scipy 1.15.3 / numpy 2.2.6); the golden
  fixtures under ``tests/golden`` are produced by calling them with the
  reference's exact argument patterns (``tests/golden/make_golden.py``), and
  ``welch_np`` / ``csd_np`` below (an independent float64 restatement of the
  published Welch algorithm, scipy/signal/_spectral_py.py::_spectral_helper)
  is checked against them.
* GNU Radio boundary (``fft.fft_vcc``, ``blocks.keep_one_in_n``,
  ``blocks.complex_to_mag[_squared]``, ``single_pole_iir_filter_ff``,
  ``nlog10_ff``, ``window.blackmanharris``; call sites
  spectrum_sensor_v2.py:85-93, psd_logger.py:43-53, local_worker.py:58-69):
  **parity unpinned** - GNU Radio (>= 3.7.2, CMakeLists.txt:113) is not in
  the reference tree nor installed, and the reference holds no test vectors.
  The ``gr_*`` functions restate the documented GR 3.7 block semantics.

Python-2 semantics that the reference relies on (integer ``/``) are
reproduced deliberately through ``_py2div``.
"""
from __future__ import annotations

import math
import struct

import numpy as np

# --------------------------------------------------------------------------
# Python-2 arithmetic helpers
# --------------------------------------------------------------------------


def _is_int(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, bool)


def _py2div(a, b):
    """Python 2 ``a / b``: floor division when both operands are ints."""
    if _is_int(a) and _is_int(b):
        return a // b
    return a / b


# --------------------------------------------------------------------------
# Helpers from python/ofdm_cr_tools.py
# --------------------------------------------------------------------------


def frange(x, y, jump):
    """ofdm_cr_tools.py:136-141 - strict ``<``, float accumulation."""
    out = []
    while x < y:
        out.append(x)
        x += jump
    return out


def frange_le(x, y, jump):
    """spectrum_sweeper.py:37-42 - inclusive ``<=`` variant."""
    out = []
    while x <= y:
        out.append(x)
        x += jump
    return out


def clc_power_freq(vector, nFFT, Sf):
    """ofdm_cr_tools.py:149-153."""
    n = len(vector)
    psd_fft = np.fft.fftshift((np.absolute(np.fft.fft(vector, nFFT)) ** 2) / n) / Sf
    return float(sum(psd_fft))


def clc_power_time(vector):
    """ofdm_cr_tools.py:144-146."""
    return float(np.sum(np.absolute(np.asarray(vector, np.complex128)) ** 2)) / len(vector)


def td_power_estimate(vector, Sf):
    """ofdm_cr_tools.py:337-339."""
    return float(np.sum(np.absolute(np.asarray(vector, np.complex128)) ** 2)) / Sf


def fft_plot_lin(data, Sf, fc, nfft):
    """ofdm_cr_tools.py:328-335: fftshift(|fft(data, nfft)|^2 / npts) / Sf over Sf/2 * linspace(-1, 1, nfft) + fc."""
    npts = len(data)
    psd = np.fft.fftshift(np.absolute(np.fft.fft(np.asarray(data, np.complex128), nfft)) ** 2 / npts) / Sf
    axis = _py2div(Sf, 2) * np.linspace(-1, 1, nfft)
    return [a + fc for a in axis], psd


def fft_plot_dB(data, Sf, fc, nfft):
    """ofdm_cr_tools.py:312-319: the same periodogram / (npts Sf) in dB (+1e-20 inside the log)."""
    axis, psd = fft_plot_lin(data, Sf, fc, nfft)
    return axis, [10 * math.log10(v + 1e-20) for v in psd]


def xcorr(a, b, length):
    """ofdm_cr_tools.py:155-161 (``len(h)/2`` is Python-2 integer division)."""
    e = np.fft.fft(a, length)
    f = np.fft.fft(b, length)
    g = f * np.conj(e)
    h = np.fft.fftshift(np.fft.ifft(g, length))
    return np.abs(h[len(h) // 2:])


def fac(data, length):
    """ofdm_cr_tools.py:163-166."""
    b = np.abs(np.fft.fft(data, length))
    b = np.fft.fftshift(np.fft.fft(b, length))
    return np.abs(b[len(b) // 2:])


def movingaverage(interval, window_size):
    """ofdm_cr_tools.py:168-170."""
    window = np.ones(int(window_size)) / float(window_size)
    return np.abs(np.convolve(interval, window, 'same'))


def _channel_sums(psd, Fr, Sf, bb_freqs, srch_bins):
    """The per-channel slice sums shared by ofdm_cr_tools.py:183-191,
    :221-229 and :239-248.  ``Sf/2`` keeps Python-2 integer semantics."""
    half = _py2div(Sf, 2)
    out = []
    f = bb_freqs[0]
    bin_n = (f + half) / Fr
    out.append(float(sum(psd[0:int(bin_n + srch_bins / 2)])))
    for f in bb_freqs[1:]:
        bin_n = (f + half) / Fr
        out.append(float(sum(psd[int(bin_n - srch_bins / 2):int(bin_n + srch_bins / 2)])))
    return out


def src_power(psd, nFFT, Fr, Sf, bb_freqs, srch_bins):
    """ofdm_cr_tools.py:232-249 - moving average then channel sums."""
    psd = movingaverage(psd, 1 * srch_bins)
    return _channel_sums(psd, Fr, Sf, bb_freqs, srch_bins)


def flattop(npts):
    """``sg.flattop(npts)`` of the reference's SciPy era (ofdm_cr_tools.py:175):
    symmetric 5-term flat-top window (scipy.signal.windows.flattop, sym=True)."""
    a = [0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368]
    if npts == 1:
        return np.ones(1)
    n = np.arange(npts)
    fac_ = 2.0 * np.pi * n / (npts - 1)
    return (a[0] - a[1] * np.cos(fac_) + a[2] * np.cos(2 * fac_)
            - a[3] * np.cos(3 * fac_) + a[4] * np.cos(4 * fac_))


def src_power_fft(vector, npts, nFFT, Fr, Sf, bb_freqs, srch_bins):
    """ofdm_cr_tools.py:173-192 - single flat-top periodogram."""
    win = flattop(npts)
    vector = np.asarray(vector) * win
    psd_fft = np.fft.fftshift((np.absolute(np.fft.fft(vector, nFFT)) ** 2) / nFFT)
    fft_axis = _py2div(Sf, 2) * np.linspace(-1, 1, nFFT)
    return psd_fft, fft_axis, _channel_sums(psd_fft, Fr, Sf, bb_freqs, srch_bins)


# --------------------------------------------------------------------------
# Welch / CSD / coherence: float64 restatement of the published algorithm
# (scipy 1.15.3, scipy/signal/_spectral_py.py::_spectral_helper) that the
# reference reaches through sg.welch.
# --------------------------------------------------------------------------


def get_window(name, nperseg):
    """Periodic (``fftbins=True``) windows as ``scipy.signal.get_window``
    builds them for welch()'s string ``window=`` argument."""
    n = np.arange(nperseg)
    if name in ('hann', 'hanning'):
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / nperseg)
    if name == 'flattop':
        a = [0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368]
        f = 2.0 * np.pi * n / nperseg
        return (a[0] - a[1] * np.cos(f) + a[2] * np.cos(2 * f)
                - a[3] * np.cos(3 * f) + a[4] * np.cos(4 * f))
    if name == 'blackmanharris':
        a = [0.35875, 0.48829, 0.14128, 0.01168]
        f = 2.0 * np.pi * n / nperseg
        return a[0] - a[1] * np.cos(f) + a[2] * np.cos(2 * f) - a[3] * np.cos(3 * f)
    if name in ('boxcar', 'rect', 'rectangular'):
        return np.ones(nperseg)
    raise ValueError('unknown window %r' % (name,))


def _segments(x, nperseg, noverlap):
    step = nperseg - noverlap
    nseg = (len(x) - noverlap) // step
    idx = np.arange(nperseg)[None, :] + step * np.arange(nseg)[:, None]
    return x[idx]


def csd_np(x, y, fs=1.0, window='hann', nperseg=256, noverlap=None, nfft=None,
           detrend='constant', scaling='density', chunk=512):
    """Two-sided cross spectral density ``mean_seg(conj(X) * Y) * scale`` in
    float64; with ``y is x`` this is Welch's PSD.  Segmentation, per-segment
    constant detrend, window, zero-padding to nfft, density / spectrum scaling
    and the mean over segments follow ``_spectral_helper``."""
    x = np.asarray(x).astype(np.complex128)
    same = y is None
    yy = x if same else np.asarray(y).astype(np.complex128)
    nperseg = int(nperseg)
    if nperseg > len(x) and isinstance(window, str):
        # scipy.signal._spectral_helper / _triage_segments: "nperseg = N is greater than input length = n, using
        # nperseg = n" - the window is then built for the shorter length and nfft keeps the caller's value.  This is
        # what fast_spectrum_scan(n_fft=0) runs into by construction: nFFT = 2^ceil(log2(npts)) >= npts
        # (ofdm_cr_tools.py:474-475 -> :214).
        nperseg = len(x)
    if noverlap is None:
        noverlap = nperseg // 2
    if nfft is None:
        nfft = nperseg
    win = get_window(window, nperseg) if isinstance(window, str) else np.asarray(window, float)
    if scaling == 'density':
        scale = 1.0 / (fs * (win * win).sum())
    elif scaling == 'spectrum':
        scale = 1.0 / win.sum() ** 2
    else:
        scale = 1.0
    step = nperseg - noverlap
    nseg = (len(x) - noverlap) // step
    acc = np.zeros(nfft, np.complex128)
    for s0 in range(0, nseg, chunk):
        s1 = min(nseg, s0 + chunk)
        idx = np.arange(nperseg)[None, :] + step * np.arange(s0, s1)[:, None]
        xs = x[idx]
        if detrend == 'constant':
            xs = xs - xs.mean(axis=1, keepdims=True)
        X = np.fft.fft(xs * win, nfft, axis=1)
        if same:
            acc += (np.conj(X) * X).sum(axis=0)
        else:
            ys = yy[idx]
            if detrend == 'constant':
                ys = ys - ys.mean(axis=1, keepdims=True)
            Y = np.fft.fft(ys * win, nfft, axis=1)
            acc += (np.conj(X) * Y).sum(axis=0)
    acc *= scale / nseg
    freqs = np.fft.fftfreq(nfft, 1.0 / fs)
    return freqs, (acc.real if same else acc)


def welch_np(x, fs=1.0, window='hann', nperseg=256, noverlap=None, nfft=None,
             detrend='constant', scaling='density'):
    return csd_np(x, None, fs, window, nperseg, noverlap, nfft, detrend, scaling)


def welch_c64(x, fs=1.0, window='hann', nperseg=256, noverlap=None, nfft=None, y=None):
    """The reference's arithmetic AS IT RUNS: GNU Radio hands ``sg.welch`` complex64 samples
    (ofdm_cr_tools.py:214,322,342; spectrum_sweeper.py:263) and SciPy keeps complex64 in single precision - segment
    mean, window product and pocketfft all in float32.  Tests use it to tell "differs from the float64 restatement
    by what single precision costs the reference itself" from a defect: with a DC line far above the noise the
    float32 REPRESENTATION of the segment mean (relative 6e-8, times sum(w) in bin 0) already costs the reference
    1e-4 ... 5e-3 of bins 0, +-1 on a one-segment input.  ``y``: cross spectrum (sg.csd).  -> float64 copy of the
    float32 / complex64 result."""
    import scipy.signal as sg
    x = np.asarray(x, np.complex64)
    if y is None:
        _, p = sg.welch(x, fs=fs, window=window, nperseg=nperseg, noverlap=noverlap, nfft=nfft, return_onesided=False)
        return p.astype(np.float64)
    _, p = sg.csd(x, np.asarray(y, np.complex64), fs=fs, window=window, nperseg=nperseg, noverlap=noverlap, nfft=nfft,
                  return_onesided=False)
    return p.astype(np.complex128)


def coherence_np(x, y, fs=1.0, window='hann', nperseg=256, noverlap=None, nfft=None,
                 detrend='constant'):
    """|Pxy|^2 / (Pxx Pyy) - the semantics of ``scipy.signal.coherence``
    (producer of coherence_detector's first input)."""
    f, pxx = welch_np(x, fs, window, nperseg, noverlap, nfft, detrend)
    _, pyy = welch_np(y, fs, window, nperseg, noverlap, nfft, detrend)
    _, pxy = csd_np(x, y, fs, window, nperseg, noverlap, nfft, detrend)
    return f, np.abs(pxy) ** 2 / (pxx * pyy), pxx, pyy, pxy


def src_power_welch(vector, npts, nFFT, Fr, Sf, bb_freqs, srch_bins):
    """ofdm_cr_tools.py:213-230: sg.welch(flattop, nperseg=nfft=nFFT) ->
    fftshift -> channel sums."""
    welch_axis, psd_welch = welch_np(vector, fs=Sf, window='flattop', nperseg=nFFT, nfft=nFFT)
    psd_al = np.fft.fftshift(psd_welch)
    axis_al = np.fft.fftshift(welch_axis)
    return psd_al, axis_al, _channel_sums(psd_al, Fr, Sf, bb_freqs, srch_bins)


def welch_plot_dB(data, Sf, fc, nfft):
    """ofdm_cr_tools.py:321-326 (default Hann, default 50 % overlap)."""
    axis, psd = welch_np(data, fs=Sf, nperseg=nfft, nfft=nfft)
    psd_al = np.fft.fftshift(psd)
    axis_al = np.fft.fftshift(axis)
    return [a + fc for a in axis_al], [10 * math.log10(p + 1e-20) for p in psd_al]


def welch_power_estimate(vector, nFFT, Sf):
    """ofdm_cr_tools.py:341-345."""
    _, psd = welch_np(vector, fs=Sf, nperseg=nFFT, nfft=nFFT)
    return float(sum(np.fft.fftshift(psd)))


def fast_spectrum_scan(vct_sample, fc, channel_rate, srch_bw, n_fft, samp_rate, method,
                       thr_leveler, noise_estimate, alpha_avg):
    """ofdm_cr_tools.py:471-537 without the plotting branch."""
    npts = len(vct_sample)
    nFFT = int(2 ** math.ceil(math.log(npts, 2))) if n_fft == 0 else n_fft
    Fr = float(samp_rate) / float(nFFT)
    Fstart = fc - _py2div(samp_rate, 2)
    Ffinish = fc + _py2div(samp_rate, 2)
    bb_freqs = frange(_py2div(-samp_rate, 2), _py2div(samp_rate, 2), channel_rate)
    srch_bins = srch_bw / Fr
    if method == 'welch':
        psd, axis, plc = src_power_welch(vct_sample, npts, nFFT, Fr, samp_rate, bb_freqs, srch_bins)
    elif method == 'fft':
        psd, axis, plc = src_power_fft(vct_sample, npts, nFFT, Fr, samp_rate, bb_freqs, srch_bins)
    else:
        raise ValueError(method)
    ax_ch = frange(Fstart, Ffinish, channel_rate)
    min_power = np.amin(plc)
    noise_estimate = (1 - alpha_avg) * noise_estimate + alpha_avg * min_power
    thr = noise_estimate * thr_leveler
    constraint = [ax_ch[i] for i, item in enumerate(plc) if item > thr]
    return thr, plc, noise_estimate, constraint


# --------------------------------------------------------------------------
# spectrum_sweeper (python/spectrum_sweeper.py)
# --------------------------------------------------------------------------


def sweeper_geometry(fft_len, sample_rate, trunc_sample_rate, fstart, ffinish, t_obs_ms):
    """spectrum_sweeper.py:62-70: probe length, tune frequencies, excess bins."""
    t_obs = t_obs_ms * 1e-3
    vector_probe_pts = int(2 ** math.ceil(math.log(sample_rate * t_obs, 2)))
    tune = frange_le(fstart + _py2div(trunc_sample_rate, 2), ffinish, trunc_sample_rate)
    if len(tune) < 1:
        tune = [_py2div(fstart + ffinish, 2)]
    freq_resolution = float(sample_rate) / float(fft_len)
    excess_bins = int(math.floor(_py2div(sample_rate - trunc_sample_rate, 2) / freq_resolution))
    return vector_probe_pts, tune, excess_bins


def sweeper_src_power(vector, nFFT, samp_rate, excess_bins):
    """spectrum_sweeper.py:260-276: welch(flattop, nperseg=nFFT/4, nfft=nFFT)
    -> fftshift -> trim -> 10 log10."""
    _, psd = welch_np(vector, fs=samp_rate, window='flattop', nperseg=int(nFFT / 4.0), nfft=nFFT)
    psd = np.fft.fftshift(psd)
    if excess_bins > 0:
        psd = psd[excess_bins:-excess_bins]
    return 10 * np.log10(psd)


def sweeper_stitch(vectors, nFFT, samp_rate, excess_bins, average):
    """spectrum_sweeper.py:207-231: concatenate per-segment PSDs in tune order;
    the blend with ``psd_old`` (re-initialised to 1e-10 each sweep, :213) is
    kept as written."""
    psd = np.array([])
    psd_old = np.array([1e-10] * (nFFT - excess_bins * 2) * len(vectors))
    for v in vectors:
        psd = np.concatenate((psd, sweeper_src_power(v, nFFT, samp_rate, excess_bins)), axis=0)
    psd = (1 - average) * psd + average * psd_old
    return psd


def sweeper_fragments(data, max_tu):
    """spectrum_sweeper.py:240-258 framing: [n_frags u8][frag_id u8][payload].
    ``fragments = int(ceil(len/max_tu)) + 1`` with Python-2 integer ``/``
    (so effectively floor + 1, :242)."""
    fragments = int(math.ceil(len(data) // max_tu)) + 1
    frames = []
    j = 0
    for i in range(fragments):
        frag = data[j:j + max_tu]
        if i == fragments - 1:
            frag = data[j:]
        frames.append(struct.pack('!B', fragments) + struct.pack('!B', i) + frag)
        j += max_tu
    return frames


def worker_fragments(fft_data, max_tu, fft_len, data_precision):
    """local_worker.py:147-172 framing; float32 passthrough or int8 cast."""
    fft_data = np.asarray(fft_data, np.float32)
    if data_precision:
        fragments = int(math.ceil(fft_len * 4 / float(max_tu)))
    else:
        fft_data = fft_data.astype(np.int8)
        fragments = int(math.ceil(fft_len / float(max_tu)))
    data = fft_data.tobytes()
    frames = []
    j = 0
    for i in range(fragments):
        frag = data[j:j + max_tu]
        if i == fragments - 1:
            frag = data[j:]
        frames.append(struct.pack('!B', fragments) + struct.pack('!B', i) + frag)
        j += max_tu
    return frames


def consumer_handler(frames, data_type, header=0, clear_on_error=False):
    """The consumers' reassembly, frame by frame: remote_client_qt.py:100-164 (header=0; on a decode error the
    pending string is kept, :161-162) and sdr_webserver/sdr_webserver_ws.py:235-287 (header=10 = ``msg_str[10:]``
    :241; the pending string is cleared on error, :279).  -> (list of decoded vectors in completion order,
    final max_fft_data)."""
    pending = b''
    max_fft_data = np.array([])
    strt = True
    out = []
    for msg in frames:
        msg = bytes(msg)[header:]
        n_frags = struct.unpack('!B', msg[0:1])[0]
        frag_id = struct.unpack('!B', msg[1:2])[0]
        body = msg[2:]
        if n_frags == 1:
            try:
                fft_data = np.frombuffer(body, data_type)
            except ValueError:
                continue
        else:
            pending += body
            if frag_id != n_frags - 1:
                continue
            try:
                fft_data = np.frombuffer(pending, data_type)
            except ValueError:
                if clear_on_error:
                    pending = b''
                continue
            pending = b''
        if len(max_fft_data) != len(fft_data):
            max_fft_data = fft_data
        if strt:
            max_fft_data = fft_data
            strt = False
        max_fft_data = np.maximum(max_fft_data, fft_data)
        out.append(fft_data)
    return out, max_fft_data


def zmq_pdu_header(nbytes):
    """PMT serialisation of cons(PMT_NIL, u8vector(nbytes)) up to the first data byte (GNU Radio pmt_serialize.cc:
    PST_PAIR 0x07, PST_NULL 0x06, PST_UNIFORM_VECTOR 0x0a, UVI_U8 0x00, uint32 BE count, npad 1, pad 0) - the ten
    bytes sdr_webserver_ws.py:241 strips."""
    return bytes([7, 6, 10, 0]) + struct.pack('>I', nbytes) + bytes([1, 0])


# --------------------------------------------------------------------------
# GNU Radio 3.7 block semantics (parity unpinned, see module docstring)
# --------------------------------------------------------------------------


def gr_blackmanharris(ntaps):
    """``gnuradio.filter.window.blackmanharris(ntaps)`` (psd_logger.py:47,
    local_worker.py:62): 4-term, 92 dB, symmetric (denominator ntaps-1)."""
    a = [0.35875, 0.48829, 0.14128, 0.01168]
    n = np.arange(ntaps)
    f = 2.0 * np.pi * n / (ntaps - 1)
    return a[0] - a[1] * np.cos(f) + a[2] * np.cos(2 * f) - a[3] * np.cos(3 * f)


def gr_decimation(sample_rate, fft_len, rate):
    """keep_one_in_n argument, spectrum_sensor_v2.py:86-87 (``int(a/b/c)``)."""
    return max(1, int(_py2div(_py2div(sample_rate, fft_len), rate)))


def gr_kept_vectors(x, fft_len, n):
    """stream_to_vector(fft_len) then keep_one_in_n(n): the LAST of every n
    vectors (indices n-1, 2n-1, ...)."""
    nvec = len(x) // fft_len
    v = np.asarray(x[:nvec * fft_len]).reshape(nvec, fft_len)
    return v[n - 1::n]


def gr_fft_vcc(vecs, window=None, shift=True):
    """fft_vcc(N, forward=True, window, shift): unnormalised forward FFT of
    in*window; shift=True swaps the output halves (fftshift)."""
    v = np.asarray(vecs).astype(np.complex128)
    if window is not None and len(window):
        v = v * np.asarray(window, float)[None, :]
    X = np.fft.fft(v, axis=1)
    return np.fft.fftshift(X, axes=1) if shift else X


def chain_sensor_v2(x, fft_len, decim=1):
    """spectrum_sensor_v2.py:85-93,116 (= multichannel_scanner.py:78-86,100):
    rectangular window, shifted FFT, |.|^2, x 1/N^2, per kept vector."""
    X = gr_fft_vcc(gr_kept_vectors(x, fft_len, decim), None, True)
    return (X.real ** 2 + X.imag ** 2) * (1.0 / float(fft_len ** 2))


def chain_sensor_v2_mean_c64(x, fft_len):
    """The same chain as GNU Radio RUNS it - single precision throughout (fft_vcc is FFTW3f, the magnitude and 1/N^2
    blocks are float32 VOLK kernels) - averaged over all vectors of the stream: the CPU counterpart of BASELINE
    config 5's per-channel PSD (multichannel_scanner.py:78-86 + the mean).  scipy.fft keeps complex64 in single
    precision; one (rows, N) batch per call, rows accumulated in float64.  bench.py times this (cpu_baseline_c5)."""
    import scipy.fft as sfft
    v = gr_kept_vectors(np.asarray(x, np.complex64), fft_len, 1)
    X = sfft.fft(v, axis=1)
    p = (X.real * X.real + X.imag * X.imag) * np.float32(1.0 / float(fft_len ** 2))
    return np.fft.fftshift(p.mean(axis=0, dtype=np.float64))


def chain_psd_logger(x, fft_len, decim=1):
    """psd_logger.py:43-56,85: Blackman-Harris FFT (no shift), magnitude,
    running peak.  The first vector initialises the peak."""
    X = gr_fft_vcc(gr_kept_vectors(x, fft_len, decim), gr_blackmanharris(fft_len), False)
    mag = np.abs(X)
    return mag, np.maximum.accumulate(mag, axis=0)


def chain_local_worker(x, fft_len, sample_rate, average, decim=1):
    """local_worker.py:58-71,79: BH-windowed shifted FFT, |.|^2,
    single_pole_iir_filter_ff(average) (y = a x + (1-a) y_prev, y_-1 = 0),
    nlog10_ff(10, N, -10log10(N) - 10log10(Sf))."""
    X = gr_fft_vcc(gr_kept_vectors(x, fft_len, decim), gr_blackmanharris(fft_len), True)
    p = X.real ** 2 + X.imag ** 2
    k = -10 * math.log10(fft_len) - 10 * math.log10(sample_rate)
    y = np.zeros(fft_len)
    lin = np.empty_like(p)
    for i in range(p.shape[0]):
        y = average * p[i] + (1.0 - average) * y
        lin[i] = y
    return lin, 10 * np.log10(lin) + k


# --------------------------------------------------------------------------
# Channel scanner state machines
# --------------------------------------------------------------------------


class ScannerState(object):
    """stats_watcher.__init__ / spectrum_scanner, spectrum_sensor_v2.py:357-393
    and :445-479 (same arithmetic in basic_spectrum_watcher :482-544 and
    multichannel_scanner.py:177-224)."""

    def __init__(self, fft_len, sample_rate, channel_space, search_bw, tune_freq=0,
                 trunc_band=1, thr_leveler=10, alpha_avg=1):
        self.fft_len = fft_len
        self.sample_rate = sample_rate
        self.thr_leveler = thr_leveler
        self.alpha_avg = alpha_avg
        self.noise_estimate = 1e-11
        self.trunc = sample_rate - trunc_band
        self.trunc_ch = _py2div(int(_py2div(self.trunc, channel_space)), 2)
        self.Fr = float(sample_rate) / float(fft_len)
        self.Fstart = tune_freq - _py2div(sample_rate, 2)
        self.Ffinish = tune_freq + _py2div(sample_rate, 2)
        self.bb_freqs = frange(_py2div(-sample_rate, 2), _py2div(sample_rate, 2), channel_space)
        self.srch_bins = search_bw / self.Fr
        self.ax_ch = frange(self.Fstart, self.Ffinish, channel_space)
        if self.trunc > 0:
            self.ax_ch = self.ax_ch[self.trunc_ch:-self.trunc_ch]
        self.plc = np.array([0.0] * len(self.ax_ch))
        self.cumulative_max_power = None
        self.threshold = 0.0

    def scan(self, samples):
        plc = src_power(samples, self.fft_len, self.Fr, self.sample_rate, self.bb_freqs, self.srch_bins)
        if self.trunc > 0:
            plc = plc[self.trunc_ch:-self.trunc_ch]
        self.plc = self.plc * 0.6 + np.array(plc) * 0.4
        self.cumulative_max_power = (np.array(plc) if self.cumulative_max_power is None
                                     else np.maximum(plc, self.cumulative_max_power))
        min_power = np.amin(plc)
        self.noise_estimate = (1 - self.alpha_avg) * self.noise_estimate + self.alpha_avg * min_power
        self.threshold = self.noise_estimate * self.thr_leveler
        return plc, [self.ax_ch[i] for i, item in enumerate(plc) if item > self.threshold]


def publish_top4(plc, ax_ch, subject_channels):
    """output_data.publish, spectrum_sensor_v2.py:228-237 (exact-match
    ``list.index`` lookup :218-220; argsort top-4, descending)."""
    idx = [ax_ch.index(ch) for ch in subject_channels]
    pwr = np.array([10 * np.log10(plc[i]) for i in idx])
    ff = pwr.argsort()[-4:][::-1]
    return pwr, [subject_channels[i] for i in ff]


def peak_hold(rows):
    """psd_watcher.run, spectrum_sensor_v2.py:351-354: bin-wise running max."""
    return np.maximum.accumulate(np.asarray(rows), axis=0)


def find_nearest_index(array, value):
    """coherence_detector.py:276-278."""
    return int((np.abs(np.asarray(array) - value)).argmin())


def coherence_axis(N, sample_rate, tune_freq):
    """coherence_detector.py:184-188: ``range(-N/2, N/2) * Fr + tune_freq``."""
    Fr = float(sample_rate) / float(N)
    return np.array(range(-(N // 2), N // 2)) * Fr + tune_freq


def coherence_scanner(data, data1, data2, idx_subject_channels, threshold, threshold_mtm):
    """coherence_detector.watcher.scanner, coherence_detector.py:254-274:
    2-bin sums at [ch-1, ch] and the three-way threshold decision."""
    coh, outcome, valve = [], [], []
    for ch in idx_subject_channels:
        c = data[(ch - 1):(ch + 1)].sum()
        l = data1[(ch - 1):(ch + 1)].sum()
        r = data2[(ch - 1):(ch + 1)].sum()
        coh.append(c)
        if c > threshold and l < threshold_mtm and r < threshold_mtm:
            outcome.append(1)
            valve.append(0)
        else:
            outcome.append(0.1)
            valve.append(1)
    return coh, outcome, valve


# --------------------------------------------------------------------------
# Synthetic IQ - shared by the golden generator, the tests and
# the bench CPU leg so that every party sees the same samples.
# --------------------------------------------------------------------------

TONES = ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071))
DC = 0.1 + 0.05j


def synth_iq(n, seed, tones=TONES, dc=DC, n0=0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.0)
    t = np.arange(n0, n0 + n, dtype=np.float64)
    for a, f in tones:
        x = x + a * np.exp(2j * np.pi * f * t)
    return (x + dc).astype(np.complex64)


# --------------------------------------------------------------------------
# The reference's own CPU call, for the bench's cpu_baseline leg only.
# --------------------------------------------------------------------------


def welch_reference_call(vector, nFFT, Sf):
    """Exactly what welch_power_estimate / welch_plot_dB execute on the CPU
    (ofdm_cr_tools.py:322,342): ``sg.welch(vector, fs=Sf, nperseg=nFFT, nfft=nFFT)``
    on the complex64 samples GNU Radio delivers (SciPy keeps them in single
    precision), then fftshift."""
    import warnings
    import scipy.signal as sg
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        axis, psd = sg.welch(vector, fs=Sf, nperseg=nFFT, nfft=nFFT)
    return np.fft.fftshift(psd)


# --------------------------------------------------------------------------
# flanck_detector (python/flanck_detector.py) - rising / falling edges of channel power
# --------------------------------------------------------------------------


class FlankState(object):
    """_queue0_watcher.__init__ / flank_detector, flanck_detector.py:262-311 and :345-399."""

    def __init__(self, fft_len, sample_rate, channel_space, search_bw, subject_channels, tune_freq=0,
                 trunc_band=1, thr_leveler=10, alpha_avg=1, peak_alpha=0):
        base = ScannerState(fft_len, sample_rate, channel_space, search_bw, tune_freq, trunc_band,
                            thr_leveler, alpha_avg)
        self.base = base
        self.ax_ch = base.ax_ch
        self.thr_leveler = thr_leveler
        self.alpha_avg = alpha_avg
        self.noise_estimate = 1e-11
        self.subject_channels = list(subject_channels)
        self.idx_subject_channels = [self.ax_ch.index(ch) for ch in self.subject_channels]
        n = len(self.subject_channels)
        self.prev_power = np.array([1.0] * n)
        self.curr_power = np.array([1.0] * n)
        self.flag = [True] * n
        self.peak_alpha = np.array([0.0] * n)
        self.peak_alpha_original = peak_alpha
        self.cumulative_statistics = {}

    def detect(self, samples):
        b = self.base
        plc = src_power(samples, b.fft_len, b.Fr, b.sample_rate, b.bb_freqs, b.srch_bins)
        if b.trunc > 0:
            plc = plc[b.trunc_ch:-b.trunc_ch]
        min_power = np.amin(plc)
        self.noise_estimate = (1 - self.alpha_avg) * self.noise_estimate + self.alpha_avg * min_power
        thr = self.noise_estimate * self.thr_leveler
        thr2 = thr * 20
        self.prev_power[:] = self.curr_power[:]
        events = []
        for k, channel in enumerate(self.idx_subject_channels):
            self.curr_power[k] = ((1 - self.peak_alpha[k]) * np.clip(plc[channel], 0, thr2)
                                  + self.peak_alpha[k] * self.prev_power[k])
            if self.curr_power[k] < thr2 and self.curr_power[k] > thr:
                self.curr_power[k] = thr2
            if self.curr_power[k] > self.prev_power[k] and self.curr_power[k] > thr and self.flag[k] is False:
                self.flag[k] = True
                self.peak_alpha[k] = 0
                f = self.ax_ch[channel]
                self.cumulative_statistics[f] = self.cumulative_statistics.get(f, 0) + 1
                events.append((f, +1))
            elif self.flag[k] is True and self.curr_power[k] < thr:
                self.flag[k] = False
                self.peak_alpha[k] = self.peak_alpha_original
                events.append((self.ax_ch[channel], -1))
        return events


# --------------------------------------------------------------------------
# python/ascii_plot.py - terminal plot of the latest dB row (consumer of the
# local_worker-style chain; rectangular window: the window of :61 is not passed)
# --------------------------------------------------------------------------


def chain_ascii_plot(x, fft_len, sample_rate, average, decim=1):
    """ascii_plot.py:57-70: as chain_local_worker but without a window."""
    vecs = gr_kept_vectors(x, fft_len, decim)
    p = np.abs(gr_fft_vcc(vecs, None, True)) ** 2
    y = np.zeros(fft_len)
    lin = []
    for r in p:
        y = average * r + (1 - average) * y
        lin.append(y.copy())
    lin = np.array(lin)
    k = -10 * math.log10(fft_len) - 10 * math.log10(sample_rate)
    return lin, 10 * np.log10(lin) + k


def ascii_make_plot(fft_data, width, height, tune_freq, sample_rate, fft_len):
    """ascii_plot.py:154-228 (ascii_plotter.__init__ + make_plot on a fresh matrix), Python-2 integer divisions."""
    axis = _py2div(sample_rate, 2) * np.linspace(-1, 1, fft_len) + tune_freq
    widthDens = len(axis) // int(width)
    matrix = [[' ' for _ in range(height)] for _ in range(width)]
    minValue, maxValue = min(fft_data), max(fft_data)
    span = math.floor((maxValue - minValue))
    out, aux = '', 0
    for i in range(width):
        ht = sum(fft_data[aux:aux + widthDens]) / widthDens
        n = int(math.floor(((ht - minValue) * (height - 1)) / span))
        for k in range(n + 1, height):
            matrix[i][k] = ' '
        matrix[i][n] = '^'
        for k in range(n):
            matrix[i][k] = '|'
        aux += widthDens
    for i in reversed(range(height)):
        matrix[width // 2][i] = '*'
        if i % 5 == 0:
            out += ('%.3f' % ((((i - 0) * span) / height) + minValue))[:6] + ' '
        else:
            out += '------ '
        for j in range(width):
            out += matrix[j][i] + ' '
        out += '\n'
    out += '------ '
    for a in range(width):
        if a % 10 == 0:
            out += '| ' + ('%.3f' % ((((a - 0) * (axis[-1] - axis[0])) / width) + axis[0]))[:5] + ' ' * (2 * 10 - 5 - 2)
    out += '\n'
    out += 'Tune freq: %s MHz, Sample rate: %s MS/s, FFT: %s W:%d L:%d\n' % (tune_freq / 1e6, sample_rate / 1e6,
                                                                             fft_len, width, height)
    out += '_ ' * width + '_ _ _ _'
    return out
