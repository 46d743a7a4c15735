"""Channel-power scanner shared by spectrum_sensor_v2 and multichannel_scanner.

Host-side state machine of stats_watcher / basic_spectrum_watcher
(python/spectrum_sensor_v2.py:357-393, :445-479, :482-544;
python/multichannel_scanner.py:177-239).  The per-channel powers come from the
device (``ofdm_cr_tools.src_power`` -> oth_channel_power); what stays here is what
the reference also does on a handful of floats per measurement: edge-channel
truncation, the 0.6/0.4 EMA, max-hold, noise-floor tracking, threshold decision
and the top-4 pick.
"""
import numpy as np

from .ofdm_cr_tools import _py2div, frange, src_power


class ChannelScanner(object):
    def __init__(self, fft_len, sample_rate, channel_space, search_bw, tune_freq=0, trunc_band=1,
                 thr_leveler=10, alpha_avg=1, ctx=None):
        self.ctx = ctx
        self.fft_len = fft_len
        self.sample_rate = sample_rate
        self.channel_space = channel_space
        self.search_bw = search_bw
        self.tune_freq = tune_freq
        self.thr_leveler = thr_leveler
        self.alpha_avg = alpha_avg
        self.noise_estimate = 1e-11                                   # spectrum_sensor_v2.py:371
        self.trunc_band = trunc_band
        self.trunc = sample_rate - trunc_band
        self.trunc_ch = _py2div(int(_py2div(self.trunc, channel_space)), 2)     # :375-376
        self.Fr = float(sample_rate) / float(fft_len)
        self.Fstart = tune_freq - _py2div(sample_rate, 2)
        self.Ffinish = tune_freq + _py2div(sample_rate, 2)
        self.bb_freqs = frange(_py2div(-sample_rate, 2), _py2div(sample_rate, 2), channel_space)
        self.srch_bins = search_bw / self.Fr
        self.ax_ch = frange(self.Fstart, self.Ffinish, channel_space)
        if self.trunc > 0:
            self.ax_ch = self.ax_ch[self.trunc_ch:-self.trunc_ch]
        self.plc = np.array([0.0] * len(self.ax_ch))
        self.threshold = 0.0
        self.cumulative_max_power = None
        self.periodic_max_power = None
        self.n_measurements = 0

    def channel_powers(self, psd):
        plc = src_power(psd, self.fft_len, self.Fr, self.sample_rate, self.bb_freqs, self.srch_bins, self.ctx)
        if self.trunc > 0:
            plc = plc[self.trunc_ch:-self.trunc_ch]
        return plc

    def basic_scan(self, psd):
        """basic_spectrum_watcher.spectrum_scanner (:533-544): powers + EMA only."""
        plc = self.channel_powers(psd)
        self.plc = self.plc * 0.6 + np.array(plc) * 0.4
        return plc

    def scan(self, psd):
        """stats_watcher.spectrum_scanner (:445-479) -> list of occupied channel frequencies."""
        plc = self.basic_scan(psd)
        self.cumulative_max_power = (np.array(plc) if self.cumulative_max_power is None
                                     else np.maximum(plc, self.cumulative_max_power))
        self.periodic_max_power = (np.array(plc) if self.periodic_max_power is None
                                   else np.maximum(plc, self.periodic_max_power))
        min_power = np.amin(plc)
        self.noise_estimate = (1 - self.alpha_avg) * self.noise_estimate + self.alpha_avg * min_power
        self.threshold = self.noise_estimate * self.thr_leveler
        self.n_measurements += 1
        return [self.ax_ch[i] for i, item in enumerate(plc) if item > self.threshold]

    def subject_index(self, subject_channels):
        """``ax_ch.index(channel)`` - exact float match, as the reference (:218-220)."""
        return [self.ax_ch.index(ch) for ch in subject_channels]


def top4(plc, idx_subject_channels, subject_channels):
    """output_data.publish (:228-237): dB of the subject channels, four strongest first."""
    pwr = np.array([10 * np.log10(plc[i]) for i in idx_subject_channels])
    ff = pwr.argsort()[-4:][::-1]
    return pwr, [subject_channels[i] for i in ff]
