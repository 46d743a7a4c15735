"""spectrum_sensor_v1 (python/spectrum_sensor_v1.py): the predecessor of spectrum_sensor_v2 - the same
FFT chain (:67-75,82) and stats watcher (:176-290: channel powers, max-hold, noise-floor threshold,
occupancy counts), always on, with optional PSD peak hold and waterfall; no top-4 output.
Constructor as spectrum_sensor_v1.py:40-42."""
from .spectrum_sensor_v2 import spectrum_sensor_v2


class spectrum_sensor_v1(spectrum_sensor_v2):
    def __init__(self, fft_len, sens_per_sec, sample_rate, channel_space=1, search_bw=1, thr_leveler=10,
                 tune_freq=0, alpha_avg=1, test_duration=1, period=3600, trunc_band=1, verbose=False,
                 psd=False, waterfall=False, subject_channels=[], ctx=None, log_directory=None):
        spectrum_sensor_v2.__init__(self, fft_len, sens_per_sec, sample_rate, channel_space, search_bw,
                                    thr_leveler, tune_freq, alpha_avg, test_duration, period, trunc_band, verbose,
                                    stats=True, psd=psd, waterfall=waterfall, output=False,
                                    subject_channels=subject_channels, ctx=ctx, log_directory=log_directory)
