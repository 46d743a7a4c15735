"""flanck_detector (python/flanck_detector.py): rising / falling edges of the power in a set of
subject channels.  Same FFT chain as spectrum_sensor_v2 (:237-245,249) on the HIP chain, channel
powers through ``src_power`` on the device, then the reference's edge logic
(_queue0_watcher.flank_detector, :345-399).  Constructor as flanck_detector.py:213-215."""
import numpy as np

from . import _hip
from .chain_block import ChainBlockMixin
from .gr_compat import sync_block
from .ofdm_cr_tools import _py2div
from .scanner import ChannelScanner
from .sensing_log import logger


class flanck_detector(ChainBlockMixin, sync_block):
    def __init__(self, fft_len, sens_per_sec, sample_rate, channel_space=1, search_bw=1, thr_leveler=10,
                 tune_freq=0, alpha_avg=1, test_duration=1, period=3600, trunc_band=1, verbose=False,
                 peak_alpha=0, subject_channels=[], ctx=None, log_directory=None, threaded=True):
        sync_block.__init__(self, 'flank detector', [np.complex64], None)
        self.fft_len = fft_len
        self.sens_per_sec = sens_per_sec
        self.sample_rate = sample_rate
        self.channel_space = channel_space
        self.search_bw = search_bw
        self.thr_leveler = thr_leveler
        self.tune_freq = tune_freq
        self.threshold = 0
        self.alpha_avg = alpha_avg
        self.peak_alpha_original = peak_alpha
        self.subject_channels = list(subject_channels)
        self.verbose = verbose
        self.ctx = ctx or _hip.default_context()
        self.decimation = max(1, int(_py2div(_py2div(sample_rate, fft_len), sens_per_sec)))      # :238-239
        chain = self.ctx.chain(fft_len, None, True, _hip.EPI_MAG2_OVER_N2, self.decimation)
        self._logger = logger(fft_len, period, test_duration, directory=log_directory)
        self._scanner = ChannelScanner(fft_len, sample_rate, channel_space, search_bw, tune_freq, trunc_band,
                                       thr_leveler, alpha_avg, self.ctx)
        self.ax_ch = self._scanner.ax_ch
        self.idx_subject_channels = self._scanner.subject_index(self.subject_channels)
        n = len(self.subject_channels)
        self.noise_estimate = 1e-11
        self.prev_power = np.array([1.0] * n)
        self.curr_power = np.array([1.0] * n)
        self.flag = [True] * n                                  # starts in pseudo-detection (:300)
        self.peak_alpha = np.array([0.0] * n)
        self.events = []
        self._chain_init(chain, threaded)      # work() / watcher plumbing: chain_block.ChainBlockMixin

    def _on_vector(self, row):
        """_queue0_watcher.run body (flanck_detector.py:320-343): the latest vector of a message."""
        self.flank_detector(row)
        lg = self._logger
        lg.settings['n_measurements'] = lg.settings.get('n_measurements', 0) + 1
        lg.n_measurements_period += 1
        lg.settings['noise_estimate'] = self.noise_estimate

    def flank_detector(self, samples):
        """flanck_detector.py:345-399."""
        plc = self._scanner.channel_powers(samples)
        min_power = np.amin(plc)
        self.noise_estimate = (1 - self.alpha_avg) * self.noise_estimate + self.alpha_avg * min_power
        thr = self.noise_estimate * self.thr_leveler
        thr2 = thr * 20                                          # second threshold limits fast growth
        self.threshold = thr
        self.prev_power[:] = self.curr_power[:]
        lg = self._logger
        for k, channel in enumerate(self.idx_subject_channels):
            self.curr_power[k] = ((1 - self.peak_alpha[k]) * np.clip(plc[channel], 0, thr2)
                                  + self.peak_alpha[k] * self.prev_power[k])
            if self.curr_power[k] < thr2 and self.curr_power[k] > thr:
                self.curr_power[k] = thr2
            if self.curr_power[k] > self.prev_power[k] and self.curr_power[k] > thr and self.flag[k] is False:
                self.flag[k] = True
                self.peak_alpha[k] = 0
                f = self.ax_ch[channel]
                lg.cumulative_statistics[f] = lg.cumulative_statistics.get(f, 0) + 1
                lg.periodic_statistic[f] = lg.periodic_statistic.get(f, 0) + 1
                self.events.append((f, +1))
            elif self.flag[k] is True and self.curr_power[k] < thr:
                self.flag[k] = False
                self.peak_alpha[k] = self.peak_alpha_original
                self.events.append((self.ax_ch[channel], -1))
