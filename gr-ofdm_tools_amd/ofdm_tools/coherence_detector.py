"""coherence_detector (python/coherence_detector.py) and the coherence_estimator that feeds it.

coherence_detector keeps the reference's interface: three float[N] inputs (coherence, MTM-L,
MTM-R; coherence_detector.py:45), last-vector rule (:236-241), per-subject-channel 2-bin sums
and the three-way threshold decision with ``valve_callback`` (:254-274).

The reference has no block that PRODUCES the coherence vector.
coherence_estimator is that producer: two complex64 streams -> Welch cross spectrum ->
magnitude-squared coherence |Pxy|^2 / (Pxx Pyy) (scipy.signal.coherence semantics, Hann,
nperseg = nfft = N, 50 % overlap, detrend constant), fftshifted so that bin order matches the
detector's axis ``range(-N/2, N/2) * Fr + tune_freq`` (:188).
"""
import numpy as np

from . import _hip, windows
from .gr_compat import sync_block, to_msg


def find_nearest_index(array, value):
    """coherence_detector.py:276-278."""
    return int((np.abs(np.asarray(array) - value)).argmin())


def find_nearest_value(array, value):
    return array[find_nearest_index(array, value)]


class coherence_detector(sync_block):
    def __init__(self, N, sample_rate, search_bw=1, threshold=10, threshold_mtm=0.2, tune_freq=0, alpha_avg=1,
                 test_duration=1, period=3600, stats=False, output=False, rate=10, subject_channels=[],
                 valve_callback=None):
        sync_block.__init__(self, 'coherence_detector', [(np.float32, N)] * 3, None)
        self.N = N
        self.sample_rate = sample_rate
        self.search_bw = search_bw
        self.threshold = threshold
        self.threshold_mtm = threshold_mtm
        self.tune_freq = tune_freq
        self.alpha_avg = alpha_avg
        self.output = output
        self.subject_channels = list(subject_channels)
        self.subject_channels_outcome = [0.1] * len(self.subject_channels)
        self.rate = rate
        self.valve_callback = valve_callback if valve_callback is not None else (lambda v: None)
        self.Fr = float(sample_rate) / float(N)
        self.srch_bins = int(search_bw / self.Fr / 2)
        self.ax_ch = np.array(range(-(N // 2), N // 2)) * self.Fr + tune_freq             # :188
        self.n_chans = len(self.subject_channels)
        self.idx_subject_channels = [find_nearest_index(self.ax_ch, ch) for ch in self.subject_channels]
        self._idx = np.array(self.idx_subject_channels, dtype=np.intp)
        self.subject_channels_coherence = [0] * self.n_chans

    def set_subject_channels_outcome(self, outcome):
        self.subject_channels_outcome = outcome

    def get_subject_channels_outcome(self):
        return self.subject_channels_outcome

    def work(self, input_items, output_items):
        v = [np.asarray(a, np.float32).reshape(-1, self.N) for a in input_items[:3]]
        n = min(len(a) for a in v)
        if n:
            self.scanner(v[0][n - 1], v[1][n - 1], v[2][n - 1])        # last vector of the call (:236-241)
        return n

    def _pair_sums(self, vector):
        """Sum of bins [idx - 1, idx] per subject channel (the slice ``[ch-1:ch+1]`` of coherence_detector.py:259-261).
        Python slice rules at the left edge are kept: idx 0 gives the slice [-1:1], which is empty -> 0."""
        idx = self._idx
        v = np.asarray(vector)
        return np.where(idx > 0, v[idx - 1] + v[idx], v.dtype.type(0))

    def scanner(self, data, data1, data2):
        """Decision stage of coherence_detector.py:254-274 for all subject channels at once: the detector fires where
        the coherence pair sum is above ``threshold`` while both MTM pair sums stay below ``threshold_mtm``; outcome
        1 where it fires, 0.1 elsewhere; the valve is driven once per channel, in channel order, 0 = fired."""
        coh = self._pair_sums(data)
        fired = (coh > self.threshold) & (self._pair_sums(data1) < self.threshold_mtm) \
            & (self._pair_sums(data2) < self.threshold_mtm)
        self.subject_channels_coherence = list(coh)
        for hit in fired:
            self.valve_callback(0 if hit else 1)
        outcome = [1 if hit else 0.1 for hit in fired]
        self.set_subject_channels_outcome(outcome)
        return outcome


class coherence_estimator(sync_block):
    """Two complex64 inputs -> fftshifted (Pxx, Pyy, Pxy, Cxy) every ``block_len`` samples."""

    def __init__(self, N, sample_rate, block_len=None, ctx=None):
        sync_block.__init__(self, 'coherence_estimator', [np.complex64, np.complex64], None)
        self.N = N
        self.sample_rate = sample_rate
        self.block_len = int(block_len if block_len is not None else 16 * N)
        self.ctx = ctx or _hip.default_context()
        self._plan = self.ctx.welch_plan(N, window=windows.get_window('hann', N), fs=float(sample_rate),
                                         fftshift=True)
        self._x = np.empty(0, np.complex64)
        self._y = np.empty(0, np.complex64)
        self.pxx = self.pyy = self.pxy = self.cxy = None
        self.message_port_register_out('coherence')

    def work(self, input_items, output_items):
        n = min(len(input_items[0]), len(input_items[1]))
        self._x = np.concatenate((self._x, np.asarray(input_items[0][:n], np.complex64)))
        self._y = np.concatenate((self._y, np.asarray(input_items[1][:n], np.complex64)))
        while len(self._x) >= self.block_len:
            self.pxx, self.pyy, self.pxy, self.cxy = self._plan.csd(self._x[:self.block_len],
                                                                    self._y[:self.block_len])
            self._x = self._x[self.block_len:]
            self._y = self._y[self.block_len:]
            self.message_port_pub('coherence', to_msg('coherence', self.cxy))      # (key . f32vector) pair
        return n
