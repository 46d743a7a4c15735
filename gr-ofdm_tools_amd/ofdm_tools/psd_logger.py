"""psd_logger (python/psd_logger.py): Blackman-Harris FFT magnitude with a running peak hold,
saved with np.save after every update (:43-56, :70-88).  Constructor as psd_logger.py:32.

Two reference slips are not reproduced: ``s`` used before assignment
(:79-81) and the first ``np.maximum(x, None)`` (:71,85) - the first vector initialises the peak."""
import time

import numpy as np

from . import _hip, windows
from .chain_block import ChainBlockMixin
from .gr_compat import sync_block
from .ofdm_cr_tools import _py2div


class psd_logger(ChainBlockMixin, sync_block):
    def __init__(self, fft_len, rate, sample_rate, ctx=None, mat_file=None, threaded=True):
        sync_block.__init__(self, 'psd_logger', [np.complex64], None)
        self.fft_len = fft_len
        self.rate = rate
        self.sample_rate = sample_rate
        self.ctx = ctx or _hip.default_context()
        self.decimation = max(1, int(_py2div(_py2div(sample_rate, fft_len), rate)))          # :44-45
        chain = self.ctx.chain(fft_len, windows.blackmanharris(fft_len), False, _hip.EPI_MAG, self.decimation)
        chain.set_peak_hold(True)
        self.mat_file = mat_file if mat_file is not None else \
            '/tmp/psd_log' + '-' + time.strftime('%y%m%d') + '-' + time.strftime('%H%M%S') + '.mat'
        self.peak_vals = None
        self._chain_init(chain, threaded)

    def _on_vector(self, row):
        """_queue_watcher.run body (:70-88): the peak vector lives on the device; fetch and save it."""
        self.peak_vals = self._chain.peak()
        if self.mat_file:
            np.save(self.mat_file, self.peak_vals)                                           # :88
