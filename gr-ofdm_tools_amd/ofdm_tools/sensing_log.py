"""logger - in-memory side of the sensing log (python/ofdm_cr_tools.py:1850-1956).

Holds the cumulative / periodic PSD peaks, per-channel max powers, occupancy
statistics and waterfall rows that the watchers of spectrum_sensor_v2 update, and
can write them in the reference's on-disk formats (np.save for arrays, ``str(dict)``
for statistics, ``%1.2e`` rows for the waterfall; ofdm_cr_tools.py:2029-2046) with
:meth:`flush`.  The background file_logger thread of the reference is not started
here: a host calls flush() when it wants the files.
"""
import os
import time

import numpy as np


class logger(object):
    def __init__(self, fft_len, periodicity, test_duration, directory=None):
        self.fft_len = fft_len
        self.periodicity = periodicity
        self.test_duration = test_duration
        self.directory = directory
        self.start_dat = time.strftime('%y%m%d')
        self.start_tim = time.strftime('%H%M%S')
        self.cumulative_statistics = {}
        self.settings = {}
        self.cumulative_psd = None
        self.cumulative_max_power = None
        self.cumulative_waterfall = []
        self.reset_periodic_vars()

    def reset_periodic_vars(self):
        self.periodic_psd_peaks = None
        self.periodic_statistic = {}
        self.periodic_max_power = None
        self.n_measurements_period = 0
        self.cumulative_waterfall = []

    def set_cumulative_psd(self, v):
        self.cumulative_psd = v

    def set_periodic_psd_peaks(self, v):
        self.periodic_psd_peaks = v

    def set_settings(self, v):
        self.settings = v

    def set_n_measurements_period(self, v):
        self.n_measurements_period = v

    def set_cumulative_statistics(self, v):
        self.cumulative_statistics = v

    def set_periodic_statistic(self, v):
        self.periodic_statistic = v

    def set_cumulative_max_power(self, v):
        self.cumulative_max_power = v

    def set_periodic_max_power(self, v):
        self.periodic_max_power = v

    def set_cumulative_waterfall(self, v):
        self.cumulative_waterfall = v

    def _path(self, stem, ext):
        return os.path.join(self.directory, '%s-%s-%s.%s' % (stem, self.start_dat, self.start_tim, ext))

    def flush(self, directory=None):
        """Write the cumulative files the way file_logger.run does (ofdm_cr_tools.py:2010-2046)."""
        if directory is not None:
            self.directory = directory
        if self.directory is None:
            raise ValueError('no log directory configured')
        os.makedirs(self.directory, exist_ok=True)
        out = {}
        if self.cumulative_psd is not None:
            out['psd'] = self._path('sdr_psd_cumulative_log', 'matz')
            with open(out['psd'], 'wb') as fh:
                np.save(fh, self.cumulative_psd)
        if self.cumulative_max_power is not None:
            out['max_power'] = self._path('sdr_max_power_cumulative_log', 'matz')
            with open(out['max_power'], 'wb') as fh:
                np.save(fh, self.cumulative_max_power)
        out['stat'] = self._path('sdr_ss_cumulative_log', 'log')
        with open(out['stat'], 'w') as fh:
            fh.write('Settings' + '\n' + str(self.settings) + '\n')
            fh.write('Statistics' + '\n' + str(self.cumulative_statistics) + '\n')
        if self.cumulative_waterfall:
            out['waterfall'] = self._path('sdr_waterfall_cumulative_log', 'matz')
            with open(out['waterfall'], 'ab') as fh:
                np.savetxt(fh, np.array(self.cumulative_waterfall), fmt='%1.2e', delimiter=',')
        return out
