"""logger - in-memory side of the sensing log (python/ofdm_cr_tools.py:1850-1956).

Holds the cumulative / periodic PSD peaks, per-channel max powers, occupancy
statistics and waterfall rows that the watchers of spectrum_sensor_v2 update, and
can write them in the reference's on-disk formats (np.save for arrays, ``str(dict)``
for statistics, ``%1.2e`` rows for the waterfall; ofdm_cr_tools.py:2029-2046) with
:meth:`flush`.  :meth:`start_file_logger` runs the reference's periodic ``file_logger`` thread
(ofdm_cr_tools.py:1958-2107): write everything, start a new period, sleep ``periodicity`` seconds, stop
after ``test_duration`` with one last write; a host loop can instead call flush() when it wants the files.
"""
import datetime
import os
import threading
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
        self._fl_thread = None
        self._fl_stop = threading.Event()
        self.files_written = 0

    # -- file_logger thread (ofdm_cr_tools.py:1906, 1958-2107) ---------------------------------------------
    def start_file_logger(self):
        if self._fl_thread is not None:
            return
        self._prepare_file_logger()
        self._fl_stop.clear()
        self._fl_thread = threading.Thread(target=self._file_logger_run, daemon=True)
        self._fl_thread.start()

    def _prepare_file_logger(self):
        if self.directory is None:       # ~/sensing-<yymmdd>-<HHMM>/ (ofdm_cr_tools.py:1855-1859)
            self.directory = os.path.join(os.path.expanduser('~'), 'sensing-%s-%s' % (self.start_dat,
                                                                                         time.strftime('%H%M')))
        self.stop_time = datetime.datetime.now() + datetime.timedelta(seconds=self.test_duration)

    def _file_logger_run(self):
        while True:                      # file_logger.run (:2010-2071)
            self.flush()
            self.files_written += 1
            if self._fl_stop.wait(self.periodicity):
                break
            if datetime.datetime.now() > self.stop_time:
                break
        self.flush(new_period=False)     # "saves data for the last time" (:2073-2106): no reset afterwards
        self.files_written += 1

    def stop_file_logger(self):
        self._fl_stop.set()
        if self._fl_thread is not None:
            self._fl_thread.join(5.0)
            self._fl_thread = None

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

    def _path(self, stem, ext, dat=None, tim=None):
        return os.path.join(self.directory, '%s-%s-%s.%s' % (stem, dat or self.start_dat, tim or self.start_tim, ext))

    @staticmethod
    def _write_stats(path, settings, statistics):
        # file_logger writes the pair twice (ofdm_cr_tools.py:2013-2017)
        with open(path, 'w') as fh:
            for _ in range(2):
                fh.write('settings ' + str(settings) + '\n')
                fh.write('statistics ' + str(statistics) + '\n')

    def flush(self, directory=None, new_period=True):
        """One pass of file_logger.run (ofdm_cr_tools.py:2010-2058): rewrite the cumulative files, write the
        periodic ones, append the waterfall rows, then start a new period (new periodic file names, periodic
        state and waterfall buffer reset).  Returns the paths written."""
        if directory is not None:
            self.directory = directory
        if self.directory is None:
            raise ValueError('no log directory configured')
        os.makedirs(self.directory, exist_ok=True)
        if not hasattr(self, '_period_stamp'):
            self._period_stamp = (self.start_dat, self.start_tim)
        pd, pt = self._period_stamp
        out = {'stat': self._path('sdr_ss_cumulative_log', 'log'),
               'periodic_stat': self._path('sdr_ss_periodic_log', 'log', pd, pt),
               'psd': self._path('sdr_psd_cumulative_log', 'matz'),
               'periodic_psd': self._path('sdr_psd_periodic_log', 'matz', pd, pt),
               'max_power': self._path('sdr_max_power_cumulative_log', 'matz'),
               'periodic_max_power': self._path('sdr_max_power_periodic_log', 'matz', pd, pt),
               'waterfall': self._path('sdr_waterfall_cumulative_log', 'matz')}
        self._write_stats(out['stat'], self.settings, self.cumulative_statistics)
        self._write_stats(out['periodic_stat'], self.settings, self.periodic_statistic)
        for key, val in (('psd', self.cumulative_psd), ('periodic_psd', self.periodic_psd_peaks),
                         ('max_power', self.cumulative_max_power), ('periodic_max_power', self.periodic_max_power)):
            with open(out[key], 'wb') as fh:
                np.save(fh, val)            # np.save(None) stores an object array, as the reference does
        with open(out['waterfall'], 'ab') as fh:
            if len(self.cumulative_waterfall):
                np.savetxt(fh, np.array(self.cumulative_waterfall), fmt='%1.2e', delimiter=',')
        if new_period:
            self._period_stamp = (time.strftime('%y%m%d'), time.strftime('%H%M%S'))
            self.reset_periodic_vars()
        return out
