"""spectrum_sensor_v1 (python/spectrum_sensor_v1.py): the predecessor of spectrum_sensor_v2.

Same FFT chain (:67-75,82: stream_to_vector, keep_one_in_n, rectangular fft_vcc with shift, |X|^2, 1/N^2) and the
same loggers, but
  * the statistics watcher always runs (:96-98; v2 makes it optional) and scans on the raw channel powers only -
    there is no 0.6/0.4 smoothed copy (:246-278 against spectrum_sensor_v2.py:455), because
  * there is no output stage: no subject-channel ranking, no ``freq_out_*`` / ``freq_msg_PDU`` ports, no strobes,
    no ``set_freqs`` (:40-104 registers no message port).
PSD peak hold and waterfall are optional as in v2 (:100-104).  Constructor as :40-42.
"""
from .spectrum_sensor_v2 import spectrum_sensor_v2


class spectrum_sensor_v1(spectrum_sensor_v2):
    _block_name = 'spectrum_sensor_v1'
    _message_ports = ()

    def __init__(self, fft_len, sens_per_sec, sample_rate, channel_space=1, search_bw=1, thr_leveler=10,
                 tune_freq=0, alpha_avg=1, test_duration=1, period=3600, trunc_band=1, verbose=False,
                 psd=False, waterfall=False, subject_channels=[], ctx=None, threaded=True, log_directory=None):
        spectrum_sensor_v2.__init__(self, fft_len, sens_per_sec, sample_rate, channel_space, search_bw,
                                    thr_leveler, tune_freq, alpha_avg, test_duration, period, trunc_band, verbose,
                                    stats=True, psd=psd, waterfall=waterfall, output=False,
                                    subject_channels=subject_channels, ctx=ctx, threaded=threaded,
                                    log_directory=log_directory)

    def _stats_watcher(self, float_data):
        """_stats_watcher.run / spectrum_scanner, :208-278: as v2's, on the raw powers (no smoothed copy kept)."""
        plc_before = self._scanner.plc
        spectrum_sensor_v2._stats_watcher(self, float_data)
        self._scanner.plc = plc_before

    def set_freqs(self, *freqs):
        raise AttributeError('spectrum_sensor_v1 has no frequency outputs (python/spectrum_sensor_v1.py:40-104)')

    def publish(self):
        return None
