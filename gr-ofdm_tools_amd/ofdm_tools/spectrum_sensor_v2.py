"""spectrum_sensor_v2 - multichannel energy detector (python/spectrum_sensor_v2.py).

Same constructor as the reference block (spectrum_sensor_v2.py:43-45, GRC make
template grc/ofdm_tools_spectrum_sensor_v2.xml:7-25).  The GNU Radio chain
stream_to_vector -> keep_one_in_n -> fft_vcc(rect, shift) -> |.|^2 -> x 1/N^2
(:85-93,116) runs as one HIP chain (oth_chain_*); the watcher logic (:357-479,
:186-237, :325-354) keeps its arithmetic and consumes the LAST vector of every
work() call, as the reference watchers do with their depth-2 message queues.
"""
import threading
import time

import numpy as np

from . import _hip
from .chain_block import ChainBlockMixin, MessageStrobe
from .gr_compat import sync_block, to_msg
from .message_pdu import message_pdu
from .ofdm_cr_tools import _py2div
from .scanner import ChannelScanner, top4
from .sensing_log import logger


class spectrum_sensor_v2(ChainBlockMixin, sync_block):
    _block_name = 'spectrum_sensor_v2'
    _message_ports = ('freq_out_0', 'freq_out_1', 'freq_out_2', 'freq_out_3', 'freq_msg_PDU')      # :78-82

    def __init__(self, fft_len, sens_per_sec, sample_rate, channel_space=1, search_bw=1, thr_leveler=10,
                 tune_freq=0, alpha_avg=1, test_duration=1, period=3600, trunc_band=1, verbose=False,
                 stats=False, psd=False, waterfall=False, output=False, subject_channels=[],
                 ctx=None, threaded=True, log_directory=None, strobe_period_ms=1000):
        sync_block.__init__(self, self._block_name, [np.complex64], None)
        self.fft_len = fft_len
        self.sens_per_sec = sens_per_sec
        self.sample_rate = sample_rate
        self.channel_space = channel_space
        self.search_bw = search_bw
        self.thr_leveler = thr_leveler
        self.tune_freq = tune_freq
        self.threshold = 0
        self.alpha_avg = alpha_avg
        self.verbose = verbose
        self.trunc_band = trunc_band
        self.stats = stats
        self.psd = psd
        self.waterfall = waterfall
        self.output = output
        self.subject_channels = list(subject_channels)
        self.top4 = [0, 0, 0, 0]
        for port in self._message_ports:
            self.message_port_register_hier_out(port)

        self.ctx = ctx or _hip.default_context()
        self.decimation = max(1, int(_py2div(_py2div(sample_rate, fft_len), sens_per_sec)))   # :86-87
        chain = self.ctx.chain(fft_len, None, True, _hip.EPI_MAG2_OVER_N2, self.decimation)
        # message_strobe x 4, 1000 ms (:108-111): set_freqs() only changes their message; start() runs them
        self._strobes = [MessageStrobe(lambda m, i=i: self.message_port_pub('freq_out_%d' % i, m),
                                       to_msg('freq', 0), strobe_period_ms) for i in range(4 if self._message_ports else 0)]
        self.PDU_messages = message_pdu(None)
        self.PDU_messages.msg_connect('out', lambda m: self.message_port_pub('freq_msg_PDU', m))

        self._logger = logger(fft_len, period, test_duration, directory=log_directory) \
            if (stats or psd or waterfall) else None
        self._scanner = ChannelScanner(fft_len, sample_rate, channel_space, search_bw, tune_freq, trunc_band,
                                       thr_leveler, alpha_avg, self.ctx)
        self._idx_subject = self._scanner.subject_index(self.subject_channels) if output else []
        self.subject_channels_pwr = np.array([1.0] * len(self.subject_channels))
        self._waterfall_group = 0
        self._lock = threading.Lock()
        self._chain_init(chain, threaded)      # work() / watcher plumbing: chain_block.ChainBlockMixin

    def start(self):
        """What top_block.start() sets going besides the stream: the four 1 Hz strobes (:108-111,125-129) and
        the periodic file_logger thread (ofdm_cr_tools.py:1906, 2010-2107)."""
        for st in self._strobes:
            st.start()
        if self._logger is not None:
            self._logger.start_file_logger()
        return True

    def stop(self):
        for st in self._strobes:
            st.stop()
        if self._logger is not None:
            self._logger.stop_file_logger()
        return ChainBlockMixin.stop(self)

    # -- watchers ---------------------------------------------------------------
    def _on_vector(self, float_data):
        with self._lock:
            if self.stats:
                self._stats_watcher(float_data)
            elif self.output:
                self._scanner.basic_scan(float_data)
            if self.psd:                                                       # psd_watcher.run :336-354
                lg = self._logger
                lg.set_cumulative_psd(float_data.copy() if lg.cumulative_psd is None
                                      else np.maximum(float_data, lg.cumulative_psd))
                lg.set_periodic_psd_peaks(float_data.copy() if lg.periodic_psd_peaks is None
                                          else np.maximum(float_data, lg.periodic_psd_peaks))
            if self.waterfall:
                # keep_one_in_n(sens_per_sec) on the PSD vector stream (:102,121-122) -> waterfall_watcher.run
                # (:304-322): one row per group of sens_per_sec vectors, counted over ALL vectors the chain produced
                # (the ones a lagging watcher dropped included).  The kept vector is the group's last; when that is
                # not the last vector of its work() call, the call's last vector (at most nrows - 1 later) stands in.
                # One row per work() call at most, also when the call spans two or more groups: the reference's
                # message_sink packs every kept vector of a scheduler call into ONE message and waterfall_watcher.run
                # decodes only its last item (:308-313), so there too a burst of kept vectors logs a single row and
                # the skipped groups are never made up (tests: test_waterfall_block_end_to_end, the chunk that spans
                # several groups).
                group = self.vector_rows_end // max(1, int(self.sens_per_sec))
                if group > self._waterfall_group:
                    self._waterfall_group = group
                    self._logger.cumulative_waterfall.append(float_data.copy())
                    self._logger.set_cumulative_waterfall(self._logger.cumulative_waterfall)
            if self.output:
                self.publish()

    def _stats_watcher(self, float_data):
        """stats_watcher.run / spectrum_scanner, :395-479."""
        sc, lg = self._scanner, self._logger
        occupied = sc.scan(float_data)
        self.threshold = sc.threshold
        lg.set_cumulative_max_power(sc.cumulative_max_power)
        lg.set_periodic_max_power(sc.periodic_max_power)
        lg.settings['n_measurements'] = lg.settings.get('n_measurements', 0) + 1
        lg.n_measurements_period += 1
        lg.settings['n_measurements_period'] = lg.n_measurements_period
        lg.settings['noise_estimate'] = sc.noise_estimate
        for el in occupied:
            lg.cumulative_statistics[el] = lg.cumulative_statistics.get(el, 0) + 1
            lg.periodic_statistic[el] = lg.periodic_statistic.get(el, 0) + 1
        if self.verbose:
            print('noise_estimate dB (channel)', 10 * np.log10(sc.noise_estimate + 1e-20))
        self.spectrum_constraint_hz = occupied

    def publish(self):
        """output_data.publish, :228-237."""
        if len(self.subject_channels) < 4:
            return
        self.subject_channels_pwr, best = top4(self._scanner.plc, self._idx_subject, self.subject_channels)
        self.set_freqs(best[0], best[1], best[2], best[3])

    def set_freqs(self, freq0, freq1, freq2, freq3):
        """:157-165 - the strobes carry the differential frequency."""
        for i, f in enumerate((freq0, freq1, freq2, freq3)):
            self.top4[i] = f
            msg = to_msg('freq', f - self.tune_freq)
            self._strobes[i].set_msg(msg)
            if not self._strobes[i].running:      # host-loop mode (no start()): emit now instead of at the next tick
                self.message_port_pub('freq_out_%d' % i, msg)

    def send_PDU_data(self):
        """send_PDU_data.run body, :178-183 (one round)."""
        for freq in self.top4:
            self.PDU_messages.post_message('freq', str(freq))

    # accessors the tests and hosts use
    @property
    def power_level_ch(self):
        return self._scanner.plc

    @property
    def noise_estimate(self):
        return self._scanner.noise_estimate

    @property
    def ax_ch(self):
        return self._scanner.ax_ch
