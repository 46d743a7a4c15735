"""spectrum_sensor - the legacy gr.sync_block sensor (python/spectrum_sensor.py).

``work()`` keeps the latest ``block_length`` samples (:71-75); a PDU request on
``PDU from_cogeng`` (:77-120) runs ``fast_spectrum_scan`` ('SC' -> 'thre', 'nois', 'cons') or the
PAPR probe ('PAPR' -> 'papr') and answers on ``PDU spect_msg``.  Constructor as :37.
``log=True`` keeps the reference's request log (:59-62,:96-120): ``/tmp/ss_log-<yymmdd>-<HHMMSS>``, one
``Time,<HHMMSS>,<field>,<value>`` line per answered quantity."""
import os
import time

import numpy as np

from . import _hip
from .gr_compat import pdu_parts, sync_block, to_msg
from .ofdm_cr_tools import SpectrumScan, fast_spectrum_scan


class _RequestLog(object):
    """The CSV the reference writes when ``log`` is set: a header line with the block's geometry (:59-62), then per
    request ``Time,<HHMMSS>,<field>,<value>`` rows (:96-120).  Every row is flushed: the reference leaves the file to
    interpreter exit, which loses the tail of a campaign that is killed."""

    def __init__(self, directory, clock=None):
        self._clock = clock or (lambda fmt: time.strftime(fmt))
        self.path = os.path.join(directory, 'ss_log' + '-' + self._clock('%y%m%d') + '-' + self._clock('%H%M%S'))
        self._fh = open(self.path, 'w')

    def row(self, *fields):
        self._fh.write(','.join(['Time', self._clock('%H%M%S')] + [str(f) for f in fields]) + '\n')
        self._fh.flush()

    def close(self):
        self._fh.close()


class spectrum_sensor(sync_block):
    def __init__(self, block_length, sample_rate=1, fft_len=1, channel_space=1, search_bw=1, method='fft',
                 thr_leveler=10, tune_freq=0, alpha_avg=1, source=None, log=False, ctx=None, log_dir='/tmp',
                 async_scan=False):
        sync_block.__init__(self, 'spectrum_sensor', [np.complex64], None)
        # the reference's public attributes (:41-58), under its names
        self.__dict__.update(block_length=block_length, sample_rate=sample_rate, fft_len=fft_len, channel_space=channel_space,
                             search_bw=search_bw, method=method, thr_leveler=thr_leveler, tune_freq=tune_freq,
                             alpha_avg=alpha_avg, source=source, log=log)
        # state the requests fill in: the last captured block, PAPR, occupied channels, threshold, channel powers, noise
        self.__dict__.update(vector_sample=[0, 0], papr=1e-10, spectrum_constraint_hz=[], threshold=0, power_level_ch=[],
                             noise_estimate=1e-11)
        self.log_file = None
        if self.log:                                                                    # :59-62
            self.log_file = _RequestLog(log_dir)
            self.log_file.row('sample_rate', sample_rate, 'channel_space', channel_space, 'channel_bw', search_bw,
                              'tune_freq', tune_freq)
        self.ctx = ctx
        # async_scan (not in the reference, whose handler runs sg.welch on the scheduler's thread): an 'SC' request only
        # ENQUEUES the scan (oth_welch_exec_async) and the three answers go out from the first work() call that finds the
        # PSD ready - neither the message handler nor work() ever waits for the GPU
        self.async_scan = async_scan
        self._scan = None
        self.message_port_register_out('PDU spect_msg')
        self.message_port_register_in('PDU from_cogeng')
        self.set_msg_handler('PDU from_cogeng', self.cogeng_rx)

    def work(self, input_items, output_items):
        in0 = input_items[0][0:self.block_length]
        self.set_vector_sample(np.array(in0, np.complex64))      # the caller's buffer dies after the call
        if self._scan is not None:
            self.collect_scan(wait=False)
        return len(in0)

    def collect_scan(self, wait=True):
        """async_scan: publish the pending 'SC' answers if the scan has finished (wait=True: when it has).  -> True
        when nothing is pending any more."""
        scan = self._scan
        if scan is None:
            return True
        result = scan.wait(self.get_noise_estimate()) if wait else scan.poll(self.get_noise_estimate())
        if result is None:
            return False
        self._scan = None
        self.threshold, self.power_level_ch, self.noise_estimate, self.spectrum_constraint_hz = result
        self._publish_spectrum_constraint()
        return True

    # request name (the PDU's body as text) -> method that answers it (python/spectrum_sensor.py:88-120)
    _REQUESTS = {'PAPR': '_answer_papr', 'SC': '_answer_spectrum_constraint'}

    def cogeng_rx(self, msg):
        parts = pdu_parts(msg)
        if parts is None:               # "Message is not a valid PDU" (:77-83): no reply
            return
        getattr(self, self._REQUESTS.get(str(parts[1]), '_answer_unknown'))()

    def _answer_papr(self):
        self.set_papr(self.get_vector_sample())
        self.send_msg('papr', self.get_papr())
        if self.log:                                                                    # :96-98
            self.log_file.row('tune_freq', self.get_tune_freq())
            self.log_file.row('papr', self.get_papr())

    def _answer_spectrum_constraint(self):
        if self.async_scan:
            if self._scan is not None:          # a request while one is in flight: answer the older one first
                self.collect_scan(wait=True)
            self._scan = SpectrumScan(self.get_vector_sample(), self.tune_freq, self.channel_space, self.search_bw,
                                      self.fft_len, self.sample_rate, self.method, self.thr_leveler, self.get_alpha_avg(),
                                      self.ctx)
            return
        self.set_spectrum_constraint_hz(self.get_vector_sample())
        self._publish_spectrum_constraint()

    def _publish_spectrum_constraint(self):
        for field, value in (('thre', self.get_threshold()), ('nois', self.get_noise_estimate()),
                             ('cons', self.get_spectrum_constraint_hz())):
            self.send_msg(field, value)
        if self.log:                                                                    # :113-117
            self.log_file.row('tune_freq[Hz]', self.get_tune_freq())
            self.log_file.row('threshold[dB]', 10 * np.log10(self.get_threshold() + 1e-20))
            self.log_file.row('noise[dB]', 10 * np.log10(self.get_noise_estimate() + 1e-20))
            self.log_file.row('spectrum_constraint[Hz]', self.get_spectrum_constraint_hz())

    def _answer_unknown(self):
        self.send_msg('unkn', 'received unknown request')
        if self.log:                                                                    # :120
            self.log_file.row('received unknown request')

    def send_msg(self, meta, data):
        self.message_port_pub('PDU spect_msg', to_msg(meta, data))

    def set_spectrum_constraint_hz(self, measure):
        (self.threshold, self.power_level_ch, self.noise_estimate,
         self.spectrum_constraint_hz) = fast_spectrum_scan(measure, self.tune_freq, self.channel_space,
                                                           self.search_bw, self.fft_len, self.sample_rate,
                                                           self.method, self.thr_leveler, self.get_noise_estimate(),
                                                           self.get_alpha_avg(), False, self.ctx)

    def set_papr(self, measure):
        """:144-151 (scalar work on one captured block)."""
        measure = np.asarray(measure)
        mean_square = np.vdot(measure, measure) / len(measure)
        peak = max(measure * np.conjugate(measure))
        self.papr = 10 * np.log10((peak / mean_square).real + 1e-20)

    def _log_setter(self, field, value):
        """The setters the reference logs (:172-201): ``Time,<HHMMSS>,<field>,<value>``."""
        if self.log:
            self.log_file.row(field, value)


# The reference block's one-line accessors (python/spectrum_sensor.py:153-206) as a table: attribute, whether it has a
# getter / a setter there, and the field name under which the setter writes a request-log row (None: not logged).
# set_papr and set_spectrum_constraint_hz above are the two "setters" that compute.
_ACCESSORS = (
    ('papr', True, False, None), ('spectrum_constraint_hz', True, False, None), ('threshold', True, False, None),
    ('noise_estimate', True, False, None), ('power_level_ch', True, False, None),
    ('alpha_avg', True, True, None), ('vector_sample', True, True, None), ('block_length', False, True, None),
    ('time_observation', False, True, None), ('fft_len', False, True, None),
    ('tune_freq', True, True, 'set_tune_freq'), ('sample_rate', True, True, 'set_samp_rate'),
    ('channel_space', True, True, 'set_channel_space'), ('search_bw', True, True, 'set_search_bw'),
    ('thr_leveler', True, True, 'set_thr_leveler'),
)


def _install_accessors(cls):
    def getter(name):
        return lambda self: getattr(self, name)

    def setter(name, field):
        def set_(self, value):
            setattr(self, name, value)
            if field:
                self._log_setter(field, value)
        return set_
    for name, has_get, has_set, field in _ACCESSORS:
        if has_get:
            setattr(cls, 'get_' + name, getter(name))
        if has_set:
            setattr(cls, 'set_' + name, setter(name, field))


_install_accessors(spectrum_sensor)
