"""local_worker (python/local_worker.py): Blackman-Harris shifted FFT -> |.|^2 ->
single_pole_iir_filter_ff(average) -> 10 log10 + k (:58-71,79), fragmented into PDUs of at most
max_tu bytes (:147-172).  Constructor as local_worker.py:37; setters as :85-109."""
import math

import numpy as np

from . import _hip, packets, windows
from .chain_block import ChainBlockMixin
from .gr_compat import pdu, sync_block
from .ofdm_cr_tools import _py2div


class local_worker(ChainBlockMixin, sync_block):
    def __init__(self, fft_len, sample_rate, average, rate, max_tu, data_precision, ctx=None, threaded=True):
        sync_block.__init__(self, 'local_worker', [np.complex64], None)
        self.fft_len = fft_len
        self.sample_rate = sample_rate
        self.average = average
        self.rate = rate
        self.max_tu = max_tu - 2                       # two bytes reserved for segmentation, :47
        self.data_precision = data_precision
        self.message_port_register_hier_out('pdus')
        self.ctx = ctx or _hip.default_context()
        self._chain_init(self.ctx.chain(fft_len, windows.blackmanharris(fft_len), True, _hip.EPI_MAG2,
                                        self._decimation()), threaded)
        # nlog10_ff's constant is fixed at construction in the reference (:67-69)
        self._k = -10 * math.log10(self.fft_len) - 10 * math.log10(self.sample_rate)
        self._set_iir()
        self.last_db = None

    def _decimation(self):
        return max(1, int(_py2div(_py2div(self.sample_rate, self.fft_len), self.rate)))     # :59-60

    def _set_iir(self):
        self._chain.set_iir_log(self.average, self._k)

    def _on_vector(self, row):
        """main_thread.run body (:126-139): the latest dB row goes out as PDU fragments."""
        self.last_db = row
        self.send_packet(row)

    def send_packet(self, db_row):
        for frame in packets.worker_fragments(db_row, self.max_tu, self.fft_len, self.data_precision):
            self.message_port_pub('pdus', pdu(frame))

    def set_rate(self, rate):
        self.rate = rate
        self._chain.set_keep_one_in_n(self._decimation())

    def set_sample_rate(self, sample_rate):
        self.sample_rate = sample_rate
        self.set_rate(self.rate)

    def set_average(self, average):
        self.average = average
        self._set_iir()

    def set_data_precision(self, data_precision):
        self.data_precision = data_precision

    def get_sample_rate(self):
        return self.sample_rate

    def get_average(self):
        return self.average
