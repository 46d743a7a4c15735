"""multichannel_scanner (python/multichannel_scanner.py): the spectrum_sensor_v2 FFT chain
(:78-86,100) + channel powers + 0.6/0.4 EMA + dB + top-4 of the subject channels (:214-239).
Constructor as multichannel_scanner.py:46-47."""
import numpy as np

from . import _hip
from .chain_block import ChainBlockMixin
from .gr_compat import sync_block, to_msg
from .message_pdu import message_pdu
from .ofdm_cr_tools import _py2div
from .scanner import ChannelScanner, top4


class multichannel_scanner(ChainBlockMixin, sync_block):
    def __init__(self, fft_len, sens_per_sec, sample_rate, channel_space=1, search_bw=1, tune_freq=0,
                 trunc_band=1, verbose=False, output=False, subject_channels=[], ctx=None, threaded=True):
        sync_block.__init__(self, 'multichannel_scanner', [np.complex64], None)
        self.fft_len = fft_len
        self.sens_per_sec = sens_per_sec
        self.sample_rate = sample_rate
        self.channel_space = channel_space
        self.search_bw = search_bw
        self.tune_freq = tune_freq
        self.verbose = verbose
        self.trunc_band = trunc_band
        self.output = output
        self.subject_channels = list(subject_channels)
        self.subject_channels_pwr = np.array([1.0] * len(self.subject_channels))
        self.top4 = [self.subject_channels[0]] * 4                              # :67 (needs >= 1 channel)
        for port in ('freq_out_0', 'freq_out_1', 'freq_out_2', 'freq_out_3', 'freq_msg_PDU'):
            self.message_port_register_hier_out(port)
        self.ctx = ctx or _hip.default_context()
        self.decimation = max(1, int(_py2div(_py2div(sample_rate, fft_len), sens_per_sec)))
        chain = self.ctx.chain(fft_len, None, True, _hip.EPI_MAG2_OVER_N2, self.decimation)
        self.PDU_messages = message_pdu(None)
        self.PDU_messages.msg_connect('out', lambda m: self.message_port_pub('freq_msg_PDU', m))
        self._scanner = ChannelScanner(fft_len, sample_rate, channel_space, search_bw, tune_freq, trunc_band,
                                       ctx=self.ctx)
        self._idx_subject = self._scanner.subject_index(self.subject_channels)
        self._chain_init(chain, threaded)

    def _on_vector(self, row):
        """basic_spectrum_watcher.run body (:196-210)."""
        self._scanner.basic_scan(row)
        self.publish()

    def publish(self):
        """basic_spectrum_watcher.publish, :227-239."""
        if len(self.subject_channels) < 4:
            return
        self.subject_channels_pwr, best = top4(self._scanner.plc, self._idx_subject, self.subject_channels)
        self.top4 = list(best)
        self.set_freqs(best[0], best[1], best[2], best[3])

    def set_freqs(self, freq0, freq1, freq2, freq3):
        for i, f in enumerate((freq0, freq1, freq2, freq3)):
            self.message_port_pub('freq_out_%d' % i, to_msg('freq', f - self.tune_freq))

    def post_top4(self):
        """output_data.run body, :130-139: one PDU per top-4 frequency."""
        for f in self.top4:
            self.PDU_messages.post_message('freq', str(f))

    @property
    def power_level_ch(self):
        return self._scanner.plc
