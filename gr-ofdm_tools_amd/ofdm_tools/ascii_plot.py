"""ascii_plot (python/ascii_plot.py): terminal PSD plot.

Chain of the reference block (:57-70,78): stream_to_vector -> keep_one_in_n -> fft_vcc(N, True, (), True)
(the Blackman-Harris window of :61 is computed but NOT passed: rectangular) -> |.|^2 ->
single_pole_iir_filter_ff(average) -> nlog10_ff(10, N, -10 log10 N - 10 log10 Sf); the watcher renders the
latest dB row with ``ascii_plotter.make_plot`` (:169-228) and posts the text on ``pkt_out``.  The chain runs
as one fused HIP launch (oth_chain_*); the rendering is host text work on ``width`` columns.

Python-2 arithmetic of the reference kept on purpose: ``widthDens = len(axis) / int(width)`` and
``matrix[width / 2]`` are integer divisions.
"""
import math
import os

import numpy as np

from . import _hip
from .chain_block import ChainBlockMixin
from .gr_compat import sync_block, to_msg
from .ofdm_cr_tools import _py2div


class ascii_plotter(object):
    def __init__(self, width, height, tune_freq, sample_rate, fft_len):
        self.width = width
        self.height = height
        self.tune_freq = tune_freq
        self.sample_rate = sample_rate
        self.fft_len = fft_len
        self.updateWindow()

    def set_axis(self, axis):
        self.axis = axis
        self.widthDens = len(self.axis) // int(self.width)
        self.matrix = [[' ' for x in range(self.height)] for y in range(self.width)]

    def updateWindow(self):
        self.set_axis(_py2div(self.sample_rate, 2) * np.linspace(-1, 1, self.fft_len) + self.tune_freq)   # :157,166

    def make_plot(self, fft_data):
        """ascii_plot.py:169-228."""
        minValue = min(fft_data)
        maxValue = max(fft_data)
        toClient = ''
        auxWidth = 0
        span = math.floor((maxValue - minValue))
        for i in range(self.width):
            htValue = sum(fft_data[auxWidth:auxWidth + self.widthDens]) / self.widthDens
            htValueNormed = int(math.floor(((htValue - minValue) * (self.height - 1)) / span))
            for k in range(htValueNormed + 1, self.height):
                self.matrix[i][k] = ' '
            self.matrix[i][htValueNormed] = '^'
            for k in range(htValueNormed):
                self.matrix[i][k] = '|'
            auxWidth += self.widthDens
        for i in reversed(range(self.height)):
            self.matrix[self.width // 2][i] = '*'
            if i % 5 == 0:
                NewValue = (((i - 0) * span) / self.height) + minValue
                toClient += ('%.3f' % NewValue)[:6] + ' '
            else:
                toClient += '------ '
            for j in range(self.width):
                toClient += self.matrix[j][i] + ' '
            toClient += '\n'
        toClient += '------ '
        for a in range(self.width):
            if a % 10 == 0:
                NewValue = (((a - 0) * (self.axis[-1] - self.axis[0])) / self.width) + self.axis[0]
                toClient += '| ' + ('%.3f' % NewValue)[:5] + ' ' * (2 * 10 - 5 - 2)
        toClient += '\n'
        toClient += 'Tune freq: %s MHz, Sample rate: %s MS/s, FFT: %s W:%d L:%d\n' % (
            self.tune_freq / 1e6, self.sample_rate / 1e6, self.fft_len, self.width, self.height)
        for a in range(self.width):
            toClient += '_ '
        toClient += '_ _ _ _'
        return toClient


class ascii_plot(ChainBlockMixin, sync_block):
    def __init__(self, fft_len, sample_rate, tune_freq, average, rate, width, height, ctx=None, threaded=False,
                 echo=False):
        sync_block.__init__(self, 'ascii plot', [np.complex64], None)
        self.fft_len = fft_len
        self.sample_rate = sample_rate
        self.average = average
        self.tune_freq = tune_freq
        self.rate = rate
        if width == 0 and height == 0:                                       # :45-48
            rows, columns = os.popen('stty size', 'r').read().split()
            self.height = int(rows) - 5
            self.width = int(columns) // 2 - 10
        else:
            self.height = height
            self.width = width
        self.echo = echo                      # the reference prints every plot (:148); off unless asked for
        self.message_port_register_hier_out('pkt_out')
        self.ctx = ctx or _hip.default_context()
        self._k = -10 * math.log10(self.fft_len) - 10 * math.log10(self.sample_rate)
        self._ascii_plotter = ascii_plotter(self.width, self.height, self.tune_freq, self.sample_rate, self.fft_len)
        chain = self.ctx.chain(fft_len, None, True, _hip.EPI_MAG2, self._decimation())
        chain.set_iir_log(self.average, self._k)
        self.last_plot = None
        self._chain_init(chain, threaded)

    def _decimation(self):
        return max(1, int(_py2div(_py2div(self.sample_rate, self.fft_len), self.rate)))     # :56-57

    def _on_vector(self, row):
        """main_thread.run body (:133-149)."""
        self.last_plot = self._ascii_plotter.make_plot(row)
        if self.echo:
            print(self.last_plot)
        self.message_port_pub('pkt_out', to_msg('ascii', self.last_plot))

    def set_rate(self, rate):
        self.rate = rate
        self._chain.set_keep_one_in_n(self._decimation())

    def set_width(self, width):
        self._ascii_plotter.width = width
        self._ascii_plotter.updateWindow()

    def set_height(self, height):
        self._ascii_plotter.height = height
        self._ascii_plotter.updateWindow()

    def set_sample_rate(self, sample_rate):
        self._ascii_plotter.sample_rate = sample_rate
        self._ascii_plotter.updateWindow()

    def set_tune_freq(self, tune_freq):
        self._ascii_plotter.tune_freq = tune_freq
        self._ascii_plotter.updateWindow()

    def set_average(self, average):
        self.average = average
        self._chain.set_iir_log(self.average, self._k)

    def get_tune_freq(self):
        return self.tune_freq

    def get_sample_rate(self):
        return self.sample_rate

    def get_average(self):
        return self.average
