"""ascii_plot (python/ascii_plot.py): terminal PSD plot.

Chain of the reference block (:57-70,78): stream_to_vector -> keep_one_in_n -> fft_vcc(N, True, (), True)
(the Blackman-Harris window of :61 is computed but NOT passed: rectangular) -> |.|^2 ->
single_pole_iir_filter_ff(average) -> nlog10_ff(10, N, -10 log10 N - 10 log10 Sf); the watcher renders the
latest dB row with ``ascii_plotter.make_plot`` (:169-228) and posts the text on ``pkt_out``.  The chain runs
as one fused HIP launch (oth_chain_*); the rendering is host text work on ``width`` columns.

Python-2 arithmetic of the reference kept on purpose: the bins per column and the centre column are integer
divisions.
"""
import math
import os

import numpy as np

from . import _hip
from .chain_block import ChainBlockMixin
from .gr_compat import sync_block, to_msg
from .ofdm_cr_tools import _py2div


class ascii_plotter(object):
    """Text rendering of one dB row (behaviour of ascii_plot.py:154-228, pinned byte for byte by
    tests/golden/ref_ascii_plot.npz, which the reference's own ``make_plot`` produced).

    The row is cut into ``width`` columns of ``bins_per_col = len(axis) // width`` bins (Python-2 integer division in
    the reference; trailing bins are not drawn).  A column's level is the mean of its bins scaled to
    ``0 .. height - 1`` over ``floor(max - min)`` dB; the column shows ``|`` below its level, ``^`` at it, the centre
    column is a line of ``*``.  Every fifth text row carries its dB value, every tenth column its frequency.
    One deviation: a level above the top row (possible because the scale uses floor(max - min)) is drawn in the top
    row; the reference indexes past its matrix there and its watcher thread dies with an IndexError."""

    def __init__(self, width, height, tune_freq, sample_rate, fft_len):
        self.width = width
        self.height = height
        self.tune_freq = tune_freq
        self.sample_rate = sample_rate
        self.fft_len = fft_len
        self.updateWindow()

    def set_axis(self, axis):
        self.axis = np.asarray(axis)
        self.widthDens = len(self.axis) // int(self.width)      # bins per column

    def updateWindow(self):
        self.set_axis(_py2div(self.sample_rate, 2) * np.linspace(-1, 1, self.fft_len) + self.tune_freq)

    def column_levels(self, row):
        """-> (levels[width] as ints, min, floor(max - min)) of a dB row."""
        d = np.asarray(row, np.float64)
        lo, hi = float(d.min()), float(d.max())
        span = math.floor(hi - lo)
        if span == 0:
            raise ZeroDivisionError('the row spans less than 1 dB')          # as the reference
        w, n = int(self.width), self.widthDens
        # left-to-right sums in double, the order Python's sum() takes (np.sum adds pairwise: last-bit differences
        # could move a level across a floor boundary)
        means = np.cumsum(d[:w * n].reshape(w, n), axis=1)[:, -1] / n
        levels = np.floor((means - lo) * (self.height - 1) / span).astype(int)
        return np.minimum(levels, self.height - 1), lo, span

    def make_plot(self, fft_data):
        w, h = int(self.width), int(self.height)
        levels, lo, span = self.column_levels(fft_data)
        rows = np.arange(h)[:, None]                                           # text row index, 0 = bottom
        grid = np.where(rows < levels, '|', np.where(rows == levels, '^', ' '))
        grid[:, w // 2] = '*'
        out = []
        for i in range(h - 1, -1, -1):
            label = ('%.3f' % (i * span / h + lo))[:6] if i % 5 == 0 else '------'
            out.append(label + ' ' + ' '.join(grid[i]) + ' \n')
        f0, f1 = self.axis[0], self.axis[-1]
        ticks = ''.join('| ' + ('%.3f' % (a * (f1 - f0) / w + f0))[:5] + ' ' * 13 for a in range(0, w, 10))
        out.append('------ ' + ticks + '\n')
        out.append('Tune freq: %s MHz, Sample rate: %s MS/s, FFT: %s W:%d L:%d\n'
                   % (self.tune_freq / 1e6, self.sample_rate / 1e6, self.fft_len, w, h))
        out.append('_ ' * w + '_ _ _ _')
        return ''.join(out)


class ascii_plot(ChainBlockMixin, sync_block):
    def __init__(self, fft_len, sample_rate, tune_freq, average, rate, width, height, ctx=None, threaded=True,
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
