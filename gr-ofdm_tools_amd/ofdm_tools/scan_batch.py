"""Batched multichannel scan (BASELINE config 5): many channel streams, one PSD each.

Every channel stream gets what multichannel_scanner computes per kept vector
(python/multichannel_scanner.py:78-86: rectangular window, shifted FFT, |.|^2 / N^2), averaged
over all its vectors, then a per-bin energy threshold against the moving-average noise floor
(the per-bin analogue of spectrum_sensor_v2.py:465-477) and the src_power channel sums
(python/ofdm_cr_tools.py:232-249).  Streams are independent: channel c goes to rank c mod world
and one all-gather returns the rows in channel order (ofdm_tools.sweep).
"""
import numpy as np

from . import _hip
from .scanner import ChannelScanner


class BatchScanPlan(object):
    def __init__(self, ctx, fft_len, sample_rate, channel_space, search_bw, thr_leveler=10, trunc_band=None):
        self.ctx = ctx
        self.fft_len = fft_len
        self.thr_leveler = thr_leveler
        self.plan = ctx.welch_plan(fft_len, noverlap=0, window=None, detrend=_hip.DETREND_NONE,
                                   scaling=_hip.SCALE_OVER_N2, fftshift=True)
        self.scanner = ChannelScanner(fft_len, sample_rate, channel_space, search_bw,
                                      trunc_band=sample_rate if trunc_band is None else trunc_band, ctx=ctx)

    def psd_rows_dev(self, iq_dptr, nsamples, nstreams, stream_stride, out_dptr):
        """Device in, device out: [nstreams][fft_len] averaged PSD rows; asynchronous."""
        return self.plan.exec_dev(iq_dptr, nsamples, out_dptr, nstreams=nstreams, stream_stride=stream_stride)

    def decide(self, rows):
        """rows: host float32 [nstreams][fft_len] -> (mask, noise floor per stream, channel powers per stream)."""
        mask, noise = self.ctx.bin_threshold(rows, self.scanner.srch_bins, self.thr_leveler)
        plc = np.array([self.scanner.channel_powers(r) for r in rows])
        return mask, noise, plc


def shard_channels(nch, rank, world):
    return list(range(rank, nch, world))
