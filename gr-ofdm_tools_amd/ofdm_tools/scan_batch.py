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

    def _slices(self):
        from .ofdm_cr_tools import _slice_bounds
        sc = self.scanner
        lo, hi = _slice_bounds(self.fft_len, sc.Fr, sc.sample_rate, sc.bb_freqs, sc.srch_bins)
        if sc.trunc > 0:
            lo, hi = lo[sc.trunc_ch:-sc.trunc_ch], hi[sc.trunc_ch:-sc.trunc_ch]
        return lo, hi

    def decide_dev(self, rows_dptr, nstreams, want_mask=True):
        """The decision stage on the device rows psd_rows_dev() left in HBM (oth_scan_decide_dev: one launch
        sequence, context-owned scratch, no copy of the rows) -> (mask uint8[nstreams][fft_len] or None,
        noise floor per stream, channel powers [nstreams][nch])."""
        lo, hi = self._slices()
        return self.ctx.scan_decide_dev(rows_dptr, nstreams, self.fft_len, self.scanner.srch_bins, self.thr_leveler,
                                        lo, hi, want_mask)

    def decide(self, rows):
        """Host rows (float32 [nstreams][fft_len]): uploaded once, then decide_dev()."""
        rows = np.ascontiguousarray(np.atleast_2d(rows), np.float32)
        d = self.ctx.alloc(rows.nbytes)
        try:
            self.ctx.h2d(d, rows)
            return self.decide_dev(d, rows.shape[0])
        finally:
            self.ctx.free(d)

    def scan_sharded(self, iq_dptr, nsamples, stream_stride, nch_total, rank, world, device, group=None):
        """BASELINE config 5 over `world` ranks: this rank holds channel streams rank, rank + world, ... back to back
        in HBM.  PSD rows and the decision stage run locally on the device; ONE all-gather returns rows, noise
        floors and channel powers in channel order on every rank (mask = rows > thr * noise is recomputed from
        them by whoever needs it).  -> (rows [nch][fft_len], noise [nch], power [nch][nchan]) torch tensors."""
        import torch
        from . import sweep
        mine = sweep.shard_segments(nch_total, rank, world)
        spr = sweep.segments_per_rank(nch_total, world)
        nchan = len(self._slices()[0])
        width = self.fft_len + 1 + nchan
        local = torch.zeros((spr, width), dtype=torch.float32, device=device)
        if mine:
            # rows, noise floors and channel powers stay on the device from the averaging kernel to the all-gather
            rows = torch.empty((len(mine), self.fft_len), dtype=torch.float32, device=device)
            noise = torch.empty(len(mine), dtype=torch.float32, device=device)
            power = torch.empty((len(mine), max(nchan, 1)), dtype=torch.float32, device=device)
            lo, hi = self._slices()
            sweep.torch_then_ctx(self.ctx, device)
            self.psd_rows_dev(iq_dptr, nsamples, len(mine), stream_stride, rows.data_ptr())
            self.ctx.scan_decide_dev_out(rows.data_ptr(), len(mine), self.fft_len, self.scanner.srch_bins,
                                         self.thr_leveler, lo, hi, noise.data_ptr(), power.data_ptr())
            sweep.ctx_then_torch(self.ctx)
            local[:len(mine), :self.fft_len] = rows
            local[:len(mine), self.fft_len] = noise
            if nchan:
                local[:len(mine), self.fft_len + 1:] = power
        allr = sweep.gather_rows(local, nch_total, rank, world, group)
        return allr[:, :self.fft_len], allr[:, self.fft_len], allr[:, self.fft_len + 1:]


def shard_channels(nch, rank, world):
    return list(range(rank, nch, world))
