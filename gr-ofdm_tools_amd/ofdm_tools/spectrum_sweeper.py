"""spectrum_sweeper (python/spectrum_sweeper.py): retune, capture, Welch each RF segment, stitch.

Constructor and setters as spectrum_sweeper.py:46-47 and :107-148.  The flowgraph side
(stream_to_vector(vector_probe_pts) -> keep_one_in_n -> message_sink, :86-89,97) is the
``work()`` below: it keeps the LAST captured vector.  ``sweep_once`` is one pass of
spectrum_stitcher.run (:207-231): per tune frequency ``_src_power`` (:260-276) =
welch(flattop, nperseg=nFFT/4, nfft=nFFT) -> fftshift -> trim excess_bins -> 10 log10, all in
one HIP plan; concatenate; blend with psd_old (re-initialised each sweep, :213); pack ``<f``;
fragment.  ``sweep_once_sharded`` runs the same sweep with one segment per rank and an
all-gather (ofdm_tools.sweep).
"""
import math
import struct
import time

import numpy as np

from . import _hip, packets, windows
from .gr_compat import pdu, sync_block
from .ofdm_cr_tools import _py2div


def frange(x, y, jump):
    """spectrum_sweeper.py:37-42 (inclusive)."""
    out = []
    while x <= y:
        out.append(x)
        x += jump
    return out


class spectrum_sweeper(sync_block):
    def __init__(self, rf_receiver, receiver_type, fft_len, sample_rate, trunc_sample_rate, fstart, ffinish,
                 rate, average, t_obs, tune_delay, max_tu, ctx=None):
        sync_block.__init__(self, 'spectrum_sweeper', [np.complex64], None)
        self.rf_receiver = rf_receiver
        self.receiver_type = receiver_type
        self.fft_len = fft_len
        self.sample_rate = sample_rate
        self.trunc_sample_rate = trunc_sample_rate
        self.fstart = fstart
        self.ffinish = ffinish
        self.rate = rate
        self.average = average
        self.max_tu = max_tu - 2
        self.t_obs = t_obs * 1e-3
        self.vector_probe_pts = int(2 ** math.ceil(math.log(sample_rate * self.t_obs, 2)))      # :63
        self.tune_delay = tune_delay * 1e-3
        self.tune_frequencies = frange(self.fstart + _py2div(self.trunc_sample_rate, 2), self.ffinish,
                                       self.trunc_sample_rate)                                   # :66
        if len(self.tune_frequencies) < 1:
            self.tune_frequencies = [_py2div(self.fstart + self.ffinish, 2)]
        self.freq_resolution = float(self.sample_rate) / float(self.fft_len)
        self.excess_bins = int(math.floor(_py2div(self.sample_rate - self.trunc_sample_rate, 2)
                                          / self.freq_resolution))                               # :69-70
        self.freq_axis = _py2div(self.sample_rate, 2) * np.linspace(-1, 1, self.fft_len)
        if self.excess_bins > 0:
            self.freq_axis = self.freq_axis[self.excess_bins:-self.excess_bins]
        self.fragments = int(math.ceil((self.fft_len * 4.0) / self.max_tu))
        self.samples = np.array([1e-10] * self.vector_probe_pts, np.complex64)                   # :84
        self.message_port_register_hier_out('pdus')
        self.ctx = ctx or _hip.default_context()
        self._decim = max(1, int(_py2div(_py2div(self.sample_rate, self.vector_probe_pts), self.rate)))
        self._count = self._decim
        self._partial = np.empty(0, np.complex64)
        nper = int(self.fft_len / 4.0)                                                           # :263
        self._plan = self.ctx.welch_plan(self.fft_len, nperseg=nper, window=windows.get_window('flattop', nper),
                                         fs=float(self.sample_rate), fftshift=True, trim_bins=self.excess_bins,
                                         db=True)
        self.psd = None

    # -- flowgraph side -----------------------------------------------------------
    def work(self, input_items, output_items):
        in0 = input_items[0]
        buf = np.concatenate((self._partial, in0)) if len(self._partial) else np.asarray(in0)
        n = self.vector_probe_pts
        nvec = len(buf) // n
        for i in range(nvec):                     # keep_one_in_n: the last of every _decim vectors
            self._count -= 1
            if self._count <= 0:
                self.set_samples(np.array(buf[i * n:(i + 1) * n], np.complex64))
                self._count = self._decim
        self._partial = np.array(buf[nvec * n:], np.complex64)
        return len(in0)

    # -- reference accessors ------------------------------------------------------
    def get_tune_delay(self):
        return self.tune_delay

    def set_tune_delay(self, tune_delay):
        self.tune_delay = tune_delay * 1e-3

    def get_samples(self):
        return self.samples

    def set_samples(self, samples):
        self.samples = samples

    def set_rate(self, rate):
        self.rate = rate
        self._decim = max(1, int(_py2div(_py2div(self.sample_rate, self.fft_len), self.rate)))   # :112
        self._count = self._decim

    def set_sample_rate(self, sample_rate):
        self.sample_rate = sample_rate
        self.set_rate(self.rate)

    def set_fstart(self, fstart):
        self.fstart = fstart

    def set_ffinish(self, ffinish):
        self.ffinish = ffinish

    def get_fstart(self, fstart=None):
        return self.fstart

    def get_ffinish(self, ffinish=None):
        return self.ffinish

    def get_sample_rate(self):
        return self.sample_rate

    def set_average(self, average):
        self.average = average

    def get_average(self):
        return self.average

    # -- stitcher -------------------------------------------------------------------
    def _src_power(self, vector):
        """spectrum_sweeper.py:260-276 on the device: dB PSD of one segment, shifted and trimmed."""
        return self._plan.exec(vector)

    def _blend_and_send(self, psd):
        psd_old = np.array([1e-10] * (self.fft_len - self.excess_bins * 2) * len(self.tune_frequencies))
        psd = (1 - self.average) * psd + self.average * psd_old                                 # :227
        self.psd = psd
        data = struct.pack('<%df' % len(psd), *psd)                                             # :229-230
        for frame in packets.sweeper_fragments(data, self.max_tu):
            self.message_port_pub('pdus', pdu(frame))
        return psd

    def sweep_once(self, sleep=time.sleep):
        """One iteration of spectrum_stitcher.run (:211-231)."""
        psd = np.array([])
        for f in self.tune_frequencies:
            try:
                self.rf_receiver.set_center_freq(f, 0)
            except Exception:
                print('cant tune receiver')
            sleep(self.tune_delay)
            psd = np.concatenate((psd, self._src_power(self.get_samples())), axis=0)
        return self._blend_and_send(psd)

    def sweep_once_sharded(self, capture, rank, world, device, group=None):
        """The same sweep with segment i on rank i mod world and one all-gather (RCCL over xGMI on GPUs).
        ``capture(i, f)`` returns what was observed at tune frequency f: host complex64 samples, or a torch
        complex64 / float32 [n][2] tensor already on ``device``.  Each segment goes through the HIP plan straight
        into this rank's row of the gather buffer (device in, device out); only the stitched wideband PSD
        comes back to the host, for the PDU fragments."""
        import torch
        from . import sweep
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError('sweep_once_sharded computes on the GPU: pass the rank\'s cuda device')
        nbins = self.fft_len - 2 * self.excess_bins

        def compute(iq, out_row):
            if not torch.is_tensor(iq):
                iq = torch.from_numpy(np.ascontiguousarray(iq, np.complex64).view(np.float32)).to(device)
            nsamples = iq.numel() // 2 if iq.dtype == torch.float32 else iq.numel()
            # a context on its own stream: torch's copy / the caller's producer kernels / the zero fill of the row
            # buffer must have landed before the plan reads and writes them, and the plan must be done before the
            # all-gather (torch's stream) reads the row
            sweep.torch_then_ctx(self.ctx, device)
            self._plan.exec_dev(iq.data_ptr(), nsamples, out_row.data_ptr())
            sweep.ctx_then_torch(self.ctx)

        wide = sweep.sweep_psd(lambda i: capture(i, self.tune_frequencies[i]), compute,
                               len(self.tune_frequencies), nbins, device, rank, world, group)
        return self._blend_and_send(wide.cpu().numpy().astype(np.float64))
