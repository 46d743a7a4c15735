"""spectrum_sweeper (python/spectrum_sweeper.py): retune, capture, Welch each RF segment, stitch.

Constructor and setters as spectrum_sweeper.py:46-47 and :107-148.  The flowgraph side
(stream_to_vector(vector_probe_pts) -> keep_one_in_n -> message_sink, :86-89,97) is the
``work()`` below: it keeps the LAST captured vector.  ``sweep_once`` is one pass of
spectrum_stitcher.run (:207-231): per tune frequency ``_src_power`` (:260-276) =
welch(flattop, nperseg=nFFT/4, nfft=nFFT) -> fftshift -> trim excess_bins -> 10 log10, all in
one HIP plan; concatenate; blend with psd_old (re-initialised each sweep, :213); pack ``<f``;
fragment.  ``sweep_once_sharded`` runs the same sweep with one segment per rank and an
all-gather (ofdm_tools.sweep).

Threads.  The reference starts ``data_colector`` and ``spectrum_stitcher`` from ``__init__``
(:99-105); the stitcher waits 2 s and then sweeps for as long as the flowgraph lives (:207-231),
with ``set_tune_delay`` / ``set_average`` forwarded to it (:110-112,:142-144).  Here ``work()`` IS
the collector (it stores the kept vector directly, no queue to pop) and ``_stitch_loop`` is the
stitcher: a daemon thread started by the constructor (``threaded=True``, the default) that calls
``sweep_once()`` until ``stop()``.  It reads ``tune_delay`` / ``average`` from the block on every
retune, so the setters reach it without forwarding.  ``start_sharded`` runs the rank-local form
of the same loop (one process per GPU): this rank retunes and captures only its own segments,
the all-gather of sweep i overlaps the kernels of sweep i + 1 (``sweep.SweepPipeline``).
"""
import math
import struct
import threading
import time
import weakref

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


def _stitch_loop(ref, wake, start_delay, one_sweep, collective):
    """spectrum_stitcher.run (:207-231): the start-up wait (:208), then sweep after sweep while keep_running.
    The thread holds the block only weakly, so a block that is dropped without stop() takes its stitcher along;
    ``wake`` ends both the start-up wait and a tune-delay sleep early when stop() is called.
    ``one_sweep(blk) -> (published, more)``: whether a stitched PSD went out (only those count in ``sweeps_done``) and
    whether to go on.  A ``collective`` loop (start_sharded) never leaves on its own rank's flag or on its own rank's
    failure - the other ranks would wait for it in the next collective until the backend's timeout - but only when
    one_sweep() reports what the ranks agreed on; a failure inside a collective sweep is carried through that sweep's
    gather and agreement by one_sweep itself and raised afterwards."""
    if start_delay > 0 and wake.wait(start_delay) and not collective:
        return
    while True:
        blk = ref()
        if blk is None or not (blk.keep_running or collective):
            return
        try:
            published, more = one_sweep(blk)
            if published:
                blk.sweeps_done += 1
            if collective and not more:
                blk.keep_running = False
                return
        except Exception as e:                       # a dead stitcher must not go unnoticed: work() re-raises it
            blk._stitch_error = e
            blk.keep_running = False
            return
        del blk


class spectrum_sweeper(sync_block):
    def __init__(self, rf_receiver, receiver_type, fft_len, sample_rate, trunc_sample_rate, fstart, ffinish,
                 rate, average, t_obs, tune_delay, max_tu, ctx=None, threaded=True, start_delay=2.0):
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
        self._threads_init()
        if threaded:
            self.start(start_delay)

    # -- threads (:99-105) --------------------------------------------------------
    def _threads_init(self):
        self.keep_running = True
        self.sweeps_done = 0                # stitched sweeps that went out (collective mode: that were complete on every rank)
        self.incomplete_sweeps = 0          # collective mode: sweeps in which some rank's rows failed (never published)
        self.captures = 0                 # vectors work() has stored (what data_colector would have set)
        self._stitch_error = None
        self._stitch_thread = None
        self._wake = threading.Event()

    def _launch(self, start_delay, one_sweep, collective=False):
        if self._stitch_thread is not None and self._stitch_thread.is_alive():      # (a stitcher that ended by itself - a
            raise RuntimeError('spectrum_sweeper: the stitcher is already running')  # finished collective loop, a stored error - is no obstacle)
        self.keep_running = True
        self._wake.clear()
        self._stitch_thread = threading.Thread(target=_stitch_loop, daemon=True,
                                               args=(weakref.ref(self), self._wake, start_delay, one_sweep, collective))
        self._stitch_thread.start()
        return True

    def start(self, start_delay=2.0):
        """Start the stitcher thread (the constructor does, as spectrum_sweeper.py:103-105; the 2 s are :208)."""
        return self._launch(start_delay, lambda blk: (blk.sweep_once(sleep=blk._sleep) is not None, True))

    def start_sharded(self, capture, rank, world, device, group=None, start_delay=0.0, publish_rank=None,
                      sweeps=None):
        """The stitcher loop of one rank of a sharded sweeper (one process per GPU): sweep after
        sweep through ``sweep.SweepPipeline`` - this rank retunes ITS receiver to its own segments only
        (``capture(i, f)`` returns what it then observed), the gather of a sweep overlaps the next sweep's kernels,
        and the stitched PSD of the sweep before is blended and sent while those run.  ``publish_rank`` = the rank
        whose 'pdus' port carries the frames (None: every rank's, each process has its own flowgraph).
        Ending: a collective that one rank has left never completes, so the ranks agree once per sweep (a one-word
        MIN all-reduce of keep_running, world > 1 only) and all leave after the same sweep - ``stop()`` on ANY rank,
        or ``sweeps`` reached, ends the loop everywhere; the last gathered sweep is still published."""
        state = {'n': 0}

        def one_sweep(blk):
            if 'pipe' not in state:
                state['pipe'] = blk._sharded_pipeline(rank, world, device, group)
            pipe = state['pipe']
            # A failure on THIS rank (capture, retune, HIP) must not keep it out of the sweep's collectives: its rows go
            # out as NaN, it votes to stop, every rank leaves after this sweep, and the error is raised here afterwards
            # (-> _stitch_error -> the next work()).  Whether every rank's rows were valid travels with the vote: a sweep
            # with a failed row is published nowhere.
            failure = []
            slot = blk._sharded_sweep(pipe, capture, device, failure)
            state['n'] += 1
            mine = not failure and (sweeps is None or state['n'] < sweeps)
            more, complete = blk._agree_to_continue(mine, world, device, group, ok=not failure)
            publish = publish_rank is None or publish_rank == rank
            prev, prev_ok = state.get('slot'), state.get('ok', False)
            state['slot'], state['ok'] = slot, complete
            if prev is not None and prev_ok and publish:      # the sweep before: its gather overlapped this sweep's kernels
                blk._blend_and_send(pipe.wideband(prev).cpu().numpy().astype(np.float64))
            if not complete:
                blk.incomplete_sweeps += 1
            if not more:
                if complete and publish:
                    blk._blend_and_send(pipe.wideband(slot).cpu().numpy().astype(np.float64))
                pipe.drain()
            if failure:
                raise failure[0]
            return complete, more
        return self._launch(start_delay, one_sweep, collective=True)

    def _agree_to_continue(self, mine, world, device, group=None, ok=True):
        """One two-word MIN all-reduce per sweep: (go on?, were this sweep's rows valid on every rank?)."""
        mine = bool(mine and self.keep_running)
        if world == 1:
            return mine, bool(ok)
        import torch
        import torch.distributed as dist
        flag = torch.tensor([1.0 if mine else 0.0, 1.0 if ok else 0.0], dtype=torch.float32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        flag = flag.cpu()
        return bool(flag[0] > 0.5), bool(flag[1] > 0.5)

    def stop(self):
        self.keep_running = False
        self._wake.set()
        t = self._stitch_thread
        if t is threading.current_thread():      # called from the stitcher itself (a handler it runs): the loop sees keep_running and
            return True                          # ends; the handle stays until then, so that start() cannot put a second one beside it
        if t is not None:
            t.join(5.0)
            if t.is_alive():      # still inside a capture / a collective: keep the handle, start() refuses a second stitcher
                return False
        self._stitch_thread = None
        return True

    def _sleep(self, seconds):
        # time.sleep(self.tune_delay) of :219, cut short by stop()
        if seconds > 0:
            self._wake.wait(seconds)

    def _check_stitcher(self):
        if getattr(self, '_stitch_error', None) is not None:
            e, self._stitch_error = self._stitch_error, None
            raise RuntimeError('spectrum_sweeper: the stitcher thread died: %r' % (e,)) from e

    # -- flowgraph side -----------------------------------------------------------
    def work(self, input_items, output_items):
        self._check_stitcher()
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
        self.captures = getattr(self, 'captures', 0) + 1

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

    def _tune(self, f):
        try:
            self.rf_receiver.set_center_freq(f, 0)
        except Exception:
            print('cant tune receiver')

    def sweep_once(self, sleep=time.sleep):
        """One iteration of spectrum_stitcher.run (:211-231)."""
        psd = np.array([])
        for f in self.tune_frequencies:
            self._tune(f)
            sleep(self.tune_delay)
            psd = np.concatenate((psd, self._src_power(self.get_samples())), axis=0)
            if not getattr(self, 'keep_running', True):      # stop() during a sweep: nothing half-stitched goes out
                return None
        return self._blend_and_send(psd)

    def sweep_once_sharded(self, capture, rank, world, device, group=None):
        """The same sweep with segment i on rank i mod world and one all-gather (RCCL over xGMI on GPUs).
        ``capture(i, f)`` returns what was observed at tune frequency f: host complex64 samples, or a torch
        complex64 / float32 [n][2] tensor already on ``device``.  Each segment goes through the HIP plan straight
        into this rank's row of the gather buffer (device in, device out); only the stitched wideband PSD
        comes back to the host, for the PDU fragments."""
        import torch
        from . import sweep
        device = self._cuda_device(device)
        nbins = self.fft_len - 2 * self.excess_bins
        wide = sweep.sweep_psd(lambda i: capture(i, self.tune_frequencies[i]),
                               lambda iq, out_row: self._segment_to_row(iq, out_row, device),
                               len(self.tune_frequencies), nbins, device, rank, world, group)
        return self._blend_and_send(wide.cpu().numpy().astype(np.float64))

    @staticmethod
    def _cuda_device(device):
        import torch
        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError('the sharded sweep computes on the GPU: pass the rank\'s cuda device')
        return device

    def _segment_to_row(self, iq, out_row, device):
        """One captured segment -> its dB row of the gather buffer (device in, device out)."""
        import torch
        from . import sweep
        device = self._cuda_device(device)
        if not torch.is_tensor(iq):
            iq = torch.from_numpy(np.ascontiguousarray(iq, np.complex64).view(np.float32)).to(device)
        nsamples = iq.numel() // 2 if iq.dtype == torch.float32 else iq.numel()
        # a context on its own stream: torch's copy / the caller's producer kernels / the zero fill of the row
        # buffer must have landed before the plan reads and writes them, and the plan must be done before the
        # all-gather (torch's stream) reads the row
        sweep.torch_then_ctx(self.ctx, device)
        self._plan.exec_dev(iq.data_ptr(), nsamples, out_row.data_ptr())
        sweep.ctx_then_torch(self.ctx)

    def _sharded_pipeline(self, rank, world, device, group=None, depth=2):
        from . import sweep
        return sweep.SweepPipeline(len(self.tune_frequencies), self.fft_len - 2 * self.excess_bins, device, rank,
                                   world, group, depth)

    def _sharded_sweep(self, pipe, capture, device, failure=None):
        """This rank's share of one sweep: retune to each of its segments, wait tune_delay, take the capture, run
        the plan into the pipeline's row; then start the gather.  -> the pipeline slot to read the sweep from.
        ``failure`` (a list): an exception in a segment is appended there instead of leaving the sweep - that row and
        the rank's remaining rows of this sweep are filled with NaN, so that the gather still takes place."""
        def compute(i, out_row):
            if failure:
                out_row.fill_(float('nan'))
                return
            try:
                f = self.tune_frequencies[i]
                self._tune(f)
                self._sleep(self.tune_delay)
                self._segment_to_row(capture(i, f), out_row, device)
            except Exception as e:
                if failure is None:
                    raise
                failure.append(e)
                out_row.fill_(float('nan'))
        return pipe.run(compute)
