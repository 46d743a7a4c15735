"""Shared work() / watcher plumbing of the blocks that run the periodogram chain.

The reference's hier blocks end in ``message_sink(..., msgq, dont_block=True)`` with ``gr.msg_queue(2)``
(python/spectrum_sensor_v2.py:71-72,97; psd_logger.py:55-56; local_worker.py:71; multichannel_scanner.py:91)
and a daemon watcher thread that pops a message, keeps only its LAST vector and processes it
(spectrum_sensor_v2.py:404-414).  The stream side never waits for the watcher: when the depth-2 queue is
full the new message is dropped.

Here ``work()`` (the gr.sync_block contract of python/spectrum_sensor.py:71-75: input valid only during the
call, must not block) hands the samples to ``oth_chain_push_async`` - pinned copy, H2D + kernels + D2H of the
latest row enqueued on the context's stream, event recorded - and gets a ticket back.  With
``threaded=True`` the ticket goes into a lossy depth-2 queue (dropped when the watcher lags, the stream side
carries on) and the watcher thread waits for the ticket's event outside the context lock, then runs
``_on_vector(row)``.  With ``threaded=False`` (deterministic hosts and tests) the ticket is collected
right away on the caller's thread - same code path, no second thread.
"""
import threading

from . import _hip
from .gr_compat import LossyQueue


class ChainBlockMixin(object):
    def _chain_init(self, chain, threaded=False):
        self._chain = chain
        self.msgq0 = LossyQueue(2)
        self._threaded = bool(threaded)
        self.keep_running = True
        self.vectors_lost = 0            # tickets the ring had already recycled when the watcher got to them
        self._watch_thread = None
        if self._threaded:
            self._watch_thread = threading.Thread(target=self._watch, daemon=True)
            self._watch_thread.start()

    # -- gr.sync_block ----------------------------------------------------------------------------
    def work(self, input_items, output_items):
        in0 = input_items[0]
        ticket = self._chain.push_async(in0)          # returns after enqueue; the GPU works behind it
        self.last_ticket = ticket
        if self._threaded:
            self.msgq0.insert_tail(ticket)            # dont_block: dropped when two are already waiting
        else:
            self._collect(ticket)
        return len(in0)

    def _collect(self, ticket):
        try:
            row, n = self._chain.wait(ticket)
        except _hip.HipError as e:
            if e.code != -5:                          # OTH_ERR_STATE: newer vectors recycled the slot
                raise
            self.vectors_lost += 1
            return
        if n:
            self._on_vector(row)

    def _watch(self):
        while self.keep_running:
            ticket = self.msgq0.delete_head(timeout=0.05)
            if ticket is not None:
                self._collect(ticket)

    def stop(self):
        self.keep_running = False
        if self._watch_thread is not None:
            self._watch_thread.join(1.0)
            self._watch_thread = None
        return True

    def _on_vector(self, row):          # pragma: no cover - abstract
        raise NotImplementedError


class MessageStrobe(object):
    """blocks.message_strobe(msg, period_ms): re-emits its current message every period
    (python/spectrum_sensor_v2.py:108-111,125-129).  ``publish(msg)`` is the downstream port."""

    def __init__(self, publish, msg=None, period_ms=1000):
        self._publish = publish
        self._msg = msg
        self.period = period_ms * 1e-3
        self._stop = threading.Event()
        self._thread = None
        self.emitted = 0

    def set_msg(self, msg):
        self._msg = msg

    def msg(self):
        return self._msg

    def start(self):
        if self._thread is None:
            self._stop.clear()
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def _run(self):
        while not self._stop.wait(self.period):
            if self._msg is not None:
                self._publish(self._msg)
                self.emitted += 1

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(1.0)
            self._thread = None

    @property
    def running(self):
        return self._thread is not None
