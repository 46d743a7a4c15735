"""Shared work() / watcher plumbing of the blocks that run the periodogram chain.

The reference's hier blocks end in ``message_sink(..., msgq, dont_block=True)`` with ``gr.msg_queue(2)``
(python/spectrum_sensor_v2.py:71-72,97; psd_logger.py:55-56; local_worker.py:71; multichannel_scanner.py:91)
and a daemon watcher thread that pops a message, keeps only its LAST vector and processes it
(spectrum_sensor_v2.py:404-414).  The stream side never waits for the watcher: when the depth-2 queue is
full the new message is dropped.

Here ``work()`` (the gr.sync_block contract of python/spectrum_sensor.py:71-75: input valid only during the
call, must not block) hands the samples to ``oth_chain_push_async`` - pinned copy, H2D + kernels + D2H of the
latest row enqueued on the context's stream, event recorded - and gets a ticket back.  With
``threaded=True`` - the default of every block, as the reference always runs its watcher threads
(spectrum_sensor_v2.py:138-155) - the ticket goes into a lossy depth-2 queue (dropped when the watcher lags, the
stream side carries on) and the watcher thread waits for the ticket's event outside the context lock, then runs
``_on_vector(row)``: ``work()`` never waits for the GPU.  ``threaded=False`` (deterministic hosts and the
tests, which pass it explicitly) collects the ticket right away on the caller's thread - same code path, no
second thread.  ``drain()`` lets a host that stops feeding wait until the watcher has caught up.
"""
import threading
import weakref

from . import _hip
from .gr_compat import LossyQueue


def _watch_loop(ref, queue):
    """Watcher thread body (the reference's _queue_watcher.run loops, e.g. spectrum_sensor_v2.py:395-414).
    Counters: ``_queued`` is written only by the thread that calls work(), ``_done`` only by this thread - one writer
    each, so the plain ``+=`` needs no lock; drain() only reads them.  An item taken off the queue always counts as
    done, whether it was processed, discarded because the block has stopped, or failed: drain() must never wait for a
    vector nobody will look at.  A failure in ``_collect`` / ``_on_vector`` (a logger's disk, a scanner's state, a HIP
    error other than "lost against newer vectors") is kept in ``_watch_error`` and raised by the next work() or
    drain() on the stream side - with threaded=False the same error would have come straight out of work(); the
    watcher itself carries on with the next vector, as the reference's loops do after their prints."""
    while True:
        item = queue.delete_head(timeout=0.05)
        blk = ref()
        if blk is None:
            return
        if item is not None:
            try:
                if blk.keep_running:
                    blk._collect(item)
            except Exception as e:
                blk._watch_error = e
                blk.watch_errors += 1
            finally:
                blk._done += 1
        if not blk.keep_running:
            return
        del blk


class ChainBlockMixin(object):
    def _chain_init(self, chain, threaded=True):
        self._chain = chain
        self.msgq0 = LossyQueue(2)
        self._threaded = bool(threaded)
        self.keep_running = True
        self.vectors_lost = 0            # tickets the ring had already recycled when the watcher got to them
        self._queued = self._done = 0    # tickets handed to / finished by the watcher thread (drain())
        self._watch_error = None         # last exception of the watcher thread, re-raised on the stream side
        self.watch_errors = 0
        self.rows_total = 0              # PSD vectors the chain has produced so far (all work() calls)
        self.vector_rows_end = self.vector_nrows = 0
        self._watch_thread = None
        if self._threaded:
            # the thread holds the block only weakly: a block that is dropped without stop() takes its watcher with it
            self._watch_thread = threading.Thread(target=_watch_loop, args=(weakref.ref(self), self.msgq0), daemon=True)
            self._watch_thread.start()

    # -- gr.sync_block ----------------------------------------------------------------------------
    def _raise_watch_error(self):
        if self._watch_error is not None:
            e, self._watch_error = self._watch_error, None
            raise RuntimeError('%s: the watcher thread failed on a vector: %r' % (type(self).__name__, e)) from e

    def work(self, input_items, output_items):
        self._raise_watch_error()
        in0 = input_items[0]
        ticket = self._chain.push_async(in0)          # returns after enqueue; the GPU works behind it
        self.last_ticket = ticket
        nrows = self._chain.ticket_rows(ticket)       # known at enqueue time: the stream's vector count goes on even
        self.rows_total += nrows                      # when the watcher drops this ticket
        item = (ticket, self.rows_total, nrows)
        if self._threaded:
            if self.msgq0.insert_tail(item):          # dont_block: dropped when two are already waiting
                self._queued += 1
        else:
            self._collect(item)
        return len(in0)

    def _collect(self, item):
        ticket, self.vector_rows_end, self.vector_nrows = item      # the vector handed to _on_vector is row rows_end - 1
        try:
            row, n = self._chain.wait(ticket)
        except _hip.HipError as e:
            if e.code != -5:                          # OTH_ERR_STATE: newer vectors recycled the slot
                raise
            self.vectors_lost += 1
            return
        if n:
            self._on_vector(row)

    def drain(self, timeout=5.0):
        """Wait until the watcher thread has consumed what work() queued (hosts that stop feeding and want the last
        vector's effects; never called from work()).  -> True when idle."""
        import time
        end = time.monotonic() + timeout
        while self._threaded and self.keep_running and self._watch_thread is not None and time.monotonic() < end:
            if self._done >= self._queued:
                break
            time.sleep(0.001)
        self._raise_watch_error()
        # a stopped block has no watcher: whatever is still queued will never be looked at, nothing is pending
        return not self._threaded or not self.keep_running or self._done >= self._queued

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
