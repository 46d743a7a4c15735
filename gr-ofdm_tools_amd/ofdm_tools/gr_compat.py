"""The slice of the GNU Radio Python block API that the sensing blocks rely on.

If GNU Radio is importable the blocks derive from ``gnuradio.gr.sync_block`` and
messages are PMTs; otherwise (this image has no GNU Radio) a local base class
with the same ``work(input_items, output_items) -> int`` contract
(python/spectrum_sensor.py:37-40,71-75 in the reference) and the same message-port
method names is used, so the blocks can be driven by any host loop or test.

Contract kept from gr.sync_block:
  * ``input_items[0]`` is a 1-D complex64 view owned by the caller and valid only
    during the call; ``work`` must not keep a reference (the HIP chain copies it
    to the device before returning);
  * the return value is the number of items consumed;
  * ``work`` never blocks on a consumer: results go to a lossy depth-2 queue,
    latest wins, exactly like ``message_sink(..., dont_block=True)`` +
    ``gr.msg_queue(2)`` in the reference (spectrum_sensor_v2.py:71-72,97).
"""
import collections
import threading

import numpy as np

try:                                            # GNU Radio is not in this image: tests/gr_standin.py stands in for it
    from gnuradio import gr as _gr
    import pmt as _pmt
    HAVE_GNURADIO = True
except Exception:                               # ImportError or a broken install
    _gr = None
    _pmt = None
    HAVE_GNURADIO = False


class LossyQueue(object):
    """gr.msg_queue(limit) fed by a message_sink with dont_block=True: when full, the new
    message is dropped and the producer carries on."""

    def __init__(self, limit=2):
        self.limit = limit
        self._q = collections.deque()
        self._cv = threading.Condition()
        self.dropped = 0

    def insert_tail(self, msg):
        with self._cv:
            if len(self._q) >= self.limit:
                self.dropped += 1
                return False
            self._q.append(msg)
            self._cv.notify()
            return True

    def delete_head(self, timeout=None):
        with self._cv:
            if not self._q:
                self._cv.wait(timeout)
            return self._q.popleft() if self._q else None

    def delete_head_nowait(self):
        with self._cv:
            return self._q.popleft() if self._q else None

    def count(self):
        with self._cv:
            return len(self._q)


def _plain(value):
    """Python scalars / lists for pmt.to_pmt: it converts None, bool, str, int, float, complex, dict, list, tuple and
    numpy ARRAYS (gnuradio/pmt/pmt_to_python.py) but not numpy scalars - and the sensing results are numpy scalars
    (np.float32 thresholds, lists of np.float64 channel frequencies)."""
    if isinstance(value, np.generic):
        return value.item()
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    return value


def to_msg(key, value):
    """(key, value) pair: pmt.cons(to_pmt(key), to_pmt(value)) under GNU Radio (python/spectrum_sensor.py:122-128), a
    tuple otherwise."""
    if HAVE_GNURADIO:
        return _pmt.cons(_pmt.to_pmt(key), _pmt.to_pmt(_plain(value)))
    return (key, value)


def pdu(payload):
    """PDU with nil metadata and a u8vector body (python/local_worker.py:168-171)."""
    if HAVE_GNURADIO:
        return _pmt.cons(_pmt.PMT_NIL, _pmt.init_u8vector(len(payload), list(bytearray(payload))))
    return (None, bytes(payload))


def pdu_parts(msg):
    """(metadata dict, body) of a PDU, or None for a message that is not a pair - "Message is not a valid PDU", which
    the reference answers with nothing (python/spectrum_sensor.py:77-86).  Under GNU Radio the body stays a PMT: the
    legacy sensor compares ``str(body)`` with its request names."""
    if HAVE_GNURADIO:
        try:
            meta, body = _pmt.car(msg), _pmt.cdr(msg)
        except Exception:
            return None
        meta = _pmt.to_python(meta)
        return (meta if isinstance(meta, dict) else {}), body
    if isinstance(msg, tuple) and len(msg) == 2:
        return (msg[0] if isinstance(msg[0], dict) else {}), msg[1]
    return None


class _LocalSyncBlock(object):
    """Stand-in for gr.sync_block when GNU Radio is absent."""

    def __init__(self, name, in_sig, out_sig):
        self._name = name
        self.in_sig = in_sig
        self.out_sig = out_sig
        self._out_ports = {}
        self._handlers = {}

    def name(self):
        return self._name

    # message ports ---------------------------------------------------------
    def message_port_register_out(self, port):
        self._out_ports.setdefault(str(port), [])

    message_port_register_hier_out = message_port_register_out

    def message_port_register_in(self, port):
        self._handlers.setdefault(str(port), None)

    def set_msg_handler(self, port, fn):
        self._handlers[str(port)] = fn

    def msg_connect(self, port, fn):
        """Subscribe a callable to an output port (host-loop equivalent of tb.msg_connect)."""
        self._out_ports.setdefault(str(port), []).append(fn)

    def message_port_pub(self, port, msg):
        for fn in self._out_ports.get(str(port), []):
            fn(msg)

    def post(self, port, msg):
        """Deliver a message to an input port (what the scheduler does for a connected port)."""
        fn = self._handlers.get(str(port))
        if fn is None:
            raise KeyError('no handler registered for port %r' % (port,))
        fn(msg)

    def work(self, input_items, output_items):   # pragma: no cover - abstract
        raise NotImplementedError

    # host loop ---------------------------------------------------------------
    def feed(self, samples, max_items=8191):
        """Drive ``work()`` the way the scheduler would: repeated calls with whatever
        is available, honouring the consumed count."""
        samples = np.asarray(samples)
        pos = 0
        while pos < len(samples):
            n = self.work([samples[pos:pos + max_items]], [])
            if n <= 0:
                break
            pos += n
        return pos


if HAVE_GNURADIO:
    class sync_block(_gr.sync_block):
        def __init__(self, name, in_sig, out_sig):
            _gr.sync_block.__init__(self, name=name, in_sig=in_sig, out_sig=out_sig)
            self._subs = {}

        def message_port_register_out(self, port):
            _gr.sync_block.message_port_register_out(self, _pmt.intern(str(port)))

        message_port_register_hier_out = message_port_register_out

        def message_port_register_in(self, port):
            _gr.sync_block.message_port_register_in(self, _pmt.intern(str(port)))

        def set_msg_handler(self, port, fn):
            _gr.sync_block.set_msg_handler(self, _pmt.intern(str(port)), fn)

        def msg_connect(self, port, fn):
            self._subs.setdefault(str(port), []).append(fn)

        def message_port_pub(self, port, msg):
            _gr.sync_block.message_port_pub(self, _pmt.intern(str(port)), msg)
            for fn in self._subs.get(str(port), []):
                fn(msg)

        def feed(self, samples, max_items=8191):
            return _LocalSyncBlock.feed(self, samples, max_items)
else:
    sync_block = _LocalSyncBlock
