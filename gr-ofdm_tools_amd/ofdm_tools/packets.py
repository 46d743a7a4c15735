"""PDU fragment wire format of the PSD producers.

frame = [n_frags:u8][frag_id:u8][payload <= max_tu]; payload is the float32-LE dB
vector, or its ``astype(int8)`` when ``data_precision`` is false.
Producers: local_worker.packet_source.send_packet (python/local_worker.py:147-172)
and spectrum_sweeper.packet_source.send_packet (python/spectrum_sweeper.py:240-258);
consumers: remote_client_qt.handler (python/remote_client_qt.py:100-164),
sdr_webserver data_processor.run (sdr_webserver/sdr_webserver_ws.py:235-287).
"""
import math
import struct

import numpy as np


def _frames(data, fragments, max_tu):
    frames = []
    j = 0
    for i in range(fragments):
        frag = data[j:j + max_tu]
        if i == fragments - 1:
            frag = data[j:]
        frames.append(struct.pack('!B', fragments) + struct.pack('!B', i) + frag)
        j += max_tu
    return frames


def worker_fragments(fft_data, max_tu, fft_len, data_precision):
    """local_worker.py:147-172: ceil(N*4/max_tu) float32 frames or ceil(N/max_tu) int8 frames."""
    fft_data = np.asarray(fft_data, np.float32)
    if data_precision:
        fragments = int(math.ceil(fft_len * 4 / float(max_tu)))
    else:
        fft_data = fft_data.astype(np.int8)
        fragments = int(math.ceil(fft_len / float(max_tu)))
    return _frames(fft_data.tobytes(), fragments, max_tu)


def sweeper_fragments(data, max_tu):
    """spectrum_sweeper.py:240-258.  The reference computes ``int(ceil(len/max_tu)) + 1`` with
    Python-2 integer division, i.e. floor + 1; the quirk is kept so that consumers that
    trust the n_frags byte keep working."""
    fragments = int(math.ceil(len(data) // max_tu)) + 1
    return _frames(data, fragments, max_tu)


def sweeper_fragment_count(nbytes, max_tu):
    """Frames per sweep the sweeper emits for an nbytes payload (floor + 1, spectrum_sweeper.py:242)."""
    return nbytes // max_tu + 1


def reassemble(frames, data_precision=None, header=0):
    """All frames of ONE vector, in any arrival order -> its payload: order by frag_id, strip the 2-byte
    fragment header (and ``header`` transport bytes in front of it).  ``data_precision`` None returns the bytes;
    True / False decodes them as the producers packed them - float32-LE dB, or the ``astype(int8)`` bytes of
    local_worker.py:155-156 (remote_client_qt.py:73-76 picks the dtype the same way)."""
    frames = [bytes(f)[header:] for f in frames]
    n = frames[0][0]
    got = {f[1]: f[2:] for f in frames if f[0] == n}
    if len(got) != n:
        raise ValueError('missing fragments: have %d of %d' % (len(got), n))
    data = b''.join(got[i] for i in range(n))
    if data_precision is None:
        return data
    return np.frombuffer(data, '<f4' if data_precision else np.int8)


ZMQ_PDU_HEADER_LEN = 10


def zmq_pdu_header(nbytes):
    """The 10 bytes a GNU Radio ZMQ PUB message sink puts in front of a PDU body: the PMT serialisation of
    cons(PMT_NIL, u8vector) up to the vector's first byte - pair tag 0x07, null tag 0x06, uniform-vector tag 0x0a,
    element type u8 0x00, the big-endian uint32 item count, one count byte of padding (1) and the pad byte.
    sdr_webserver data_processor.run drops exactly these (sdr_webserver/sdr_webserver_ws.py:241)."""
    return bytes([0x07, 0x06, 0x0a, 0x00]) + struct.pack('>I', nbytes) + bytes([0x01, 0x00])


class FragmentReassembler(object):
    """The consumer side of the wire format as a stream: frames go in one at a time, a decoded vector comes out
    when the frame carrying the LAST frag_id arrives.

    Follows remote_client_qt.handler (python/remote_client_qt.py:100-164) and sdr_webserver's data_processor.run
    (sdr_webserver/sdr_webserver_ws.py:235-287, ``header=10``: it strips the ZMQ/PMT header first):
      * a frame with n_frags == 1 is decoded on its own and does not touch the pending payload;
      * otherwise payloads are appended in ARRIVAL order (frag_id only marks the end, :140-141), so a lost middle
        fragment shortens the payload (at the usual max_tu of 1470 + 2, not a multiple of four, float32 then ends in
        the error branch below) and a lost final fragment glues two vectors together - the consumers have that
        weakness (tests/golden/ref_consumers.npz holds what the reference's own two consumers made of such a stream)
        and a drop-in keeps the format, so ``strict=True`` is offered for hosts that would rather drop such a vector
        (frag_ids must run 0 .. n-1) than plot it;
      * a payload whose length is not a multiple of the item size is discarded (np.fromstring raises there and the
        consumers print an error, :132-133,161-162); the web consumer then clears the pending payload (:279), the Qt
        one keeps it - ``clear_on_error`` says which;
      * peak hold: ``max_data`` restarts from the vector whenever its length changes, else element-wise maximum
        (:119-128)."""

    def __init__(self, data_precision=True, header=0, strict=False, clear_on_error=True):
        self.dtype = np.dtype('<f4') if data_precision else np.dtype(np.int8)
        self.header = int(header)
        self.strict = bool(strict)
        self.clear_on_error = bool(clear_on_error)
        self.pending = b''
        self._next_id = 0
        self.max_data = None
        self.vectors = self.errors = 0

    def set_data_precision(self, data_precision):
        self.dtype = np.dtype('<f4') if data_precision else np.dtype(np.int8)

    def _decode(self, payload):
        if len(payload) % self.dtype.itemsize:
            self.errors += 1
            return None
        v = np.frombuffer(payload, self.dtype)
        if self.max_data is None or len(self.max_data) != len(v):
            self.max_data = v
        self.max_data = np.maximum(self.max_data, v)
        self.vectors += 1
        return v

    def push(self, frame):
        """-> the decoded vector this frame completes, or None."""
        frame = bytes(frame)[self.header:]
        n_frags, frag_id, payload = frame[0], frame[1], frame[2:]
        if n_frags == 1:
            return self._decode(payload)
        if self.strict and frag_id != self._next_id:
            self.pending, self._next_id = (payload, 1) if frag_id == 0 else (b'', 0)
            self.errors += 1
            return None
        self.pending += payload
        self._next_id = frag_id + 1
        if frag_id != n_frags - 1:
            return None
        v = self._decode(self.pending)
        if v is not None or self.clear_on_error:
            self.pending = b''
        self._next_id = 0
        return v
