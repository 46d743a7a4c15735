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


def reassemble(frames):
    """What remote_client_qt.handler does: order by frag_id, strip the 2-byte header."""
    n = frames[0][0]
    got = {f[1]: f[2:] for f in frames if f[0] == n}
    if len(got) != n:
        raise ValueError('missing fragments: have %d of %d' % (len(got), n))
    return b''.join(got[i] for i in range(n))
