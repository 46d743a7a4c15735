"""CPU tests of the host logic around the HIP path: wire format, decision stage, block API
surface, and the multi-rank sweep sharding over gloo (world_size 2, and 8 for BASELINE config 4's own shape)."""
import inspect
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_cpu as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packet_framing_matches_golden_and_oracle():
    from ofdm_tools import packets
    raw = open(os.path.join(ROOT, 'tests', 'golden', 'fragments.bin'), 'rb').read()
    pos, groups = 0, []
    for _ in range(3):
        n = struct.unpack_from('<I', raw, pos)[0]
        pos += 4
        frames = []
        for _ in range(n):
            ln = struct.unpack_from('<I', raw, pos)[0]
            pos += 4
            frames.append(raw[pos:pos + ln])
            pos += ln
        groups.append(frames)
    db = (np.arange(4096, dtype=np.float32) * 0.01 - 90).astype('<f4')
    assert packets.worker_fragments(db, 1470, 4096, True) == groups[0]
    assert packets.worker_fragments(db, 1470, 4096, False) == groups[1]
    assert packets.sweeper_fragments(db.tobytes(), 1470) == groups[2]
    assert packets.reassemble(groups[0]) == db.tobytes()
    assert packets.reassemble(list(reversed(groups[2]))) == db.tobytes()       # any arrival order
    with pytest.raises(ValueError):
        packets.reassemble(groups[0][:-1])
    # the sweeper's floor+1 rule emits an empty last frame when the length divides exactly
    fr = packets.sweeper_fragments(b'x' * 2940, 1470)
    assert len(fr) == 3 and fr[2][2:] == b'' and fr == R.sweeper_fragments(b'x' * 2940, 1470)


def test_packers_match_the_reference_packers_around_fragment_boundaries():
    """f1 against the reference's OWN packet_source.send_packet bodies (tests/golden/ref_threads.npz, made by
    make_golden.reference_thread_fixtures): local_worker.py:147-172 and spectrum_sweeper.py:240-258 at payload lengths on
    and around multiples of max_tu - the worker's ceil() and the sweeper's floor() + 1 with its empty closing frame."""
    from ofdm_tools import packets
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_threads.npz'))
    assert str(g['source']) == 'reference'

    def frames_of(blob):
        raw, pos, frames = bytes(blob), 0, []
        while pos < len(raw):
            ln = struct.unpack_from('<I', raw, pos)[0]
            frames.append(raw[pos + 4:pos + 4 + ln])
            pos += 4 + ln
        return frames
    for n in g['frame_lengths']:
        v = (np.arange(n, dtype=np.float32) * 0.5 - 70).astype('<f4')
        want_w, want_s = frames_of(g['worker_frames_%d' % n]), frames_of(g['sweeper_frames_%d' % n])
        assert packets.worker_fragments(v, 1472, int(n), True) == want_w, n
        assert packets.sweeper_fragments(v.tobytes(), 1472) == want_s, n
        assert packets.sweeper_fragment_count(4 * int(n), 1472) == len(want_s)
        assert packets.reassemble(want_w) == v.tobytes() and packets.reassemble(want_s) == v.tobytes()
    raw = open(os.path.join(ROOT, 'tests', 'golden', 'fragments.bin'), 'rb').read()
    assert raw == bytes(g['fragments_bin'])          # the byte fixture used above IS what the reference packers emit


def read_consumer_fixture():
    raw = open(os.path.join(ROOT, 'tests', 'golden', 'fragments_consumer.bin'), 'rb').read()
    pos = [0]

    def u32():
        v = struct.unpack_from('<I', raw, pos[0])[0]
        pos[0] += 4
        return v

    def blob():
        n = u32()
        b = raw[pos[0]:pos[0] + n]
        pos[0] += n
        return b
    streams = []
    for _ in range(u32()):
        header, itemsize, nframes = u32(), u32(), u32()
        frames = [blob() for _ in range(nframes)]
        dt = '<f4' if itemsize == 4 else np.int8
        streams.append((header, itemsize == 4, frames, [np.frombuffer(blob(), dt) for _ in range(u32())]))
    assert pos[0] == len(raw)
    return streams


def test_ref_f1_consumers_against_the_reference_handlers(golden):
    """f1, receiving side, against the reference's OWN consumers (ref_consumers.npz: ``data_processor.run`` of
    sdr_webserver_ws.py:235-287 and ``remote_client_qt.handler`` of remote_client_qt.py:100-164, run frame by frame):
    the product reassembler must complete a vector at the same frames with the same bytes, hold the same peak, count
    the same errors - also where fragments are lost.  What the lossy stream shows of the format: max_tu = 1470 is
    not a multiple of four, so a lost middle fragment leaves a ragged payload (the error branch, not a shorter
    vector); the web consumer then drops what is pending, the Qt client keeps it and glues the next vectors to it."""
    from ofdm_tools import packets
    g = golden('ref_consumers.npz')

    def frames_of(tag):
        raw, out, pos = bytes(g['frames_' + tag]), [], 0
        for n in g['frames_%s_len' % tag]:
            out.append(raw[pos:pos + int(n)])
            pos += int(n)
        return out
    hdr = lambda fr: R.zmq_pdu_header(len(fr)) + fr      # noqa: E731
    for tag in ('f32', 'sweeper', 'lossy'):               # the web server: header stripped, pending dropped on error
        ra = packets.FragmentReassembler(True, header=10, clear_on_error=True)
        got = [(k, v) for k, v in ((k, ra.push(hdr(fr))) for k, fr in enumerate(frames_of(tag))) if v is not None]
        assert [k for k, _ in got] == list(g['web_%s_at' % tag]), tag
        for j, (_, v) in enumerate(got):
            assert v.tobytes() == bytes(g['web_%s_out_%d' % (tag, j)]), (tag, j)
        assert ra.errors == int(g['web_%s_errors' % tag])
    for tag, precision in (('f32', True), ('i8', False), ('lossy', True)):      # the Qt client: bare frames, pending kept
        ra = packets.FragmentReassembler(precision, header=0, clear_on_error=False)
        frames, j = frames_of(tag), 0
        for k, fr in enumerate(frames):
            v = ra.push(fr)
            if v is None:
                continue
            assert k == int(g['qt_%s_at' % tag][j]), (tag, k)
            data, peak = ('curve0', 'curve1') if fr[0] == 1 else ('curve1', 'curve0')      # :127-129 against :154-156
            assert v.dtype == g['qt_%s_%s_%d' % (tag, data, j)].dtype
            assert v.tobytes() == g['qt_%s_%s_%d' % (tag, data, j)].tobytes()      # (a glued vector holds NaN patterns)
            assert np.array_equal(ra.max_data, g['qt_%s_%s_%d' % (tag, peak, j)], equal_nan=True)
            assert np.array_equal(ra.max_data, g['qt_%s_peak_%d' % (tag, j)], equal_nan=True)
            j += 1
        assert j == len(g['qt_%s_at' % tag]) and ra.errors == int(g['qt_%s_errors' % tag])
        assert len(ra.pending) == int(g['qt_%s_pending' % tag])
    # the axis the client plots against (:125): sample_rate / 2 * linspace(-1, 1, n) + tune_freq, in MHz
    assert np.allclose(g['qt_f32_axis_mhz'], (2.0e6 / 2 * np.linspace(-1, 1, 256) + 100.0e6) / 1e6, rtol=0, atol=1e-9)


def test_fragment_consumers_int8_float_and_zmq_header():
    """remote_client_qt.py:100-164 / sdr_webserver_ws.py:235-287: the stream reassembler against the byte fixture
    (float32 frames behind the 10-byte ZMQ/PMT header, bare int8 frames, sweeper frames with their floor+1 count)."""
    from ofdm_tools import packets
    streams = read_consumer_fixture()
    assert [(h, p, len(f), len(v)) for h, p, f, v in streams] == [(10, True, 18, 3), (0, False, 6, 3), (10, True, 12, 2)]
    for header, precision, frames, vectors in streams:
        ra = packets.FragmentReassembler(precision, header=header)
        got = [v for v in (ra.push(f) for f in frames) if v is not None]
        assert len(got) == len(vectors) and all(np.array_equal(a, b) and a.dtype == b.dtype
                                                for a, b in zip(got, vectors))
        assert np.array_equal(ra.max_data, np.maximum.reduce(vectors)) and ra.errors == 0
        if header:
            assert all(f[:header] == packets.zmq_pdu_header(len(f) - header) for f in frames)
        # the one-vector form: any arrival order, the same decode
        n = frames[0][header]
        assert np.array_equal(packets.reassemble(list(reversed(frames[:n])), precision, header), vectors[0])
    # int8 frames are what local_worker sends with data_precision False: the float dB vector cast to int8
    row = (np.arange(2048, dtype=np.float32) * 0.02 - 95).astype('<f4')
    assert np.array_equal(streams[1][3][0], row.astype(np.int8))
    assert packets.worker_fragments(row, 1470, 2048, False) == streams[1][2][:2]
    # arrival-order concatenation is the consumers' behaviour: a lost fragment shortens an int8 vector ...
    _, _, i8frames, i8vec = streams[1]
    ra = packets.FragmentReassembler(False)
    short = ra.push(i8frames[1])                                     # frame 0 of the vector never arrived
    ref, _ = R.consumer_handler(i8frames[1:2], np.int8, 0)
    assert len(short) == 2048 - 1470 and np.array_equal(short, ref[0]) and np.array_equal(short, i8vec[0][1470:])
    # ... and makes a float vector undecodable (1470 is not a multiple of 4): an error, nothing delivered
    header, precision, frames, vectors = streams[0]
    lossy = [f for i, f in enumerate(frames) if i != 2]
    ra = packets.FragmentReassembler(True, header=10, clear_on_error=True)        # the web consumer (:279)
    out = [v for v in (ra.push(f) for f in lossy) if v is not None]
    ref, _ = R.consumer_handler(lossy, '<f4', 10, clear_on_error=True)
    assert ra.errors == 1 and len(out) == len(ref) == 2 and all(np.array_equal(a, b) for a, b in zip(out, ref))
    ra = packets.FragmentReassembler(True, header=10, clear_on_error=False)       # the Qt client keeps the bytes
    out = [v for v in (ra.push(f) for f in lossy) if v is not None]
    ref, _ = R.consumer_handler(lossy, '<f4', 10, clear_on_error=False)
    assert len(out) == len(ref) and all(np.array_equal(a, b) for a, b in zip(out, ref))
    # strict=True drops the damaged vector and resynchronises on the next frag_id 0
    ra = packets.FragmentReassembler(True, header=10, strict=True)
    out = [v for v in (ra.push(f) for f in lossy) if v is not None]
    assert len(out) == 2 and np.array_equal(out[0], vectors[1]) and np.array_equal(out[1], vectors[2])
    # a lost FINAL fragment glues two vectors (frag_id only marks the end)
    glued = i8frames[:1] + i8frames[2:4]
    ra = packets.FragmentReassembler(False)
    out = [v for v in (ra.push(f) for f in glued) if v is not None]
    ref, _ = R.consumer_handler(glued, np.int8, 0)
    assert len(out) == 1 and len(out[0]) == 1470 + 2048 and np.array_equal(out[0], ref[0])


def test_coherence_detector_decision_stage(golden):
    import ofdm_tools
    g = golden('coherence_scanner.npz')
    calls = []
    det = ofdm_tools.coherence_detector(int(g['N']), int(g['sample_rate']), threshold=10, threshold_mtm=0.2,
                                        tune_freq=int(g['tune_freq']), subject_channels=list(g['subject_channels']),
                                        valve_callback=calls.append)
    assert det.idx_subject_channels == list(g['idx'])
    stale = np.zeros_like(g['d0'])
    # two vectors in one call: only the last one is scanned (last-vector rule)
    n = det.work([np.stack([stale, g['d0']]), np.stack([stale, g['d1']]), np.stack([stale, g['d2']])], [])
    assert n == 2
    assert det.get_subject_channels_outcome() == list(g['outcome'])
    assert calls == list(g['valve'])
    assert np.allclose(det.subject_channels_coherence, g['coherence'])


def test_blocks_expose_every_public_method_of_the_reference_blocks():
    """The methods GRC callbacks and flowgraph code reach on the reference's block classes (every `def` of the class
    bodies in python/*.py but __init__; listed here once, from the reference tree, so that the test runs without it):
    setters a flowgraph's variable callbacks call must exist under the same names."""
    import ofdm_tools
    methods = {
        'spectrum_sensor_v2': ['set_freqs'],
        'multichannel_scanner': ['set_freqs'],
        'local_worker': ['get_average', 'get_sample_rate', 'set_average', 'set_data_precision', 'set_rate',
                         'set_sample_rate'],
        'spectrum_sweeper': ['get_average', 'get_ffinish', 'get_fstart', 'get_sample_rate', 'get_samples', 'get_tune_delay',
                             'set_average', 'set_ffinish', 'set_fstart', 'set_rate', 'set_sample_rate', 'set_samples',
                             'set_tune_delay'],
        'spectrum_sensor': ['cogeng_rx', 'get_alpha_avg', 'get_channel_space', 'get_noise_estimate', 'get_papr',
                            'get_power_level_ch', 'get_sample_rate', 'get_search_bw', 'get_spectrum_constraint_hz',
                            'get_thr_leveler', 'get_threshold', 'get_tune_freq', 'get_vector_sample', 'send_msg',
                            'set_alpha_avg', 'set_block_length', 'set_channel_space', 'set_fft_len', 'set_papr',
                            'set_sample_rate', 'set_search_bw', 'set_spectrum_constraint_hz', 'set_thr_leveler',
                            'set_time_observation', 'set_tune_freq', 'set_vector_sample', 'work'],
        'coherence_detector': ['get_subject_channels_outcome', 'set_subject_channels_outcome'],
        'ascii_plot': ['get_average', 'get_sample_rate', 'get_tune_freq', 'set_average', 'set_height', 'set_rate',
                       'set_sample_rate', 'set_tune_freq', 'set_width'],
    }
    for cls, names in methods.items():
        missing = [m for m in names if not callable(getattr(getattr(ofdm_tools, cls), m, None))]
        assert not missing, (cls, missing)


def test_block_constructor_signatures_match_the_reference():
    """Argument names/order/defaults of the reference constructors."""
    import ofdm_tools

    def names(cls):
        sig = inspect.signature(cls.__init__)
        return [(p.name, p.default) for p in list(sig.parameters.values())[1:]]

    def check(cls, expected):
        got = names(cls)[:len(expected)]
        assert [n for n, _ in got] == [n for n, _ in expected], cls
        for (n, d), (_, e) in zip(got, expected):
            if e is not inspect.Parameter.empty:
                assert d == e, (cls, n)
    E = inspect.Parameter.empty
    check(ofdm_tools.spectrum_sensor_v2, [('fft_len', E), ('sens_per_sec', E), ('sample_rate', E),
                                          ('channel_space', 1), ('search_bw', 1), ('thr_leveler', 10),
                                          ('tune_freq', 0), ('alpha_avg', 1), ('test_duration', 1), ('period', 3600),
                                          ('trunc_band', 1), ('verbose', False), ('stats', False), ('psd', False),
                                          ('waterfall', False), ('output', False), ('subject_channels', [])])
    check(ofdm_tools.psd_logger, [('fft_len', E), ('rate', E), ('sample_rate', E)])
    check(ofdm_tools.coherence_detector, [('N', E), ('sample_rate', E), ('search_bw', 1), ('threshold', 10),
                                          ('threshold_mtm', 0.2), ('tune_freq', 0), ('alpha_avg', 1),
                                          ('test_duration', 1), ('period', 3600), ('stats', False), ('output', False),
                                          ('rate', 10), ('subject_channels', []), ('valve_callback', None)])
    check(ofdm_tools.spectrum_sweeper, [('rf_receiver', E), ('receiver_type', E), ('fft_len', E), ('sample_rate', E),
                                        ('trunc_sample_rate', E), ('fstart', E), ('ffinish', E), ('rate', E),
                                        ('average', E), ('t_obs', E), ('tune_delay', E), ('max_tu', E)])
    check(ofdm_tools.multichannel_scanner, [('fft_len', E), ('sens_per_sec', E), ('sample_rate', E),
                                            ('channel_space', 1), ('search_bw', 1), ('tune_freq', 0),
                                            ('trunc_band', 1), ('verbose', False), ('output', False),
                                            ('subject_channels', [])])
    check(ofdm_tools.local_worker, [('fft_len', E), ('sample_rate', E), ('average', E), ('rate', E), ('max_tu', E),
                                    ('data_precision', E)])
    check(ofdm_tools.spectrum_sensor_v1, [('fft_len', E), ('sens_per_sec', E), ('sample_rate', E),
                                          ('channel_space', 1), ('search_bw', 1), ('thr_leveler', 10),
                                          ('tune_freq', 0), ('alpha_avg', 1), ('test_duration', 1), ('period', 3600),
                                          ('trunc_band', 1), ('verbose', False), ('psd', False), ('waterfall', False),
                                          ('subject_channels', [])])
    check(ofdm_tools.flanck_detector, [('fft_len', E), ('sens_per_sec', E), ('sample_rate', E), ('channel_space', 1),
                                       ('search_bw', 1), ('thr_leveler', 10), ('tune_freq', 0), ('alpha_avg', 1),
                                       ('test_duration', 1), ('period', 3600), ('trunc_band', 1), ('verbose', False),
                                       ('peak_alpha', 0), ('subject_channels', [])])
    check(ofdm_tools.ascii_plot, [('fft_len', E), ('sample_rate', E), ('tune_freq', E), ('average', E), ('rate', E),
                                  ('width', E), ('height', E)])
    check(ofdm_tools.spectrum_sensor, [('block_length', E), ('sample_rate', 1), ('fft_len', 1), ('channel_space', 1),
                                       ('search_bw', 1), ('method', 'fft'), ('thr_leveler', 10), ('tune_freq', 0),
                                       ('alpha_avg', 1), ('source', None), ('log', False)])
    for fn in ('frange', 'movingaverage', 'src_power', 'src_power_welch', 'src_power_fft', 'xcorr', 'fac',
               'fast_spectrum_scan', 'welch_plot_dB', 'welch_power_estimate', 'clc_power_freq'):
        assert callable(getattr(ofdm_tools.ofdm_cr_tools, fn))


def test_scanner_geometry_and_slice_bounds_match_oracle():
    from ofdm_tools import ofdm_cr_tools as T
    from ofdm_tools.scanner import ChannelScanner
    for (N, Sf, cs, sbw, tb) in [(1024, 1000000, 25e3, 12.5e3, 800000), (16384, 1000000, 15625.0, 10e3, 1000000),
                                 (512, 250001, 12.5e3, 3e3, 200000)]:
        a = ChannelScanner.__new__(ChannelScanner)
        ChannelScanner.__init__(a, N, Sf, cs, sbw, tune_freq=5000, trunc_band=tb)
        b = R.ScannerState(N, Sf, cs, sbw, tune_freq=5000, trunc_band=tb)
        assert a.ax_ch == b.ax_ch and a.trunc_ch == b.trunc_ch and a.bb_freqs == b.bb_freqs
        lo, hi = T._slice_bounds(N, a.Fr, Sf, a.bb_freqs, a.srch_bins)
        psd = np.arange(N, dtype=np.float64)
        ref = R._channel_sums(psd, a.Fr, Sf, a.bb_freqs, a.srch_bins)
        assert [float(psd[l:h].sum()) for l, h in zip(lo, hi)] == ref


def test_keep_one_in_n_capture_of_the_sweeper_without_gpu():
    # the capture side of spectrum_sweeper is pure host logic; build it without a context
    import ofdm_tools
    blk = ofdm_tools.spectrum_sweeper.__new__(ofdm_tools.spectrum_sweeper)
    blk.vector_probe_pts, blk._decim, blk._count = 64, 3, 3
    blk._partial = np.empty(0, np.complex64)
    blk.samples = None
    x = np.arange(64 * 7 + 5).astype(np.complex64)
    for lo in range(0, len(x), 50):
        assert blk.work([x[lo:lo + 50]], []) == len(x[lo:lo + 50])
    assert np.array_equal(blk.get_samples(), x[64 * 5:64 * 6])       # vectors 2 and 5 kept; 5 is the latest


def test_legacy_spectrum_sensor_request_log(tmp_path, monkeypatch):
    """spectrum_sensor.py:59-62,96-120: /tmp/ss_log-<date>-<time>, a geometry header and one CSV row per answered
    quantity (the scan itself is stubbed: no GPU here)."""
    import importlib
    import ofdm_tools
    mod = importlib.import_module('ofdm_tools.spectrum_sensor')
    monkeypatch.setattr(mod, 'fast_spectrum_scan', lambda *a: (2e-7, [1e-8, 3e-7], 4e-8, [25000.0]))
    clock = {'%y%m%d': '261004', '%H%M%S': '101112'}
    monkeypatch.setattr(mod.time, 'strftime', lambda fmt: clock[fmt])
    blk = ofdm_tools.spectrum_sensor(64, sample_rate=1000000, fft_len=64, channel_space=25e3, search_bw=12.5e3,
                                     tune_freq=433000000, log=True, log_dir=str(tmp_path))
    assert blk.log_file.path == str(tmp_path / 'ss_log-261004-101112')
    out = []
    blk.msg_connect('PDU spect_msg', out.append)
    x = (np.arange(64) % 7 - 3 + 1j).astype(np.complex64)
    assert blk.work([x], []) == 64
    for req in ('SC', 'PAPR', 'what'):
        blk.post('PDU from_cogeng', ({}, req))
    papr = blk.get_papr()
    assert open(blk.log_file.path).read().splitlines() == [
        'Time,101112,sample_rate,1000000,channel_space,25000.0,channel_bw,12500.0,tune_freq,433000000',
        'Time,101112,tune_freq[Hz],433000000',
        'Time,101112,threshold[dB],' + str(10 * np.log10(2e-7 + 1e-20)),
        'Time,101112,noise[dB],' + str(10 * np.log10(4e-8 + 1e-20)),
        'Time,101112,spectrum_constraint[Hz],[25000.0]',
        'Time,101112,tune_freq,433000000',
        'Time,101112,papr,' + str(papr),
        'Time,101112,received unknown request']
    assert [m[0] for m in out] == ['thre', 'nois', 'cons', 'papr', 'unkn']
    quiet = ofdm_tools.spectrum_sensor(64, log=False)
    assert quiet.log_file is None


def test_chain_block_watcher_errors_surface_on_the_stream_side():
    """A watcher thread that fails on a vector must not die silently (the depth-2 queue would fill and every later
    vector would be dropped): the error is raised by the next work() / drain(), the watcher carries on, and after
    stop() drain() returns at once."""
    import time
    from ofdm_tools.chain_block import ChainBlockMixin

    class Chain(object):
        def __init__(self):
            self.t = 0

        def push_async(self, in0):
            self.t += 1
            return self.t

        def ticket_rows(self, ticket):
            return 1

        def wait(self, ticket):
            return np.full(4, float(ticket), np.float32), 1

    class Blk(ChainBlockMixin):
        def __init__(self):
            self.seen, self.fail = [], True
            self._chain_init(Chain(), threaded=True)

        def _on_vector(self, row):
            if self.fail:
                raise IOError('disk full')
            self.seen.append(float(row[0]))

    blk = Blk()
    x = np.zeros(8, np.complex64)
    assert blk.work([x], []) == 8
    end = time.monotonic() + 5
    while blk.watch_errors == 0 and time.monotonic() < end:
        time.sleep(0.001)
    with pytest.raises(RuntimeError, match='watcher thread failed'):
        blk.work([x], [])
    blk.fail = False
    assert blk.work([x], []) == 8 and blk.drain(5.0)         # the watcher is still alive and drain() is honest
    assert blk.seen == [2.0] and blk._watch_thread.is_alive()
    blk.fail = True
    blk.work([x], [])
    with pytest.raises(RuntimeError, match='disk full'):
        end = time.monotonic() + 5
        while time.monotonic() < end:
            blk.drain(0.05)
    blk.stop()
    blk.msgq0.insert_tail((99, 99, 1))                        # something nobody will ever look at
    blk._queued += 1
    t0 = time.monotonic()
    assert blk.drain(5.0) and time.monotonic() - t0 < 0.5


class _FakePlan(object):
    def __init__(self, nbins):
        self.nbins, self.calls, self.fail = nbins, 0, False

    def exec(self, vector):
        self.calls += 1
        if self.fail:
            raise RuntimeError('device lost')
        return np.full(self.nbins, float(np.real(vector[0])))


class _FakeCtx(object):
    def welch_plan(self, nfft, nperseg=None, window=None, fs=1.0, fftshift=False, trim_bins=0, db=False):
        assert nperseg == nfft // 4 and len(window) == nperseg and fftshift and db      # spectrum_sweeper.py:263-276
        self.plan = _FakePlan(nfft - 2 * trim_bins)
        return self.plan


class _Rx(object):
    def __init__(self):
        self.tuned = []

    def set_center_freq(self, f, chan):
        self.tuned.append(f)


def test_spectrum_sweeper_stitcher_thread_start_stop_without_gpu():
    """spectrum_sweeper.py:99-105,110-112,207-231: the constructor starts the stitcher; it retunes, sleeps tune_delay,
    takes get_samples(), stitches and sends, for as long as the block runs; setters reach it; stop() joins it; a
    failure inside it surfaces in work()."""
    import time
    import ofdm_tools
    from ofdm_tools import packets
    rx, ctx = _Rx(), _FakeCtx()
    blk = ofdm_tools.spectrum_sweeper(rx, 'rtl', 1024, 2000000, 1750000, 100e6, 107e6, 1e9, 0.25, 1, 1, 1472,
                                      ctx=ctx, start_delay=0.05)
    assert blk._stitch_thread is not None and blk._stitch_thread.daemon
    k, nbins = len(blk.tune_frequencies), 1024 - 2 * blk.excess_bins
    frames = []
    blk.msg_connect('pdus', lambda m: frames.append(m[1]))
    assert blk.sweeps_done == 0 and rx.tuned == []           # still inside the start-up wait (:208)
    x = np.full(blk.vector_probe_pts, 3.0 + 0j, np.complex64)
    end = time.monotonic() + 10
    while blk.sweeps_done < 2 and time.monotonic() < end:
        assert blk.work([x], []) == len(x)                  # the flowgraph thread plays data_colector
        time.sleep(0.001)
    assert blk.sweeps_done >= 2 and blk.captures > 0
    per_sweep = packets.sweeper_fragment_count(4 * k * nbins, 1470)
    assert rx.tuned[:2 * k] == blk.tune_frequencies * 2 and len(frames) >= 2 * per_sweep
    second = packets.reassemble(frames[per_sweep:2 * per_sweep], True)
    assert second.shape == (k * nbins,)
    assert np.allclose(second, (1 - 0.25) * 3.0 + 0.25 * 1e-10)                     # the :227 blend with average
    blk.set_average(0.5)                                      # :142-144 - the running stitcher uses it from now on
    blk.set_tune_delay(0)
    n0 = blk.sweeps_done
    while blk.sweeps_done < n0 + 2 and time.monotonic() < end:
        time.sleep(0.001)
    tail = frames[-per_sweep:] if frames[-1][1] == per_sweep - 1 else None
    blk.stop()
    assert blk._stitch_thread is None and not blk.keep_running
    n = (len(rx.tuned), len(frames))
    time.sleep(0.03)
    assert (len(rx.tuned), len(frames)) == n
    last = packets.reassemble(frames[-per_sweep:], True)
    assert tail is None or np.allclose(last, 0.5 * 3.0 + 0.5 * 1e-10)
    # a stitcher that dies is reported by the next work() call instead of silently never sweeping again
    blk.start(0.0)
    ctx.plan.fail = True
    end = time.monotonic() + 5
    while blk._stitch_error is None and time.monotonic() < end:
        time.sleep(0.001)
    with pytest.raises(RuntimeError, match='stitcher thread died'):
        blk.work([x], [])
    blk.stop()
    # threaded=False: nothing runs until the host calls sweep_once itself
    quiet = ofdm_tools.spectrum_sweeper(_Rx(), 'rtl', 1024, 2000000, 1750000, 100e6, 107e6, 1e9, 0.0, 1, 0, 1472,
                                        ctx=_FakeCtx(), threaded=False)
    time.sleep(0.02)
    assert quiet._stitch_thread is None and quiet.rf_receiver.tuned == [] and quiet.sweeps_done == 0


SHARDED_LOOP_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'gr-ofdm_tools_amd'))
import numpy as np, torch, torch.distributed as dist
import ofdm_tools
from ofdm_tools import packets
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)

class Plan(object):
    pass
class Ctx(object):
    def welch_plan(self, nfft, **kw):
        return Plan()
class Rx(object):
    def __init__(self): self.tuned = []
    def set_center_freq(self, f, chan): self.tuned.append(f)

class Blk(ofdm_tools.spectrum_sweeper):
    def _segment_to_row(self, iq, out_row, device):      # CPU ranks: a stand-in for the HIP plan
        out_row.fill_(float(iq))

rx = Rx()
blk = Blk(rx, 'rtl', 1024, 2000000, 1750000, 100e6, 100e6 + 1750000 * %(nseg)d - 1, 1e9, 0.0, 1, 0, 1472, ctx=Ctx(),
          threaded=False)
k, nbins = len(blk.tune_frequencies), 1024 - 2 * blk.excess_bins
assert k == %(nseg)d, k
frames = []
blk.msg_connect('pdus', lambda m: frames.append(m[1]))
sweep_no = [0]
fail_at = %(fail_at)s
def capture(i, f):
    assert f == blk.tune_frequencies[i]
    if fail_at is not None and rank == 1 and blk.sweeps_done + 1 == fail_at and i == mine[-1]:
        raise IOError('receiver gone')                  # in the LAST of this rank's segments of that sweep
    return 1000.0 * (blk.sweeps_done + 1) + i           # which sweep, which segment
sweeps = %(sweeps)s
mine = list(range(rank, k, world))
blk.start_sharded(capture, rank, world, torch.device('cpu'), publish_rank=0, sweeps=sweeps)
if sweeps is None and fail_at is None and rank == 1:    # stop() on ONE rank ends the loop on all of them
    while blk.sweeps_done < 3: time.sleep(0.001)
    blk.stop()
end = time.monotonic() + 60
while blk.keep_running and time.monotonic() < end: time.sleep(0.002)
assert not blk.keep_running, 'the loop is still running: a rank is stuck in a collective'
blk.stop()
n = blk.sweeps_done
if fail_at is not None:
    # the failing rank took part in that sweep's gather and vote: BOTH ranks left after it, nothing of it was published,
    # and the error comes out of the failing rank's next work()
    assert n == fail_at - 1 and blk.incomplete_sweeps == 1, (n, blk.incomplete_sweeps)
    if rank == 1:
        assert isinstance(blk._stitch_error, IOError)
        try:
            blk.work([np.zeros(4, np.complex64)], [])
            raise SystemExit('work() did not raise')
        except RuntimeError as e:
            assert 'receiver gone' in repr(e.__cause__)
    else:
        assert blk._stitch_error is None, blk._stitch_error
    ran = n + 1
else:
    assert blk._stitch_error is None, blk._stitch_error
    assert n >= 3 and (sweeps is None or n == sweeps), n
    ran = n
assert rx.tuned == [blk.tune_frequencies[i] for i in mine] * ran, (rx.tuned, ran)
per = packets.sweeper_fragment_count(4 * k * nbins, 1470)
if rank == 0:
    assert len(frames) == n * per, (len(frames), n, per)
    for s in range(n):
        wide = packets.reassemble(frames[s * per:(s + 1) * per], True).reshape(k, nbins)
        assert np.array_equal(wide[:, 0], 1000.0 * (s + 1) + np.arange(k)), (s, wide[:, 0])
else:
    assert frames == []
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok', n)
'''


@pytest.mark.parametrize('nseg,sweeps,fail_at', [(8, 4, None), (5, None, None), (6, None, 3)])
def test_sharded_stitcher_loop_world_size_2_gloo(nseg, sweeps, fail_at, tmp_path):
    """start_sharded: each rank retunes only to its own segments, the sweeps come out in tune order on the publishing
    rank, and every rank leaves the loop after the same sweep (sweeps=N, stop() on one rank, or - round 4's advisor
    finding - a capture that fails on one rank: that rank still joins the sweep's gather and vote)."""
    script = tmp_path / 'worker.py'
    script.write_text(SHARDED_LOOP_WORKER % {'root': ROOT, 'nseg': nseg, 'sweeps': sweeps, 'fail_at': fail_at})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29540 + nseg), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert outs[0].split()[-1] == outs[1].split()[-1]           # the same number of sweeps on both ranks


GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'gr-ofdm_tools_amd'))
import numpy as np, torch, torch.distributed as dist
from ofdm_tools import sweep
from oracle import ref_cpu as R
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
nseg, nfft, excess, fs = %(nseg)d, 1024, 64, 2.0e6
nbins = nfft - 2 * excess
done = []
def capture(i):
    return R.synth_iq(8192, 2000 + i)
def compute(iq, out_row):            # the oracle stands in for the HIP plan on CPU ranks
    done.append(1)
    out_row.copy_(torch.from_numpy(R.sweeper_src_power(iq, nfft, fs, excess).astype(np.float32)))
wide = sweep.sweep_psd(capture, compute, nseg, nbins, torch.device('cpu'), rank, world)
ref = np.concatenate([R.sweeper_src_power(capture(i), nfft, fs, excess) for i in range(nseg)]).astype(np.float32)
assert wide.shape == (nseg * nbins,), wide.shape
assert np.array_equal(wide.numpy(), ref), 'tune order broken'
assert len(done) == len(sweep.shard_segments(nseg, rank, world))
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok', len(done))
'''


@pytest.mark.parametrize('nseg', [8, 5])
def test_sweep_sharding_world_size_2_gloo(nseg, tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(GLOO_WORKER % {'root': ROOT, 'nseg': nseg})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + nseg), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert 'ok %d' % ((nseg + 1) // 2) in outs[0] and 'ok %d' % (nseg // 2) in outs[1]


def test_sweep_sharding_world_size_8_gloo_one_segment_per_rank(tmp_path):
    """BASELINE config 4's own shape - 8 RF segments, 8 ranks, one segment each, ONE all-gather, wideband PSD in tune
    order on every rank - rehearsed over gloo on the CPU (the oracle stands in for the HIP plan): the 8-GPU run is the
    driver's to launch, the partition and the reorder are checked here."""
    script = tmp_path / 'worker8.py'
    script.write_text(GLOO_WORKER % {'root': ROOT, 'nseg': 8})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29618', WORLD_SIZE='8', OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(8)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok 1' % r in o, o            # every rank computed exactly its own segment


def test_sweep_shard_assignment():
    from ofdm_tools import sweep
    assert sweep.shard_segments(8, 3, 8) == [3]
    assert sweep.shard_segments(8, 1, 4) == [1, 5]
    assert sorted(sum((sweep.shard_segments(13, r, 4) for r in range(4)), [])) == list(range(13))
    assert sweep.segments_per_rank(13, 4) == 4


def test_sensing_log_file_formats(tmp_path):
    """On-disk formats of the reference's file_logger (python/ofdm_cr_tools.py:2010-2058)."""
    from ofdm_tools.sensing_log import logger
    lg = logger(1024, 3600, 10, directory=str(tmp_path))
    lg.settings = {'fft_len': 1024, 'n_measurements': 3}
    lg.cumulative_statistics = {100.0e6: 2}
    lg.periodic_statistic = {100.0e6: 1}
    lg.set_cumulative_psd(np.arange(4, dtype=np.float32))
    lg.set_periodic_psd_peaks(np.arange(4, dtype=np.float32) * 2)
    lg.set_cumulative_max_power(np.array([1.0, 2.0]))
    lg.set_periodic_max_power(np.array([0.5, 2.0]))
    lg.cumulative_waterfall.append(np.array([1.234e-5, 6.5e-7], np.float32))
    out = lg.flush()
    want = ("settings {'fft_len': 1024, 'n_measurements': 3}\nstatistics {100000000.0: 2}\n") * 2
    assert open(out['stat']).read() == want
    assert 'statistics {100000000.0: 1}' in open(out['periodic_stat']).read()
    assert np.array_equal(np.load(out['psd']), np.arange(4, dtype=np.float32))
    assert np.array_equal(np.load(out['periodic_psd']), np.arange(4, dtype=np.float32) * 2)
    assert np.array_equal(np.load(out['periodic_max_power']), [0.5, 2.0])
    assert open(out['waterfall']).read() == '1.23e-05,6.50e-07\n'
    # a new period started: periodic state is reset, cumulative state kept, waterfall appends
    assert lg.periodic_psd_peaks is None and lg.periodic_statistic == {} and lg.cumulative_waterfall == []
    assert lg.cumulative_statistics == {100.0e6: 2}
    lg.cumulative_waterfall.append(np.array([2.0, 3.0], np.float32))
    out2 = lg.flush()
    assert open(out2['waterfall']).read() == '1.23e-05,6.50e-07\n2.00e+00,3.00e+00\n'
    assert np.load(out2['periodic_psd'], allow_pickle=True).item() is None


def test_ref_f2_sensing_log_session_against_the_reference_logger(golden, tmp_path, monkeypatch):
    """f2 against the reference's OWN ``logger`` + ``file_logger`` (ofdm_cr_tools.py:1850-2107; ref_sensing_log.npz holds
    every file they left behind for one scripted campaign on a frozen clock): the product logger, driven through the
    same setters at the same moments, must leave the same directory - names, statistics text and waterfall rows byte
    for byte, the np.save files value for value (dtype and shape included; ``None`` where a period saw no update)."""
    import ast
    import io
    from ofdm_tools import sensing_log
    g = golden('ref_sensing_log.npz')
    S = ast.literal_eval(str(g['session']))
    stamps, slept = S['stamps'], []

    class Clock(object):
        k = 0

        def strftime(self, fmt):
            if fmt == '%y%m%d':
                return stamps[self.k][0]
            if fmt == '%H%M':
                return stamps[0][2]
            self.k += 1
            return stamps[self.k - 1][1]

    def apply(lg, phase):
        for key, val in phase.items():
            if key in ('cumulative_psd', 'periodic_psd_peaks', 'cumulative_max_power', 'periodic_max_power'):
                val = np.array(val, np.float32 if 'psd' in key else np.float64)
            elif key == 'cumulative_waterfall':
                val = [np.array(r, np.float32) for r in val]
            getattr(lg, 'set_' + key)(val)

    class FakeDatetime(object):
        class datetime(object):
            @staticmethod
            def now():
                return 1000.0 + S['periodicity'] * len(slept)

        @staticmethod
        def timedelta(seconds):
            return seconds

    monkeypatch.setattr(sensing_log, 'time', Clock())
    monkeypatch.setattr(sensing_log, 'datetime', FakeDatetime)
    monkeypatch.setenv('HOME', str(tmp_path))
    lg = sensing_log.logger(S['fft_len'], S['periodicity'], S['test_duration'])
    lg.set_settings(S['settings'])
    apply(lg, S['phases'][0])

    def wait(seconds):                      # the file_logger thread's sleep: the watchers hand over the next phase
        slept.append(seconds)
        apply(lg, S['phases'][len(slept)])
        return False

    lg._fl_stop.wait = wait
    lg._prepare_file_logger()
    lg._file_logger_run()
    assert slept == [S['periodicity']] * 2 and lg.files_written == 3
    got = sorted(os.path.relpath(os.path.join(r, f), str(tmp_path)) for r, _, fs in os.walk(str(tmp_path)) for f in fs)
    assert got == list(g['names'])
    for i, name in enumerate(g['names']):
        want = bytes(g['file_%d' % i])
        have = open(os.path.join(str(tmp_path), name), 'rb').read()
        if name.endswith('.log') or 'waterfall' in name:
            assert have == want, name
        else:
            a, b = np.load(io.BytesIO(have), allow_pickle=True), np.load(io.BytesIO(want), allow_pickle=True)
            assert a.dtype == b.dtype and a.shape == b.shape, name
            assert (a.item() is None and b.item() is None) if a.dtype == object else np.array_equal(a, b), name


# ---- multi-rank long-stream Welch / coherence and the batched scanner over gloo ----

GLOO_LONG_STREAM = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'gr-ofdm_tools_amd'))
import numpy as np, torch, torch.distributed as dist
from ofdm_tools import sweep
from oracle import ref_cpu as R
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
nfft, step, fs = 1024, 512, 2.0e6
n = %(nsamples)d
x = R.synth_iq(n, 1002)
y = (0.7 * np.roll(x, 5) + 0.5 * R.synth_iq(n, 1004)).astype(np.complex64)
win = R.get_window('hann', nfft)
scale = 1.0 / (fs * (win * win).sum())
nseg_all = (n - (nfft - step)) // step
calls = []

# the oracle stands in for oth_welch_partial_dev / oth_csd_partial_dev on the CPU ranks: raw sums of one run
def welch_partial(first, cnt, out):
    calls.append((first, cnt))
    xs = x[first:first + cnt]
    k = (cnt - (nfft - step)) // step
    _, p = R.welch_np(xs, nperseg=nfft, nfft=nfft, scaling='none')
    out.copy_(torch.from_numpy((p * k).astype(np.float32)))
    return k

def csd_partial(first, cnt, out):
    xs, ys = x[first:first + cnt], y[first:first + cnt]
    k = (cnt - (nfft - step)) // step
    _, pxx = R.welch_np(xs, nperseg=nfft, nfft=nfft, scaling='none')
    _, pyy = R.welch_np(ys, nperseg=nfft, nfft=nfft, scaling='none')
    _, pxy = R.csd_np(xs, ys, nperseg=nfft, nfft=nfft, scaling='none')
    v = np.concatenate([pxx * k, pyy * k, np.stack([pxy.real, pxy.imag], 1).reshape(-1) * k]).astype(np.float32)
    out.copy_(torch.from_numpy(v))
    return k

dev = torch.device('cpu')
psd, nseg = sweep.welch_time_sharded(welch_partial, lambda s, k: s.to(torch.float64) * (scale / k), n, nfft, step,
                                     nfft, dev, rank, world)
assert nseg == nseg_all, (nseg, nseg_all)
_, ref = R.welch_np(x, fs=fs, nperseg=nfft, nfft=nfft)
err = float(np.max(np.abs(psd.numpy() - ref) / ref))
assert err < 2e-6, err
# every rank holds the same bits
mine = psd.to(torch.float64).clone()
both = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert all(torch.equal(both[0], b) for b in both)
first, cnt, s0, k = sweep.time_shard(n, nfft, step, rank, world)
assert calls == ([(first, cnt)] if k else []), (calls, first, cnt)

def csd_scale(s, k):
    s = s.to(torch.float64).numpy()
    pxx, pyy, pxy = s[:nfft], s[nfft:2 * nfft], s[2 * nfft:].reshape(-1, 2)
    pxy = pxy[:, 0] + 1j * pxy[:, 1]
    return pxx * scale / k, pyy * scale / k, pxy * scale / k, np.abs(pxy) ** 2 / (pxx * pyy)
(pxx, pyy, pxy, cxy), nseg = sweep.welch_time_sharded(csd_partial, csd_scale, n, nfft, step, 4 * nfft, dev, rank, world)
_, rc, rxx, ryy, rxy = R.coherence_np(x, y, fs=fs, nperseg=nfft, nfft=nfft)
assert nseg == nseg_all
assert np.max(np.abs(pxx - rxx) / rxx) < 2e-6 and np.max(np.abs(pyy - ryy) / ryy) < 2e-6
assert np.max(np.abs(pxy - rxy) / np.sqrt(rxx * ryy)) < 2e-6 and np.max(np.abs(cxy - rc)) < 1e-5

# batched scanner: channel c on rank c mod world, rows come back in channel order
nch, nbins = %(nch)d, 48
mine = sweep.shard_segments(nch, rank, world)
local = torch.zeros((sweep.segments_per_rank(nch, world), nbins))
for j, c in enumerate(mine):
    local[j] = torch.arange(nbins, dtype=torch.float32) + 1000.0 * c
rows = sweep.gather_rows(local, nch, rank, world)
assert rows.shape == (nch, nbins)
assert torch.equal(rows[:, 0], 1000.0 * torch.arange(nch, dtype=torch.float32))
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok', nseg, k)
'''


@pytest.mark.parametrize('nsamples,nch', [(512 * 40 + 512, 8), (512 * 37 + 512 + 99, 5), (1024, 3)])
def test_long_stream_and_channel_sharding_world_size_2_gloo(nsamples, nch, tmp_path):
    """Welch and coherence of one long stream cut into contiguous time runs with a halo (even, ragged, and a
    stream with a single segment so that rank 1 owns nothing), partial sums + counts all-gathered and summed
    in rank order; batched-scanner rows gathered back into channel order."""
    script = tmp_path / 'worker.py'
    script.write_text(GLOO_LONG_STREAM % {'root': ROOT, 'nsamples': nsamples, 'nch': nch})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29600 + nch), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    nseg = (nsamples - 512) // 512
    per = -(-nseg // 2)
    assert 'ok %d %d' % (nseg, min(per, nseg)) in outs[0] and 'ok %d %d' % (nseg, nseg - min(per, nseg)) in outs[1]


def test_long_stream_and_64_channel_sharding_world_size_8_gloo(tmp_path):
    """BASELINE config 5's shape over 8 ranks - 64 channel rows, channel c on rank c mod 8, gathered back into channel
    order - and a long stream cut into 8 time runs with halos (Welch and coherence partial sums all-gathered and summed
    in rank order, the same bits on every rank), rehearsed over gloo on the CPU."""
    nsamples = 512 * 83 + 512 + 77
    script = tmp_path / 'worker8.py'
    script.write_text(GLOO_LONG_STREAM % {'root': ROOT, 'nsamples': nsamples, 'nch': 64})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29664', WORLD_SIZE='8', OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(8)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    nseg = (nsamples - 512) // 512
    per = -(-nseg // 8)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok %d %d' % (r, nseg, max(0, min(per, nseg - r * per))) in o, o


def test_time_shard_covers_every_segment_once():
    from ofdm_tools import sweep
    for n, nper, step, world in [(2 ** 20, 4096, 2048, 8), (100000, 4096, 2048, 3), (4096, 4096, 2048, 4),
                                 (4095, 4096, 2048, 2), (50000, 1024, 1024, 5), (9000, 1024, 300, 7)]:
        nseg = (n - (nper - step)) // step if n >= nper else 0
        seen = []
        for r in range(world):
            first, cnt, s0, k = sweep.time_shard(n, nper, step, r, world)
            assert first == s0 * step and (cnt == 0) == (k == 0)
            if k:
                assert cnt == (k - 1) * step + nper and first + cnt <= n
            seen += list(range(s0, s0 + k))
        assert seen == list(range(nseg))


def test_ascii_plotter_matches_the_reference_byte_for_byte(golden):
    """tests/golden/ref_ascii_plot.npz holds the text the reference's OWN ascii_plotter.make_plot
    (ascii_plot.py:169-228) produced for four rows (make_golden.py --reference runs its unmodified method body under
    Python-2 integer division): the product renderer and the oracle's restatement must both reproduce every byte."""
    from ofdm_tools.ascii_plot import ascii_plotter
    g = golden('ref_ascii_plot.npz')
    assert str(g['source']) == 'reference'
    for i in range(int(g['n'])):
        W, H, N, Sf = (int(v) for v in g['case_%d' % i][:4])
        tf = float(g['case_%d' % i][4])
        want = g['text_%d' % i].tobytes().decode('ascii')
        row = g['row_%d' % i]
        assert ascii_plotter(W, H, tf, Sf, N).make_plot(row) == want, i
        assert R.ascii_make_plot(row, W, H, tf, Sf, N) == want, i


def test_ascii_plotter_layout_and_oracle():
    """ascii_plot.py:154-228: the text plot of a dB row (host work; the block's chain is a GPU test)."""
    from ofdm_tools.ascii_plot import ascii_plotter
    N, W, H = 1024, 64, 20
    rng = np.random.default_rng(9)
    row = (-90 + 25 * np.exp(-0.5 * ((np.arange(N) - 700) / 30.0) ** 2) + rng.random(N)).astype(np.float32)
    pl = ascii_plotter(W, H, 100.0e6, 2000000, N)
    txt = pl.make_plot(row)
    assert txt == R.ascii_make_plot(row, W, H, 100.0e6, 2000000, N)
    lines = txt.split('\n')
    assert len(lines) == H + 3 and all(len(ln) == 7 + 2 * W for ln in lines[:H])
    top = [ln[7::2] for ln in lines[:H]]                         # one character per column, top row first
    col = 700 // (N // W)                                        # the bump sits in this column: tallest bar
    peak_row = [i for i, ln in enumerate(top) if ln[col] == '^'][0]
    assert peak_row == min(i for i, ln in enumerate(top) for c in ln if c == '^')
    assert all(ln[W // 2] == '*' for ln in top)                  # centre marker
    assert lines[H + 1].startswith('Tune freq: 100.0 MHz, Sample rate: 2.0 MS/s, FFT: 1024 W:64 L:20')
    assert pl.make_plot(row) == txt                              # a second call gives the same picture again
    # a level above the top text row (the scale divides by floor(max - min)) is clipped, not an IndexError
    tall = np.full(N, -80.0, np.float32)
    tall[:N // W] = -69.05                                       # column 0 at 10.95 dB over floor(10.95) = 10
    lv, lo, span = ascii_plotter(W, H, 0.0, 2000000, N).column_levels(tall)
    assert span == 10 and lv[0] == H - 1 and lv[1] == 0


def test_chain_block_plumbing_with_a_fake_chain():
    """ofdm_tools.chain_block without a GPU: a fake chain stands in for oth_chain_*.  Default = watcher thread
    (spectrum_sensor_v2.py:138-155): work() only enqueues, the vector arrives through the lossy depth-2 queue, drain()
    waits for it, rows_total counts every vector the pushes produced (dropped tickets included), a stalled watcher
    drops instead of back-pressuring, stop() joins, and a block that is dropped without stop() takes its thread along."""
    import gc
    import threading
    import time
    import weakref
    from ofdm_tools.chain_block import ChainBlockMixin
    from ofdm_tools import _hip

    class FakeChain(object):
        def __init__(self):
            self.t, self.rows = 0, {}

        def push_async(self, x):
            self.t += 1
            self.rows[self.t] = (np.full(4, float(self.t), np.float32), len(x) // 4)
            return self.t

        def ticket_rows(self, t):
            return self.rows[t][1]

        def wait(self, t):
            if t <= self.t - 4:
                raise _hip.HipError(-5, 'oth_chain_wait', 'overwritten')
            row, n = self.rows[t]
            return (row if n else None), n

    class Blk(ChainBlockMixin):
        def __init__(self, threaded=True):
            self.seen, self.gate = [], threading.Event()
            self.gate.set()
            self._chain_init(FakeChain(), threaded)

        def _on_vector(self, row):
            self.gate.wait(5.0)
            self.seen.append((float(row[0]), self.vector_rows_end, self.vector_nrows))

    x = np.zeros(8, np.complex64)
    blk = Blk()
    assert blk._threaded and blk._watch_thread.is_alive()
    assert blk.work([x], []) == 8 and blk.drain(2.0)
    assert blk.seen == [(1.0, 2, 2)] and blk.rows_total == 2
    blk.work([x[:3]], [])                       # a push that completes no vector: nothing to hand on
    assert blk.drain(2.0) and len(blk.seen) == 1 and blk.rows_total == 2
    blk.gate.clear()                            # stall the watcher: depth 2 + the one in progress survive
    t0 = time.perf_counter()
    for _ in range(12):
        blk.work([x], [])
    assert time.perf_counter() - t0 < 0.5 and blk.msgq0.dropped >= 9 and blk.rows_total == 26
    blk.gate.set()
    assert blk.drain(5.0)
    assert len(blk.seen) + blk.vectors_lost + blk.msgq0.dropped == 13
    assert blk.seen[-1][1] <= 26 and all(n == 2 for _, _, n in blk.seen)
    th = blk._watch_thread
    assert blk.stop() and not th.is_alive()
    # inline form: same bookkeeping, no thread
    inline = Blk(threaded=False)
    inline.work([x], [])
    assert inline.seen == [(1.0, 2, 2)] and inline._watch_thread is None and inline.drain(0.1)
    # a dropped block does not keep its watcher (or itself) alive
    orphan = Blk()
    th, ref = orphan._watch_thread, weakref.ref(orphan)
    del orphan
    gc.collect()
    th.join(2.0)
    assert ref() is None and not th.is_alive()
