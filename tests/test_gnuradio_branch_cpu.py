"""The `gnuradio.gr.sync_block` branch of ofdm_tools.gr_compat - the configuration an existing flowgraph runs - executed
against strict stand-ins for `gnuradio.gr` and `pmt` (tests/gr_standin.py; GNU Radio itself is not in this image).

What the reference's blocks promise a flowgraph (and GRC's generated code relies on):
  python/spectrum_sensor.py:37-40,66-69     gr.sync_block.__init__(name, in_sig=[np.complex64], out_sig=None), ports
                                            registered as pmt.intern(...) symbols, a handler on 'PDU from_cogeng'
  python/spectrum_sensor.py:71-128          work() returns the consumed count; 'SC' -> three pmt.cons(meta, data) PDUs
  python/spectrum_sensor_v2.py:78-82        freq_out_0..3 + freq_msg_PDU; ("freq", value) pairs (:162-165)
  python/local_worker.py:74,168-171         'pdus': pmt.cons(PMT_NIL, u8vector)
  grc/ofdm_tools_*.xml                      the message <source>/<sink> names of the ten sensing blocks
"""
import numpy as np
import pytest

from gr_standin import installed

# message ports of the reference's GRC block descriptors (grc/ofdm_tools_<block>.xml <sink>/<source> of type message);
# coherence_detector's XML declares 'msg_PDU', which the reference class never registers (coherence_detector.py:39-82)
GRC_PORTS = {
    'spectrum_sensor_v2': ((), ('freq_out_0', 'freq_out_1', 'freq_out_2', 'freq_out_3', 'freq_msg_PDU')),
    'multichannel_scanner': ((), ('freq_out_0', 'freq_out_1', 'freq_out_2', 'freq_out_3', 'freq_msg_PDU')),
    'psd_logger': ((), ()),
    'coherence_detector': ((), ()),
    'spectrum_sweeper': ((), ('pdus',)),
    'local_worker': ((), ('pdus',)),
    'spectrum_sensor': (('PDU from_cogeng',), ('PDU spect_msg',)),
    'spectrum_sensor_v1': ((), ()),
    'flanck_detector': ((), ()),
    'ascii_plot': ((), ('pkt_out',)),
}


class FakeChain(object):
    """oth_chain_* without a GPU: every push of k * nfft samples 'produces' k vectors; the latest row is a ramp."""

    def __init__(self, nfft):
        self.nfft, self.t, self.rows = nfft, 0, {}
        self.calls = []

    def push_async(self, x):
        self.t += 1
        self.rows[self.t] = len(x) // self.nfft
        return self.t

    def ticket_rows(self, t):
        return self.rows[t]

    def wait(self, t):
        n = self.rows[t]
        if any(c[0] == 'iir' for c in self.calls):      # the IIR + nlog10 chains hand out dB rows
            row = np.linspace(-120.0, -60.0, self.nfft).astype(np.float32)
        else:
            row = np.linspace(1e-9, 2e-9, self.nfft).astype(np.float32)
        return (row if n else None), n

    def set_keep_one_in_n(self, n):
        self.calls.append(('keep', n))

    def set_iir_log(self, a, k):
        self.calls.append(('iir', a, k))

    def set_peak_hold(self, on):
        self.calls.append(('peak', on))

    def peak(self):
        return np.ones(self.nfft, np.float32)

    def close(self):
        pass


class FakePlan(object):
    def __init__(self, nfft, trim):
        self.out_len = nfft - 2 * trim

    def exec(self, x):
        self.last_nseg = 1
        return np.full(self.out_len, -90.0, np.float32)

    def close(self):
        pass


class FakeCtx(object):
    def chain(self, nfft, window=None, fftshift=True, epilogue=0, keep_one_in_n=1):
        return FakeChain(nfft)

    def welch_plan(self, nfft, nperseg=None, noverlap=None, window=None, detrend=1, scaling=1, fs=1.0, fftshift=False,
                   trim_bins=0, db=False, kernel=0):
        return FakePlan(nfft, trim_bins)

    def channel_power(self, psd, srch_bins, lo, hi, want_movavg=False):
        m = int(srch_bins)
        ma = np.abs(np.convolve(np.asarray(psd, np.float64), np.ones(m) / float(srch_bins), 'same')) if m > 0 else psd
        out = np.array([ma[a:b].sum() for a, b in zip(lo, hi)], np.float32)
        return (out, ma.astype(np.float32)) if want_movavg else out


class Rx(object):
    def __init__(self):
        self.tuned = []

    def set_center_freq(self, f, chan):
        self.tuned.append(f)


def build_all(T, tmp_path):
    ctx = FakeCtx()
    N, Sf = 256, 1000000
    kw = dict(ctx=ctx, threaded=False)
    # four subject channels that sit exactly on the channel axis (the top-4 publication needs four: :228-237)
    subj = list(T.scanner.ChannelScanner(N, Sf, 25e3, 12.5e3, 433e6, 800000, 10, 0.5, ctx).ax_ch[3:7])
    return {
        'spectrum_sensor_v2': T.spectrum_sensor_v2(N, 10, Sf, 25e3, 12.5e3, 10, 433e6, 0.5, 1, 3600, 800000, False, False,
                                                   False, False, True, subj, **kw),
        'multichannel_scanner': T.multichannel_scanner(N, 10, Sf, 25e3, 12.5e3, 433e6, 800000, False, True,
                                                       subj, **kw),
        'psd_logger': T.psd_logger(N, 10, Sf, mat_file=str(tmp_path / 'psd.mat'), **kw),
        'coherence_detector': T.coherence_detector(N, Sf, subject_channels=[25e3], valve_callback=lambda v: None),
        'spectrum_sweeper': T.spectrum_sweeper(Rx(), 'uhd', N, Sf, 800000, 400e6, 402e6, 10, 0.5, 0.1, 1, 1472, ctx=ctx,
                                               threaded=False),
        'local_worker': T.local_worker(N, Sf, 0.8, 10, 1472, 'float32', **kw),
        'spectrum_sensor': T.spectrum_sensor(N, Sf, N, 25e3, 12.5e3, 'welch', 10, 433e6, 0.5, ctx=ctx),
        'spectrum_sensor_v1': T.spectrum_sensor_v1(N, 10, Sf, 25e3, 12.5e3, 10, 433e6, 0.5, 1, 3600, 800000, **kw),
        'flanck_detector': T.flanck_detector(N, 10, Sf, 25e3, 12.5e3, 10, 433e6, 0.5, 1, 3600, 800000,
                                             log_directory=str(tmp_path), **kw),
        'ascii_plot': T.ascii_plot(N, Sf, 433e6, 0.8, 10, 80, 20, **kw),
    }


def test_blocks_derive_from_gr_sync_block_and_register_the_grc_ports(tmp_path):
    with installed() as (T, pmt, gr):
        assert T.gr_compat.HAVE_GNURADIO and issubclass(T.gr_compat.sync_block, gr.sync_block)
        blocks = build_all(T, tmp_path)
        for name, blk in blocks.items():
            assert isinstance(blk, gr.sync_block), name
            # constructor contract of python/spectrum_sensor.py:37-40 / the io_signatures of the hier blocks
            if name == 'coherence_detector':
                assert blk.gr_in_sig == [(np.float32, 256)] * 3, name                 # coherence_detector.py:45
            else:
                assert blk.gr_in_sig == [np.complex64], name
            assert blk.gr_out_sig is None, name
            assert isinstance(blk.gr_name, str) and blk.gr_name
            want_in, want_out = GRC_PORTS[name]
            assert tuple(blk.gr_in_ports) == want_in, (name, blk.gr_in_ports)
            assert tuple(blk.gr_out_ports) == want_out, (name, blk.gr_out_ports)
        assert blocks['spectrum_sensor'].gr_handlers['PDU from_cogeng'].__name__ == 'cogeng_rx'
        for blk in blocks.values():
            if hasattr(blk, 'stop'):
                blk.stop()


def test_work_returns_the_consumed_count_and_messages_are_pmts(tmp_path):
    with installed() as (T, pmt, gr):
        blocks = build_all(T, tmp_path)
        N = 256
        x = (np.arange(3 * N) % 11 - 5 + 0.5j).astype(np.complex64)
        for name in ('spectrum_sensor_v2', 'multichannel_scanner', 'psd_logger', 'local_worker', 'spectrum_sensor_v1',
                     'flanck_detector', 'ascii_plot', 'spectrum_sweeper'):
            assert blocks[name].work([x], []) == len(x), name
        assert blocks['spectrum_sensor'].work([x], []) == N            # consumes at most block_length (:72-75)
        rows = np.ones((2, N), np.float32)
        assert blocks['coherence_detector'].work([rows, rows, rows], []) == 2

        # local_worker: pmt.cons(PMT_NIL, u8vector) frames, [n_frags][frag_id][payload] (local_worker.py:147-172)
        pubs = blocks['local_worker'].gr_published
        assert pubs and all(port == 'pdus' for port, _ in pubs)
        meta, body = pmt.car(pubs[0][1]), pmt.cdr(pubs[0][1])
        assert meta is pmt.PMT_NIL and pmt.is_u8vector(body)
        frame = bytes(bytearray(pmt.u8vector_elements(body)))
        assert frame[0] == len(pubs) and frame[1] == 0 and len(frame) <= 1472

        # the scanner's top-4 publication: ("freq", value) pairs on freq_out_i (multichannel_scanner.py:227-239)
        pubs = blocks['multichannel_scanner'].gr_published
        assert {p for p, _ in pubs} <= {'freq_out_0', 'freq_out_1', 'freq_out_2', 'freq_out_3', 'freq_msg_PDU'} and pubs
        for port, msg in pubs:
            assert pmt.is_pair(msg)
            if port.startswith('freq_out'):
                assert str(pmt.car(msg)) == 'freq' and isinstance(pmt.to_python(pmt.cdr(msg)), float)

        # v2: set_freqs only changes the strobes' messages (:157-165); they are PMT pairs too
        v2 = blocks['spectrum_sensor_v2']
        v2.set_freqs(433e6 + 1.0, 433e6 + 2.0, 433e6 + 3.0, 433e6 + 4.0)      # the differential frequency goes out
        assert [pmt.to_python(pmt.cdr(s.msg())) for s in v2._strobes] == [1.0, 2.0, 3.0, 4.0]

        # ascii_plot: ('ascii', text) on pkt_out
        port, msg = blocks['ascii_plot'].gr_published[-1]
        assert port == 'pkt_out' and str(pmt.car(msg)) == 'ascii' and isinstance(pmt.to_python(pmt.cdr(msg)), str)

        # the sweeper's stitcher publishes u8vector PDUs on 'pdus' (spectrum_sweeper.py:230-258)
        sw = blocks['spectrum_sweeper']
        sw.tune_delay = 0.0
        sw.sweep_once()
        assert sw.gr_published and all(p == 'pdus' and pmt.is_u8vector(pmt.cdr(m)) for p, m in sw.gr_published)
        for blk in blocks.values():
            if hasattr(blk, 'stop'):
                blk.stop()


def test_legacy_sensor_answers_pmt_requests(tmp_path, monkeypatch):
    """python/spectrum_sensor.py:77-128 with real PMT shapes: a cons(meta, 'SC') request is answered by three
    cons(to_pmt(field), to_pmt(value)) PDUs; numpy scalars in the results must not reach pmt.to_pmt as such."""
    with installed() as (T, pmt, gr):
        import importlib
        mod = importlib.import_module('ofdm_tools.spectrum_sensor')
        monkeypatch.setattr(mod, 'fast_spectrum_scan',
                            lambda *a: (np.float32(2e-7), [np.float64(1e-8), np.float64(3e-7)], np.float64(4e-8),
                                        [np.float64(25000.0)]))
        blk = T.spectrum_sensor(64, 1000000, 64, 25e3, 12.5e3, 'welch', 10, 433000000, 0.5, ctx=FakeCtx())
        assert blk.work([np.ones(100, np.complex64)], []) == 64
        post = lambda m: blk.scheduler_post('PDU from_cogeng', m)      # noqa: E731
        post(pmt.cons(pmt.to_pmt({'REQ': 'x'}), pmt.intern('SC')))
        post(pmt.cons(pmt.PMT_NIL, pmt.intern('PAPR')))
        post(pmt.cons(pmt.PMT_NIL, pmt.intern('what')))
        post(pmt.intern('not a pdu'))                                   # no reply (:79-83)
        got = [(port, str(pmt.car(m)), pmt.to_python(pmt.cdr(m))) for port, m in blk.gr_published]
        assert [g[0] for g in got] == ['PDU spect_msg'] * 5
        assert [g[1] for g in got] == ['thre', 'nois', 'cons', 'papr', 'unkn']
        assert got[0][2] == pytest.approx(2e-7) and got[1][2] == pytest.approx(4e-8) and got[2][2] == [25000.0]
        assert all(type(v) in (float, list, str) for _, _, v in got)
        assert got[4][2] == 'received unknown request'


def test_the_local_branch_is_back_after_the_stand_ins_leave():
    import ofdm_tools
    from ofdm_tools import gr_compat
    assert not gr_compat.HAVE_GNURADIO and gr_compat.sync_block is gr_compat._LocalSyncBlock
    assert issubclass(ofdm_tools.spectrum_sensor, gr_compat._LocalSyncBlock)
