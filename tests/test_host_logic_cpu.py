"""CPU tests of the host logic around the HIP path: wire format, decision stage, block API
surface, and the multi-rank sweep sharding over gloo (world_size 2)."""
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
