#!/usr/bin/env python3
"""bench.py - IQ Msamples/s through the 4096-point Welch PSD on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]                 (N > 1: spawns its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic IQ that is already resident in HBM.

N = 1  workload "C2" (the configuration BASELINE.json's metric is quoted on): one 2^28-sample complex64
       stream (2 GiB), Hann, nperseg = nfft = 4096, 50 % overlap, detrend constant, density scaling
       (the welch() call of ofdm_cr_tools.py:342): welch4096ws kernel + the cross-workgroup finalize.
       Extra keys on the same line: `host_visible_ms_per_step` (the same step through the host-output entry point:
       launch to PSD in host memory, SURVEY 8d's end point), `sweep_c4` (the N > 1 workload run on this one GPU, so
       that the N-GPU speed-up is a plain division), `sweep_c4_ref` (the same sweep through the call the reference's
       block actually makes, spectrum_sweeper.py:263: flattop, nperseg = 1024 zero-padded to 4096, step 512),
       `csd_c3` (BASELINE config 3: two-channel cross spectrum / coherence, 2 x 2^26 samples, 16 B per sample
       pair), `scan_c5` (BASELINE config 5: 64 channel streams x 2^22 samples, 16384-pt rectangular |X|^2/N^2 mean +
       the device decision stage), `c1` (BASELINE config 1 at its own size: 2^20 samples, 1024-pt rectangular chain, the 128
       eight-row means + a7 channel sums), `welch_32768` / `welch_65536` (the lengths above the tuned kernels: one workgroup / a
       pair of workgroups per segment, 2^27
       samples), each with its own `roofline` whose `kernel` is the recipe the library recorded for
       the launch (oth__debug_last_recipe) and whose `traffic` is the figure of the builder's rocprofv3 PMC passes of
       the same configuration (profiles/traffic.json: not measured by this run); `h2d_inclusive` (host buffer -> PSD through the streaming entry point);
       `cpu_baseline` (+ `_parallel`, `_c5`, `_c1`).
N > 1  --workload c5: BASELINE config 5 over the ranks (64 channel streams, stream c on rank c mod N, PSD rows + decision
       stage per rank, one all-gather of rows + noise floors + channel powers: BatchScanPlan.scan_sharded); --workload c2: the
       2^28-sample stream cut into one contiguous run of segments per rank (sweep.welch_long_stream: one all-gather of raw
       sums).  Both STRONG scaling, each with `ranks_seen`, a prefix parity number and a `scaling_base` naming the N = 1 key
       to divide by.  Default (the driver's line):
N > 1  workload "C4" (BASELINE config 4 = the north star's 8-segment sweep), STRONG scaling: a FIXED sweep of
       8 RF segments x 2^27 samples, segment i on rank i mod N, the same Welch parameters + fftshift + 256-bin
       trim + dB (spectrum_sweeper.py:260-276), then ONE all-gather (RCCL) of the 3584-bin rows into tune order
       (spectrum_sweeper.py:223).  It runs the shipped path, ofdm_tools.sweep.SweepPipeline (what
       spectrum_sweeper's sharded sweep is built on): device in, device out, the gather of sweep i overlapping
       the kernels of sweep i + 1, all gathers complete before the clock stops.  value = 8 * 2^27 samples
       per sweep / time per sweep.

Timing: W warm-up steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize() on both
sides; every step is also bracketed by events on the compute stream.  `ms_per_step` is the MEDIAN of the K
per-step event times (max over ranks), `value` follows from it; `wall_ms_per_step` is the bracketed wall time
/ K (max over ranks).  Before the warm-up the same step runs untimed for --ramp-ms so that the device has left
its idle clock level.

Launching.  Under a launcher (WORLD_SIZE set) the process is one rank.  Typed as `python bench.py --gpus N` with
N > 1 it spawns its N ranks itself - child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started
BEFORE this process makes any HIP call (a process that has touched the GPU is never re-executed) - waits for them,
forwards rank 0's JSON line and exits non-zero if any rank failed.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'gr-ofdm_tools_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

NFFT = 4096
LOG2_SAMPLES = 28
SWEEP_SEGMENTS = 8
SWEEP_LOG2_SAMPLES = 27
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
TONES = ((0.5, 0.1234), (0.05, -0.31), (2.0, 0.4071))
DC = 0.1 + 0.05j


def cpu_baseline(nfft):
    """The reference's CPU path (scipy.signal.welch on complex64, ofdm_cr_tools.py:342) on a
    bounded sample of the same workload, one process."""
    import numpy as np
    from oracle import ref_cpu as R
    n = 1 << 26
    parts = [R.synth_iq(1 << 22, 1002 + i, n0=i << 22) for i in range(n >> 22)]   # chunked: bounds host RAM
    x = np.concatenate(parts)
    del parts
    R.welch_reference_call(x[:1 << 20], nfft, 1.0)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        R.welch_reference_call(x, nfft, 1.0)
        times.append(time.perf_counter() - t0)
    t = sorted(times)[1]
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    single = {'value': n / t / 1e6, 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
              'sample': 'scipy.signal.welch(complex64, hann, nperseg=nfft=4096, 50%% overlap) on the first 2^26 '
                        'samples of the C2 recipe, median of 3, host has %d cores (%s)' % (os.cpu_count(), model)}
    # the same call split over threads (SciPy's FFT and NumPy's elementwise kernels release the GIL): contiguous
    # runs of segments with a 2048-sample halo, per-run mean x segment count summed - what a user of the
    # reference could do on this host without changing its arithmetic.  Informational, second object.  The
    # worker count is the CPU share this job may use (scheduler affinity, at most the 16 a one-GPU box grants),
    # not os.cpu_count(): the host's other cores belong to other jobs.
    from concurrent.futures import ThreadPoolExecutor
    try:
        share = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        share = os.cpu_count() or 1
    threads = max(1, min(16, share))
    nseg = (n - nfft // 2) // (nfft // 2)
    bounds = [nseg * i // threads for i in range(threads + 1)]

    def run(i):
        a, b = bounds[i], bounds[i + 1]
        if b <= a:
            return 0.0
        return R.welch_reference_call(x[a * (nfft // 2):(b - 1) * (nfft // 2) + nfft], nfft, 1.0) * (b - a)

    ptimes = []
    with ThreadPoolExecutor(threads) as pool:
        for _ in range(3):
            t0 = time.perf_counter()
            total = sum(pool.map(run, range(threads))) / nseg
            ptimes.append(time.perf_counter() - t0)
    ref = R.welch_reference_call(x, nfft, 1.0)
    dev = float(np.max(np.abs(total - ref) / ref))
    parallel = {'value': n / sorted(ptimes)[1] / 1e6, 'unit': 'Msamples/s', 'cores': threads, 'kind': 'port',
                'sample': 'same call and samples, %d threads (this job\'s CPU share; the host has %d cores) over '
                          'contiguous segment runs (2048-sample halo), median of 3, max rel deviation from the '
                          'one-thread result %.1e' % (threads, os.cpu_count() or 0, dev)}
    return single, parallel


def cpu_baseline_c5(nfft=16384, nch=4, log2_samples=22):
    """SURVEY 8d baseline (iii): the GNU Radio chain of config 5 restated on the CPU in the arithmetic GNU Radio uses
    (single precision: oracle.chain_sensor_v2_mean_c64 = stream_to_vector -> fft_vcc(rect, shift) -> |.|^2 -> 1/N^2,
    multichannel_scanner.py:78-86, averaged per channel stream), one thread, on a bounded sample."""
    from oracle import ref_cpu as R
    S = 1 << log2_samples
    xs = [R.synth_iq(S, 3000 + c) for c in range(nch)]
    R.chain_sensor_v2_mean_c64(xs[0][:1 << 18], nfft)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        for x in xs:
            R.chain_sensor_v2_mean_c64(x, nfft)
        times.append(time.perf_counter() - t0)
    return {'value': nch * S / sorted(times)[1] / 1e6, 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'sample': '%d of the 64 channel streams of scan_c5 (2^%d samples each), %d-pt rect FFT + |X|^2/N^2 + mean in '
                      'single precision (scipy.fft on complex64, one (rows, N) batch per stream), median of 3'
                      % (nch, log2_samples, nfft)}


def cpu_baseline_c1(nfft=1024, log2_samples=20):
    """BASELINE config 1 on the CPU, one core: the GNU Radio chain of spectrum_sensor_v2 restated in the arithmetic GNU Radio
    uses (single precision, scipy.fft on complex64: stream_to_vector -> fft_vcc(rect, shift) -> |.|^2 -> 1/N^2,
    spectrum_sensor_v2.py:85-93), the 128 eight-row means and per mean row the a7 channel sums (oracle.src_power: moving
    average + slice sums, ofdm_cr_tools.py:232-249) - the whole 2^20-sample configuration, median of 5."""
    import numpy as np
    import scipy.fft as sfft
    from oracle import ref_cpu as R
    n, Sf, cs, sbw = 1 << log2_samples, 1000000, 25e3, 12.5e3
    x = R.synth_iq(n, 1001)
    Fr = float(Sf) / nfft
    bb = R.frange(-Sf // 2, Sf // 2, cs)

    def run():
        v = x.reshape(-1, nfft)
        X = sfft.fft(v, axis=1)
        p = np.fft.fftshift((X.real * X.real + X.imag * X.imag) * np.float32(1.0 / float(nfft ** 2)), axes=1)
        mean8 = p.reshape(-1, 8, nfft).mean(axis=1)
        return [R.src_power(row, nfft, Fr, Sf, bb, sbw / Fr) for row in mean8]

    run()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    return {'value': n / sorted(times)[2] / 1e6, 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'sample': 'the whole C1 configuration: 2^%d samples, %d-pt rect FFT + |X|^2/N^2 in single precision (scipy.fft on '
                      'complex64), 128 eight-row means, 40 channel sums per mean row (Python, as the reference); median of 5'
                      % (log2_samples, nfft)}


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as child processes and forward rank
    0's line.  Nothing here touches the GPU: torch.cuda.device_count() reads the device list without creating a HIP
    context on this image, and the children are ordinary new processes (Popen), never an exec of this one."""
    import torch
    have = torch.cuda.device_count()
    rehearse = os.environ.get('BENCH_REHEARSE') == '1'
    if have == 0:
        sys.exit('bench.py needs an MI355X: torch.cuda.device_count() is 0 (there is no CPU fallback)')
    if have < args.gpus and not rehearse:
        sys.exit('bench.py --gpus %d: only %d GPU(s) visible (BENCH_REHEARSE=1 rehearses the N-rank path on one '
                 'device over gloo)' % (args.gpus, have))
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is read to its end first (it prints one line); a rank that dies takes the others' collectives
    # with it, so the first failure ends the rest (by their own PIDs)
    failed = None
    out0 = b''
    pending = set(range(args.gpus))
    import select
    fd0 = procs[0].stdout
    while pending:
        r, _, _ = select.select([fd0], [], [], 0.2) if fd0 else ([], [], [])
        if r:
            chunk = os.read(fd0.fileno(), 65536)
            if chunk:
                out0 += chunk
            else:
                fd0 = None
        for i in list(pending):
            rc = procs[i].poll()
            if rc is not None:
                pending.discard(i)
                if rc != 0 and failed is None:
                    failed = (i, rc)
        if failed is not None:
            for i in pending:
                procs[i].terminate()
            for i in pending:
                try:
                    procs[i].wait(10)
                except subprocess.TimeoutExpired:
                    procs[i].kill()
            break
    if procs[0].stdout:
        out0 += procs[0].stdout.read() or b''
    if failed is not None:
        sys.stderr.write('bench.py: rank %d exited with code %d\n' % failed)
        return failed[1] if failed[1] > 0 else 1
    lines = [ln for ln in out0.decode().splitlines() if ln.startswith('{')]
    if not lines:
        sys.stderr.write('bench.py: rank 0 printed no result line\n')
        return 1
    print(lines[-1])
    return 0


def timed_steps(torch, dist, dev, step, fence, steps, multi):
    """K steps between two fences; per-step events on the compute stream.  -> (wall seconds max over ranks,
    per-step ms list max-reduced over ranks)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i].record()
        step()
    ev[steps].record()
    fence()
    t1 = time.perf_counter()
    per = torch.tensor([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)], dtype=torch.float64, device=dev)
    wall = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(per, op=dist.ReduceOp.MAX)
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    return float(wall.item()), [float(v) for v in per.cpu()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--ramp-ms', type=float, default=150.0,
                    help='untimed device clock ramp before the warm-up steps (an idle MI355X sits at 775 MHz sclk)')
    ap.add_argument('--log2-samples', type=int, default=LOG2_SAMPLES, help='C2 stream length (N = 1)')
    ap.add_argument('--sweep-log2-samples', type=int, default=SWEEP_LOG2_SAMPLES,
                    help='samples per RF segment of the 8-segment sweep (N > 1, and the sweep_c4 key at N = 1)')
    ap.add_argument('--workload', choices=('c4', 'c5', 'c2'), default='c4',
                    help='N > 1 only: c4 = the 8-segment sweep (default; the driver\'s line), c5 = BASELINE config 5 with its 64 '
                         'channel rows over the ranks (BatchScanPlan.scan_sharded: one all-gather of rows + floors + powers), '
                         'c2 = the 2^28-sample stream cut into one run of segments per rank (sweep.welch_long_stream: one '
                         'all-gather of raw sums)')
    ap.add_argument('--scan-log2-samples', type=int, default=22, help='samples per channel stream of --workload c5')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='N = 1: skip the sweep_c4, csd_c3, scan_c5, host-visible and h2d_inclusive keys')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))                 # before anything in this process touches the GPU

    # the pool's host driver only supports dmabuf IPC; RCCL needs this before the HIP runtime starts
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    import torch.distributed as dist
    from ofdm_tools import _hip, scan_batch, sweep, windows

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)')
    # rehearsal on a one-GPU box: BENCH_REHEARSE=1 puts every rank on device 0 and uses gloo (RCCL refuses
    # two ranks on one device); the driver's real multi-GPU runs use one GPU per rank and nccl (= RCCL)
    rehearse = os.environ.get('BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # BENCH_FORCE_COLLECTIVE=1: run the N > 1 code path (RCCL init, all-gather, barriers) with whatever world
    # size the launcher gives, 1 included - a one-GPU box can then exercise the real nccl backend
    force = os.environ.get('BENCH_FORCE_COLLECTIVE') == '1'
    multi = world > 1 or force
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # a dedicated non-blocking stream, made torch's current stream: the library's kernels, torch's fills and the
    # event dependencies of the NCCL collectives all refer to it (the legacy null stream would serialise
    # against every blocking stream in the process)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = _hip.Context(local_rank, stream=stream.cuda_stream)
    assert ctx.on_torch_stream()
    hann = windows.get_window('hann', NFFT)

    def fence():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def ramp(fn):
        # clock ramp (setup, untimed, not a step count): the part idles at its lowest sclk level and needs some
        # tens of ms of load before it holds its sustained clock (local kernels only, no collective)
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(8):
                fn()
            torch.cuda.synchronize(dev)

    def profile_traffic(cfg, alg_bytes):
        """HBM bytes per launch of configuration `cfg` as the builder's rocprofv3 PMC passes measured them (FETCH_SIZE x 2
        + WRITE_SIZE per the guide), scaled to this run's launch size by the algorithmic bytes - NOT measured by this run."""
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get(cfg)
        except (OSError, ValueError, AttributeError):
            tj = None
        if not tj:
            return None, None
        ratio = tj['ratio_to_algorithmic']
        return ratio * alg_bytes, ('%s: %.3f x algorithmic (builder box; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                                   'tools/prof_driver.py %s, not measured by this run)' % (tj.get('source', 'profiles/'), ratio, cfg))

    def wflattop(n):
        return windows.get_window('flattop', n)

    # ---------------------------------------------------------------- the 8-segment sweep (C4) ----------
    def sweep_bench(steps, warmup, ref_call=False):
        """ref_call: the call spectrum_sweeper's stitcher actually makes (spectrum_sweeper.py:263: flattop, nperseg =
        fft_len / 4 zero-padded to fft_len, SciPy's default 50 % overlap of nperseg) instead of the Hann nperseg = 4096 form
        BASELINE's metric is quoted on - four transforms per 2048 new samples: VALU-bound by construction."""
        S = 1 << args.sweep_log2_samples
        trim = 256
        nbins = NFFT - 2 * trim
        mine = sweep.shard_segments(SWEEP_SEGMENTS, rank, world)
        seg = {}
        for i in mine:                                   # this rank's RF segments, resident in HBM
            seg[i] = torch.empty((S, 2), dtype=torch.float32, device=dev)
            ctx.synth_iq(seg[i].data_ptr(), S, 2000 + i, TONES, DC)
        pkw = dict(nperseg=NFFT // 4, window=wflattop(NFFT // 4)) if ref_call else dict(window=hann)
        plan = ctx.welch_plan(NFFT, fs=2.0e6, fftshift=True, trim_bins=trim, db=True, **pkw)
        pipe = sweep.SweepPipeline(SWEEP_SEGMENTS, nbins, dev, rank, world)

        def compute(i, out_row):
            plan.exec_dev(seg[i].data_ptr(), S, out_row.data_ptr())

        last = [0]

        def step():
            last[0] = pipe.run(compute)

        def full_fence():
            pipe.drain()
            fence()

        if mine:
            ramp(lambda: plan.exec_dev(seg[mine[0]].data_ptr(), S, pipe.local[0][0].data_ptr()))
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, full_fence, steps, multi)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        wide = pipe.wideband(last[0])
        assert int(wide.numel()) == SWEEP_SEGMENTS * nbins and bool(torch.isfinite(wide).all())
        # parity of this rank's first segment on a prefix, against the oracle (outside the timed region)
        err = None
        if rank == 0:
            from oracle import ref_cpu as R
            pre = seg[0][:1 << 20].cpu().numpy().view(np.complex64).reshape(-1)
            if ref_call:
                ref = 10 ** (R.sweeper_src_power(pre, NFFT, 2.0e6, trim) / 10)      # the reference's _src_power restated
            else:
                _, ref = R.welch_np(pre, fs=2.0e6, nperseg=NFFT, nfft=NFFT)
                ref = np.fft.fftshift(ref)[trim:-trim]
            chk = ctx.welch_plan(NFFT, fs=2.0e6, fftshift=True, trim_bins=trim, db=True, **pkw)
            err = float(np.max(np.abs(10 ** (chk.exec(pre).astype(np.float64) / 10) - ref) / ref))
        med = statistics.median(per)
        total = SWEEP_SEGMENTS * S
        kavg = kern_ms / max(launches, 1)
        ach = 8.0 * S / (kavg * 1e-3) / 1e9 if kavg else 0.0
        traffic, tsrc = profile_traffic('C4ref' if ref_call else 'C4', 8 * S)
        out = {'value': total / (med * 1e-3) / 1e6, 'unit': 'Msamples/s', 'ms_per_sweep': med,
               'wall_ms_per_sweep': 1e3 * wall / steps, 'segments': SWEEP_SEGMENTS, 'samples_per_segment': S,
               'segments_on_rank0': len(mine), 'steps': steps,
               'kernel_avg_ms': kavg, 'launches': int(launches),
               'config': {'workload': ('C4 as the reference block calls it: %d RF segments x 2^%d samples, flattop nperseg = 1024 '
                                       'zero-padded to nfft = 4096, step 512 (spectrum_sweeper.py:263), fftshift + 256-bin trim + dB'
                                       if ref_call else 'C4: %d RF segments x 2^%d samples, 4096-pt Hann Welch 50%% overlap, fftshift + '
                                       '256-bin trim + dB') % (SWEEP_SEGMENTS, args.sweep_log2_samples)},
               'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                            'frac': ach / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': tsrc,
                            'kernel': plan.last_recipe() + ' (one launch per RF segment)',
                            'kernel_avg_ms': kavg, 'launches': int(launches),
                            'algorithmic_bytes_per_launch': 8 * S,
                            'whole_sweep_frac': 8.0 * total / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS},
               'parity_prefix_max_rel_err': err}
        if ref_call:
            # Eight transforms per 4096 new samples: this workload cannot approach the byte roofline, its ceiling is VALU issue
            # (verdict r5).  achieved = VALU wave-instructions per second - the instruction count per launch is the builder's
            # PMC figure (profiles/traffic.json, scaled by the launch size; not measured by this run), the time is this run's;
            # peak = 1024 SIMDs x one wave-instruction per 2 cycles at the 2.4 GHz peak engine clock.  The HBM figure stays
            # beside it as `hbm`.
            try:
                tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))['C4ref']
                instr = tj['valu_wave_instructions_per_launch'] * (8.0 * S) / tj['algorithmic_bytes_per_launch']
                src = tj['valu_source']
            except (OSError, ValueError, KeyError):
                instr, src = None, None
            hbm = {k: out['roofline'][k] for k in ('achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'whole_sweep_frac')}
            peak = 1024 * 2.4e9 / 2 / 1e9
            ach_i = instr / (kavg * 1e-3) / 1e9 if (instr and kavg) else None
            out['roofline'].update({'bound': 'valu', 'achieved': ach_i, 'peak': peak, 'unit': 'G wave-instructions/s',
                                    'frac': ach_i / peak if ach_i else None, 'traffic': None,
                                    'instructions_per_launch': instr, 'instructions_source': src, 'hbm': hbm})
            for k in ('traffic_source', 'whole_sweep_frac'):
                out['roofline'].pop(k, None)
        del seg
        return out, S

    def ranks_seen():
        """Who took part, as the collective backend sees it: every rank's PCI address (domain, bus, device) and
        device index all-gathered through the process group (RCCL on a real run) - the distinct addresses must be
        as many as the ranks unless this is a one-device rehearsal."""
        pr = torch.cuda.get_device_properties(dev)
        mine = torch.tensor([rank, local_rank, int(getattr(pr, 'pci_domain_id', 0)), int(getattr(pr, 'pci_bus_id', -1)),
                             int(getattr(pr, 'pci_device_id', -1))], dtype=torch.int64,
                            device=dev if not rehearse else 'cpu')
        allr = torch.empty(world * 5, dtype=torch.int64, device=mine.device)
        if world > 1 or force:
            dist.all_gather_into_tensor(allr, mine)
        else:
            allr[:5] = mine
        rows = [[int(v) for v in r] for r in allr.view(world, 5).cpu()]
        return {'world_size': world, 'backend': dist.get_backend() if multi else None,
                'devices': ['%04x:%02x:%02x.0 (rank %d, cuda:%d)' % (r[2], r[3], r[4], r[0], r[1]) for r in rows],
                'distinct_devices': len({(r[2], r[3], r[4]) for r in rows})}

    # ---------------------------------------------------------------- two-channel csd / coherence (C3) ----
    def csd_bench(steps, warmup):
        n = 1 << 26
        x = torch.empty((n, 2), dtype=torch.float32, device=dev)
        y = torch.empty((n, 2), dtype=torch.float32, device=dev)
        ctx.synth_iq(x.data_ptr(), n, 1003, TONES, DC)
        ctx.synth_iq(y.data_ptr(), n, 1004, (), 0j)
        y.mul_(0.5).add_(torch.roll(x, 5, 0), alpha=0.7)      # y = 0.7 x delayed by 5 samples + independent noise
        plan = ctx.welch_plan(NFFT, window=hann, fs=1.0)
        pxx, pyy, cxy = (torch.zeros(NFFT, dtype=torch.float32, device=dev) for _ in range(3))
        pxy = torch.zeros((NFFT, 2), dtype=torch.float32, device=dev)

        def step():
            plan.csd_exec_dev(x.data_ptr(), y.data_ptr(), n, pxx.data_ptr(), pyy.data_ptr(), pxy.data_ptr(), cxy.data_ptr())

        ramp(step)
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, steps, False)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        assert bool(torch.isfinite(cxy).all())
        from oracle import ref_cpu as R
        m = 1 << 20
        px = x[:m].cpu().numpy().view(np.complex64).reshape(-1)
        py = y[:m].cpu().numpy().view(np.complex64).reshape(-1)
        _, rc, rxx, ryy, rxy = R.coherence_np(px, py, nperseg=NFFT, nfft=NFFT)
        gxx, gyy, gxy, gc = ctx.welch_plan(NFFT, window=hann, fs=1.0).csd(px, py)
        err = {'pxx': float(np.max(np.abs(gxx - rxx) / rxx)), 'pyy': float(np.max(np.abs(gyy - ryy) / ryy)),
               'pxy_over_sqrt_pxx_pyy': float(np.max(np.abs(gxy - rxy) / np.sqrt(rxx * ryy))),
               'cxy_abs': float(np.max(np.abs(gc - rc)))}
        med = statistics.median(per)
        kavg = kern_ms / max(launches, 1)
        ach = 16.0 * n / (kavg * 1e-3) / 1e9 if kavg else 0.0
        traffic, tsrc = profile_traffic('C3', 16 * n)
        return {'value': n / (med * 1e-3) / 1e6, 'unit': 'Msample-pairs/s', 'ms_per_step': med,
                'wall_ms_per_step': 1e3 * wall / steps, 'steps': steps, 'kernel_avg_ms': kavg, 'launches': int(launches),
                'config': {'workload': 'C3: two complex64 streams x 2^26 samples (y = 0.7 x delayed by 5 + noise), 4096-pt '
                                       'Hann, 50% overlap, detrend constant -> Pxx, Pyy, Pxy, Cxy on the device'},
                'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': ach / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': tsrc,
                             'kernel': plan.last_recipe(),
                             'kernel_avg_ms': kavg, 'launches': int(launches), 'algorithmic_bytes_per_launch': 16 * n,
                             'whole_step_frac': 16.0 * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                'parity_prefix_max_err': err}

    # ---------------------------------------------------------------- batched scanner (C5) -----------------
    def scan_bench(steps, warmup):
        nch, S, N = 64, 1 << 22, 16384
        Sf = 1000000
        iq = torch.empty((nch * S, 2), dtype=torch.float32, device=dev)
        for c in range(nch):
            ctx.synth_iq(iq.data_ptr() + 8 * c * S, S, 3000 + c, TONES, DC)
        bp = scan_batch.BatchScanPlan(ctx, N, Sf, 15625.0, 10e3, thr_leveler=10)
        lo, hi = bp._slices()
        rows = torch.zeros((nch, N), dtype=torch.float32, device=dev)
        noise = torch.zeros(nch, dtype=torch.float32, device=dev)
        power = torch.zeros((nch, max(len(lo), 1)), dtype=torch.float32, device=dev)
        mask = torch.zeros((nch, N), dtype=torch.uint8, device=dev)

        def step():
            bp.psd_rows_dev(iq.data_ptr(), S, nch, S, rows.data_ptr())
            ctx.scan_decide_dev_out(rows.data_ptr(), nch, N, bp.scanner.srch_bins, bp.thr_leveler, lo, hi,
                                    noise.data_ptr(), power.data_ptr(), mask.data_ptr())

        ramp(step)
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, steps, False)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        # parity of channel 0 (prefix of 2^20 samples: PSD row, noise floor, mask, channel sums) against the oracle
        from oracle import ref_cpu as R
        m = 1 << 20
        pre = iq[:m].cpu().numpy().view(np.complex64).reshape(-1)
        d_pre, d_row = ctx.alloc(m * 8), ctx.alloc(N * 4)
        try:
            ctx.h2d(d_pre, pre)
            bp.psd_rows_dev(d_pre, m, 1, m, d_row)
            got_mask, got_noise, got_pw = bp.decide_dev(d_row, 1)
            got_row = ctx.d2h(d_row, (N,), np.float32)
        finally:
            ctx.free(d_pre)
            ctx.free(d_row)
        ref_row = R.chain_sensor_v2(pre, N).mean(axis=0)
        ma = R.movingaverage(ref_row.astype(np.float32), bp.scanner.srch_bins)
        ref_noise = float(ma.min())
        ref_pw = np.array(R.src_power(ref_row.astype(np.float32), N, bp.scanner.Fr, Sf, bp.scanner.bb_freqs,
                                      bp.scanner.srch_bins))
        if bp.scanner.trunc > 0:
            ref_pw = ref_pw[bp.scanner.trunc_ch:-bp.scanner.trunc_ch]
        ref_mask = ref_row > bp.thr_leveler * ref_noise
        margin = np.abs(ref_row - bp.thr_leveler * ref_noise) / (bp.thr_leveler * ref_noise)
        err = {'psd_row': float(np.max(np.abs(got_row - ref_row) / ref_row)),
               'noise_floor': abs(float(got_noise[0]) - ref_noise) / ref_noise,
               'channel_power': float(np.max(np.abs(np.asarray(got_pw)[0] - ref_pw) / ref_pw)),
               'mask_mismatches_beyond_1e-4_of_the_threshold': int(np.sum(((got_mask[0] != 0) != ref_mask) & (margin >= 1e-4)))}
        med = statistics.median(per)
        kavg = kern_ms / max(launches, 1)
        total = nch * S
        ach = 8.0 * total / (kavg * 1e-3) / 1e9 if kavg else 0.0
        traffic, tsrc = profile_traffic('C5', 8 * total)
        return {'value': total / (med * 1e-3) / 1e6, 'unit': 'Msamples/s', 'ms_per_step': med,
                'wall_ms_per_step': 1e3 * wall / steps, 'steps': steps, 'kernel_avg_ms': kavg, 'launches': int(launches),
                'config': {'workload': 'C5: 64 channel streams x 2^22 samples, 16384-pt rectangular |X|^2/N^2 mean per stream '
                                       '(multichannel_scanner.py:78-86) + moving average, noise floor, per-bin mask and %d '
                                       'channel sums on the device (oth_scan_decide_dev_out)' % len(lo)},
                'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': ach / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': tsrc,
                             'kernel': bp.plan.last_recipe(),
                             'kernel_note': 'kernel_avg_ms brackets the transform launch only; finalize_l4 and the two '
                                            'decision-stage launches are in ms_per_step / whole_step_frac',
                             'kernel_avg_ms': kavg, 'launches': int(launches), 'algorithmic_bytes_per_launch': 8 * total,
                             'whole_step_frac': 8.0 * total / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                'parity_prefix_max_rel_err': err}

    # ---------------------------------------------------------------- N > 1: config 5 and the long stream over the ranks ----
    def scan_sharded_bench(steps, warmup):
        """SURVEY 8e row 3 (multichannel_scanner.py:78-100 batched over ranks): channel stream c on rank c mod N, PSD rows +
        decision stage local, ONE all-gather of rows + noise floors + channel powers per step.  STRONG scaling: 64 streams."""
        nch, S, N = 64, 1 << args.scan_log2_samples, 16384
        Sf = 1000000
        mine = scan_batch.shard_channels(nch, rank, world)
        iq = torch.empty((max(len(mine), 1) * S, 2), dtype=torch.float32, device=dev)
        for j, c in enumerate(mine):
            ctx.synth_iq(iq.data_ptr() + 8 * j * S, S, 3000 + c, TONES, DC)
        bp = scan_batch.BatchScanPlan(ctx, N, Sf, 15625.0, 10e3, thr_leveler=10)
        got = [None]

        def step():
            got[0] = bp.scan_sharded(iq.data_ptr(), S, S, nch, rank, world, dev)

        if mine:
            rows0 = torch.zeros((len(mine), N), dtype=torch.float32, device=dev)
            ramp(lambda: bp.psd_rows_dev(iq.data_ptr(), S, len(mine), S, rows0.data_ptr()))
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, steps, multi)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        rows, noise, power = got[0]
        assert tuple(rows.shape) == (nch, N) and bool(torch.isfinite(rows).all()) and bool((noise > 0).all())
        err = None
        if rank == 0:      # channel 0 (this rank's first stream), prefix of 2^20 samples, against the oracle
            from oracle import ref_cpu as R
            m = min(S, 1 << 20)
            pre = iq[:m].cpu().numpy().view(np.complex64).reshape(-1)
            d_pre, d_row = ctx.alloc(m * 8), ctx.alloc(N * 4)
            try:
                ctx.h2d(d_pre, pre)
                bp.psd_rows_dev(d_pre, m, 1, m, d_row)
                got_row = ctx.d2h(d_row, (N,), np.float32)
            finally:
                ctx.free(d_pre)
                ctx.free(d_row)
            ref_row = R.chain_sensor_v2(pre, N).mean(axis=0)
            err = float(np.max(np.abs(got_row - ref_row) / ref_row))
        med = statistics.median(per)
        kavg = kern_ms / max(launches, 1)
        local_bytes = 8.0 * len(mine) * S
        ach = local_bytes / (kavg * 1e-3) / 1e9 if kavg else 0.0
        return {'value': nch * S / (med * 1e-3) / 1e6, 'ms': med, 'wall_ms': 1e3 * wall / steps, 'kavg': kavg, 'launches': int(launches),
                'ach': ach, 'alg_bytes': local_bytes, 'kernel': bp.plan.last_recipe(), 'err': err,
                'workload': 'C5 over %d ranks: 64 channel streams x 2^%d samples, stream c on rank c mod %d, 16384-pt rectangular '
                            '|X|^2/N^2 mean + decision stage per rank, all_gather of rows + noise floors + channel powers '
                            '(BatchScanPlan.scan_sharded)' % (world, args.scan_log2_samples, world),
                'parallelism': 'channel-stream-per-gpu x%d' % world,
                'scaling_base': 'scan_c5.value of the --gpus 1 line (the same 64 streams on one GPU; same stream length when '
                                '--scan-log2-samples is 22)'}

    def long_stream_bench(steps, warmup):
        """ONE stream of 2^log2 samples cut into contiguous runs of segments, one per rank (run g holds its
        neighbour's first 2048 samples again), raw sums per rank, ONE all-gather, rank-order sum, scaling.  STRONG scaling."""
        n = 1 << args.log2_samples
        first, nloc, _, nseg_local = sweep.time_shard(n, NFFT, NFFT // 2, rank, world)
        iq = torch.empty((max(nloc, NFFT), 2), dtype=torch.float32, device=dev)
        ctx.synth_iq(iq.data_ptr(), max(nloc, NFFT), 1002 + rank, TONES, DC)      # (every run its own noise: the result is a PSD of the whole)
        plan = ctx.welch_plan(NFFT, window=hann, fs=1.0)
        got = [None]

        def step():
            got[0] = sweep.welch_long_stream(plan, iq.data_ptr(), first, n, dev, rank, world)

        scratch = torch.zeros(NFFT, dtype=torch.float32, device=dev)
        if nseg_local:
            ramp(lambda: plan.partial_dev(iq.data_ptr(), nloc, scratch.data_ptr()))
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, steps, multi)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        psd, nseg_total = got[0]
        assert nseg_total == plan.nseg(n) and bool(torch.isfinite(psd).all()) and int(psd.numel()) == NFFT
        err = None
        if rank == 0:
            from oracle import ref_cpu as R
            pre = iq[:1 << 20].cpu().numpy().view(np.complex64).reshape(-1)
            _, ref = R.welch_np(pre, fs=1.0, nperseg=NFFT, nfft=NFFT)
            err = float(np.max(np.abs(ctx.welch_plan(NFFT, window=hann, fs=1.0).exec(pre) - ref) / ref))
        med = statistics.median(per)
        kavg = kern_ms / max(launches, 1)
        local_bytes = 8.0 * nseg_local * (NFFT // 2)
        ach = local_bytes / (kavg * 1e-3) / 1e9 if kavg else 0.0
        return {'value': n / (med * 1e-3) / 1e6, 'ms': med, 'wall_ms': 1e3 * wall / steps, 'kavg': kavg, 'launches': int(launches),
                'ach': ach, 'alg_bytes': local_bytes, 'kernel': plan.last_recipe(), 'err': err,
                'workload': 'C2 over %d ranks: one 2^%d-sample stream, contiguous runs of ceil(nseg / %d) segments per rank with a '
                            '2048-sample halo, 4096-pt Hann Welch 50%% overlap, all_gather of the raw |X|^2 sums, rank-order sum, '
                            'density scaling (sweep.welch_long_stream)' % (world, args.log2_samples, world),
                'parallelism': 'time-shard-per-gpu x%d' % world,
                'scaling_base': '`value` of the --gpus 1 line (the same stream on one GPU)'}

    # ---------------------------------------------------------------- config 1 at its own size ---------------
    def c1_bench(steps, warmup):
        """SURVEY 8d C1 (examples/spectrum_sensor_test.grc: samp_rate 1e6, channel spacing 25 kHz, search bandwidth 12.5 kHz):
        2^20 samples -> 1024 vectors of 1024 points (rect, shifted, |X|^2 / N^2) -> 128 eight-row means -> moving average +
        40 channel sums per mean row, all on the device: the 8-vector runs are the streams of ONE Welch launch without overlap
        (the batched scanner's form), the a7 stage is oth_scan_decide_dev_out.  A 8 MiB workload: launch-bound, not
        bandwidth-bound - `frac` says how far."""
        n, N, Sf, cs, sbw = 1 << 20, 1024, 1000000, 25e3, 12.5e3
        iq = torch.empty((n, 2), dtype=torch.float32, device=dev)
        ctx.synth_iq(iq.data_ptr(), n, 1001, TONES, DC)
        bp = scan_batch.BatchScanPlan(ctx, N, Sf, cs, sbw, thr_leveler=10)
        lo, hi = bp._slices()
        rows = torch.zeros((128, N), dtype=torch.float32, device=dev)
        noise = torch.zeros(128, dtype=torch.float32, device=dev)
        power = torch.zeros((128, len(lo)), dtype=torch.float32, device=dev)

        def step():
            bp.psd_rows_dev(iq.data_ptr(), 8192, 128, 8192, rows.data_ptr())
            ctx.scan_decide_dev_out(rows.data_ptr(), 128, N, bp.scanner.srch_bins, bp.thr_leveler, lo, hi, noise.data_ptr(),
                                    power.data_ptr())

        ramp(step)
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, steps, False)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        from oracle import ref_cpu as R
        x = iq.cpu().numpy().view(np.complex64).reshape(-1)
        ref_mean = R.chain_sensor_v2(x, N).reshape(128, 8, N).mean(axis=1)
        got_rows, got_pw = rows.cpu().numpy(), power.cpu().numpy()
        ref_pw = np.array([R.src_power(r.astype(np.float32), N, bp.scanner.Fr, Sf, bp.scanner.bb_freqs, bp.scanner.srch_bins)
                           for r in ref_mean])
        err = {'mean_rows': float(np.max(np.abs(got_rows - ref_mean) / ref_mean)),
               'channel_power': float(np.max(np.abs(got_pw - ref_pw) / ref_pw))}
        med = statistics.median(per)
        kavg = kern_ms / max(launches, 1)
        ach = 8.0 * n / (kavg * 1e-3) / 1e9 if kavg else 0.0
        return {'value': n / (med * 1e-3) / 1e6, 'unit': 'Msamples/s', 'ms_per_step': med, 'wall_ms_per_step': 1e3 * wall / steps,
                'steps': steps, 'kernel_avg_ms': kavg, 'launches': int(launches),
                'config': {'workload': 'C1 at its own size: 2^20 complex64 samples, 1024-pt rectangular |fftshift(FFT)|^2 / N^2, 1024 '
                                       'vectors -> 128 eight-row means (one launch, 128 streams of 8 vectors) + moving average and %d '
                                       'channel sums per mean row on the device (Sf 1e6, 25 kHz / 12.5 kHz)' % len(lo)},
                'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
                             'traffic': None, 'kernel': bp.plan.last_recipe(), 'kernel_avg_ms': kavg, 'launches': int(launches),
                             'algorithmic_bytes_per_launch': 8 * n,
                             'whole_step_frac': 8.0 * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             'note': '8 MiB per launch: a handful of microseconds of HBM time - the step is launch latency'},
                'parity_max_rel_err': err}

    # ---------------------------------------------------------------- the lengths above the tuned kernels ----
    def big_welch_bench(nfft, steps, warmup):
        """Round 6: Hann Welch, 50 % overlap, detrend constant at 32768 / 65536 points - what fast_spectrum_scan(n_fft=0) picks for
        blocks of 16 Ki ... 64 Ki samples (ofdm_cr_tools.py:474-475) and a flowgraph reaches with --nfft - on 2^27 resident samples.
        One launch: the segment inside one workgroup (32768) or a pair of workgroups (65536: even / odd bins), csrc/welch32k.hip.
        `kernel_avg_ms` is the HIP-event time of one call's launches together, `achieved` = 8 B x samples / that."""
        n = 1 << 27
        iq = torch.empty((n, 2), dtype=torch.float32, device=dev)
        ctx.synth_iq(iq.data_ptr(), n, 1002, TONES, DC)
        plan = ctx.welch_plan(nfft, window=windows.get_window('hann', nfft), fs=1.0)
        out = torch.zeros(nfft, dtype=torch.float32, device=dev)

        def step():
            plan.exec_dev(iq.data_ptr(), n, out.data_ptr())

        ramp(step)
        for _ in range(warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, steps, False)
        kern_ms, calls = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        from oracle import ref_cpu as R
        pre = iq[:1 << 21].cpu().numpy().view(np.complex64).reshape(-1)
        _, ref = R.welch_np(pre, fs=1.0, nperseg=nfft, nfft=nfft)
        err = float(np.max(np.abs(ctx.welch_plan(nfft, window=windows.get_window('hann', nfft), fs=1.0).exec(pre) - ref) / ref))
        med = statistics.median(per)
        kavg = kern_ms / max(calls, 1)
        ach = 8.0 * n / (kavg * 1e-3) / 1e9 if kavg else 0.0
        traffic, tsrc = profile_traffic('w%d' % nfft, 8 * n)
        plan_recipe[nfft] = plan.last_recipe() + (' (sub-block sums + K1 + K2 per workspace chunk, summed)' if ':r16' in plan.last_recipe() else '')
        plan.close()
        return {'value': n / (med * 1e-3) / 1e6, 'unit': 'Msamples/s', 'ms_per_step': med, 'wall_ms_per_step': 1e3 * wall / steps,
                'steps': steps, 'kernel_avg_ms': kavg, 'calls': int(calls),
                'config': {'workload': '2^27-sample complex64 stream, %d-pt Hann Welch, 50%% overlap, detrend constant, density '
                                       '(%s)' % (nfft, 'the 256 KiB segment stays in one workgroup' if nfft == 32768 else
                                                 'the 512 KiB segment goes to a pair of workgroups: even / odd bins')},
                'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
                             'traffic': traffic, 'traffic_source': tsrc, 'kernel': plan_recipe[nfft],
                             'kernel_avg_ms': kavg, 'launches': int(calls), 'algorithmic_bytes_per_launch': 8 * n,
                             'whole_step_frac': 8.0 * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                'parity_prefix_max_rel_err': err}

    plan_recipe = {}

    result = None
    if multi and args.workload != 'c4':
        seen = ranks_seen()
        r = (scan_sharded_bench if args.workload == 'c5' else long_stream_bench)(args.steps, args.warmup)
        if rank == 0:
            result = {
                'metric': 'IQ Msamples/s Welch-PSD (4096-pt, 50% ovlp)' if args.workload == 'c2' else
                          'IQ Msamples/s batched scanner (64 x 16384-pt PSD + per-bin threshold)',
                'value': r['value'], 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': r['ms'], 'wall_ms_per_step': r['wall_ms'], 'higher_is_better': True, 'scaling': 'strong',
                'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': r['workload'], 'parallelism': r['parallelism']},
                'roofline': {'bound': 'hbm', 'achieved': r['ach'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': r['ach'] / HBM_PEAK_GBPS, 'traffic': None, 'kernel': r['kernel'] + ' (rank 0\'s launch)',
                             'kernel_avg_ms': r['kavg'], 'launches': r['launches'],
                             'algorithmic_bytes_per_launch': r['alg_bytes']},
                'parity_prefix_max_rel_err': r['err'], 'ranks_seen': seen, 'scaling_base': r['scaling_base'],
                'cpu_baseline': None, 'device': ctx.device_name(),
            }
    elif multi:
        seen = ranks_seen()
        sw, S = sweep_bench(args.steps, args.warmup)
        if rank == 0:
            kavg = sw['kernel_avg_ms']
            achieved = 8.0 * S / (kavg * 1e-3) / 1e9 if kavg else 0.0
            result = {
                'metric': 'IQ Msamples/s Welch-PSD (4096-pt, 50% ovlp)',
                'value': sw['value'], 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': sw['ms_per_sweep'], 'wall_ms_per_step': sw['wall_ms_per_sweep'],
                'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
                'data': 'synthetic',
                'config': {'workload': 'C4: fixed sweep of %d RF segments x 2^%d samples, segment i on rank i mod %d, '
                                       '4096-pt Hann Welch 50%% overlap, fftshift + 256-bin trim + dB, all_gather of '
                                       '3584-bin rows into tune order (ofdm_tools.sweep.SweepPipeline)'
                                       % (SWEEP_SEGMENTS, args.sweep_log2_samples, world),
                           'nfft': NFFT, 'noverlap': NFFT // 2, 'window': 'hann', 'segments': SWEEP_SEGMENTS,
                           'samples_per_segment': S, 'parallelism': 'segment-per-gpu x%d' % world},
                'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                             'frac': achieved / HBM_PEAK_GBPS, 'traffic': sw['roofline']['traffic'],
                             'traffic_source': sw['roofline']['traffic_source'], 'kernel': sw['roofline']['kernel'],
                             'kernel_avg_ms': kavg, 'launches': sw['launches'],
                             'algorithmic_bytes_per_launch': 8 * S},
                'parity_prefix_max_rel_err': sw['parity_prefix_max_rel_err'],
                'ranks_seen': seen,
                'scaling_base': 'sweep_c4.value of the --gpus 1 line (the same fixed sweep on one GPU): speed-up at N = '
                                'this value / that value',
                'cpu_baseline': None, 'device': ctx.device_name(),
            }
    else:
        n = 1 << args.log2_samples
        iq = torch.empty((n, 2), dtype=torch.float32, device=dev)               # the IQ ring buffer in HBM
        ctx.synth_iq(iq.data_ptr(), n, 1002, TONES, DC)
        plan = ctx.welch_plan(NFFT, window=hann, fs=1.0)
        nseg = plan.nseg(n)
        out = [torch.zeros(NFFT, dtype=torch.float32, device=dev) for _ in range(2)]
        count = [0]

        def step():
            count[0] += 1
            plan.exec_dev(iq.data_ptr(), n, out[count[0] & 1].data_ptr())

        ramp(step)
        for _ in range(args.warmup):
            step()
        ctx.set_timing(True)
        ctx.get_timing(reset=True)
        wall, per = timed_steps(torch, dist, dev, step, fence, args.steps, False)
        kern_ms, launches = ctx.get_timing(reset=True)
        ctx.set_timing(False)
        psd = out[count[0] & 1]
        assert bool(torch.isfinite(psd).all())
        med = statistics.median(per)
        kavg_ms = kern_ms / max(launches, 1)
        achieved = 8.0 * n / (kavg_ms * 1e-3) / 1e9
        # HBM bytes per launch: NOT measured by this run (PMC counters need rocprofv3 around the process) - the
        # figure of the builder's last profile of the same kernel, labelled as such
        traffic, traffic_source = profile_traffic('C2', 8 * n)
        # sanity / parity on a prefix, outside the timed region
        from oracle import ref_cpu as R
        pre = iq[:1 << 20].cpu().numpy().view(np.complex64).reshape(-1)
        chk = ctx.welch_plan(NFFT, window=hann, fs=1.0)
        _, ref = R.welch_np(pre, fs=1.0, nperseg=NFFT, nfft=NFFT)
        err = float(np.max(np.abs(chk.exec(pre) - ref) / ref))
        probe_ms = ctx.stream_read_probe(iq.data_ptr(), n * 8, 5)
        result = {
            'metric': 'IQ Msamples/s Welch-PSD (4096-pt, 50% ovlp)',
            'value': n / (med * 1e-3) / 1e6, 'unit': 'Msamples/s', 'n_gpus': 1, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': med, 'wall_ms_per_step': 1e3 * wall / args.steps,
            'ms_per_step_min_max': [min(per), max(per)],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C2: 2^%d-sample complex64 stream, 4096-pt Hann Welch, 50%% overlap, detrend '
                                   'constant, density' % args.log2_samples,
                       'nfft': NFFT, 'noverlap': NFFT // 2, 'window': 'hann', 'samples_per_gpu': n,
                       'segments_per_gpu': nseg, 'parallelism': 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': plan.last_recipe(),
                         'kernel_note': 'one launch per step: the pilot of the constant detrend is formed in this kernel\'s '
                                        'prologue (round 5), so kernel_avg_ms covers it; finalize_wide_kernel is in ms_per_step',
                         'kernel_avg_ms': kavg_ms, 'launches': int(launches),
                         'algorithmic_bytes_per_launch': 8 * n,
                         'read_probe_GBps': 8.0 * n / (probe_ms * 1e-3) / 1e9},
            'parity_prefix_max_rel_err': err,
            'device': ctx.device_name(),
        }
        if not args.no_extras:
            # SURVEY 8d ends the metric at "PSD available on host": the same step through the host-output entry point
            # (oth_welch_exec with a device source: kernels + 16 KiB D2H + stream synchronisation inside the call)
            # (the prefix parity check above ran on the CPU for a few hundred ms: the device has dropped to its idle clock
            # and needs the same untimed ramp as every other section - without it this loop read 30-40 us per step more)
            ramp(step)
            for _ in range(3):
                plan.exec_device_src(iq.data_ptr(), n)
            hv = []
            for _ in range(max(10, args.steps // 4)):
                t0 = time.perf_counter()
                plan.exec_device_src(iq.data_ptr(), n)
                hv.append((time.perf_counter() - t0) * 1e3)
            result['host_visible_ms_per_step'] = statistics.median(hv)
            result['host_visible_value'] = n / (statistics.median(hv) * 1e-3) / 1e6
            result['host_visible_note'] = ('oth_welch_exec(src_is_device=1): launch -> float32[4096] PSD in host memory, '
                                           'one blocking call per step (no overlap between steps; the finalize launch writes '
                                           'the row and a completion word into pinned memory, the call polls the word), median '
                                           'of %d host-clock times; `value` is the pipelined device-side rate' % len(hv))
            # host buffer -> PSD on the host through the streaming entry point (pinned staging ring, asynchronous
            # H2D + kernels): the PCIe-inclusive rate; never `value`
            # sixteen DISTINCT 32 MiB chunks (512 MiB of host memory: well past the CPU's last-level cache, so the
            # source of every copy comes from DRAM as a recorded stream's would)
            m, chunk = 1 << 26, 1 << 22
            rng = np.random.default_rng(5)
            hosts = []
            for i in range(m // chunk):
                h = np.empty(chunk, np.complex64)
                h.real = rng.standard_normal(chunk, dtype=np.float32)
                h.imag = 0.25 + i
                hosts.append(h)
            sp = ctx.welch_plan(NFFT, window=hann, fs=1.0)
            for h in hosts[:4]:
                sp.accumulate(h)
            sp.finalize()
            t0 = time.perf_counter()
            for h in hosts:
                sp.accumulate(h)
            sp.finalize()
            dt = time.perf_counter() - t0
            del hosts
            result['h2d_inclusive'] = {'value': m / dt / 1e6, 'unit': 'Msamples/s', 'GBps': 8.0 * m / dt / 1e9,
                                       'sample': '2^26 samples from sixteen distinct pageable host buffers of 2^22 samples '
                                                 'through oth_welch_accumulate (asynchronous: chunks above 1 MiB use the '
                                                 'runtime\'s staged copy from pageable memory, smaller ones a pinned '
                                                 'ring) + oth_welch_finalize'}
            del iq
            torch.cuda.empty_cache()
            sw, _ = sweep_bench(max(10, args.steps // 4), max(3, args.warmup // 4))
            result['sweep_c4'] = sw
            torch.cuda.empty_cache()
            swr, _ = sweep_bench(max(5, args.steps // 10), max(2, args.warmup // 10), ref_call=True)
            result['sweep_c4_ref'] = swr
            torch.cuda.empty_cache()
            result['csd_c3'] = csd_bench(max(10, args.steps // 4), max(3, args.warmup // 4))
            torch.cuda.empty_cache()
            result['scan_c5'] = scan_bench(max(10, args.steps // 4), max(3, args.warmup // 4))
            torch.cuda.empty_cache()
            result['c1'] = c1_bench(max(10, args.steps // 4), max(3, args.warmup // 4))
            torch.cuda.empty_cache()
            for nf in (32768, 65536):
                result['welch_%d' % nf] = big_welch_bench(nf, max(5, args.steps // 10), max(2, args.warmup // 10))
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            result['cpu_baseline'], result['cpu_baseline_parallel'] = cpu_baseline(NFFT)
            result['cpu_baseline_c5'] = cpu_baseline_c5()
            result['cpu_baseline_c1'] = cpu_baseline_c1()
        else:
            result['cpu_baseline'] = None
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
