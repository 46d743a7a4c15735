#!/usr/bin/env python3
"""bench.py - IQ Msamples/s through the 4096-point Welch PSD on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic IQ that is already
resident in HBM: the radix-16 welch4096 kernel + the cross-workgroup finalize
kernel (scale, fftshift, trim); at N > 1 also the all-gather (RCCL) that
reassembles the wideband PSD of the sweep on every rank (issued asynchronously so that it
overlaps the next sweep's kernel; all gathers are complete before the clock stops).

N = 1  workload "C2": one 2^28-sample complex64 stream (2 GiB), Hann, nperseg =
       nfft = 4096, 50 % overlap, detrend constant, density scaling (BASELINE
       config 2 = the welch() call of ofdm_cr_tools.py:342).
N > 1  workload "C4-weak": one 2^28-sample RF segment PER RANK with the same Welch
       parameters + fftshift + 256-bin trim (spectrum_sweeper.py:260-276), then
       all_gather of the 3584-bin rows in tune order (spectrum_sweeper.py:223).
       Weak scaling: per-GPU work is fixed.

Before the W warm-up steps the same step runs untimed for --ramp-ms (default 150 ms) so that
the device has left its idle clock level; the K timed steps are bracketed by barrier +
torch.cuda.synchronize() on both sides as the contract says.

Prints ONE JSON line on rank 0.  `value` = samples processed by all ranks / the
max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'gr-ofdm_tools_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

NFFT = 4096
LOG2_SAMPLES = 28
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
    # reference could do on this host without changing its arithmetic.  Informational, second object.
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, os.cpu_count() or 1))
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
                'sample': 'same call and samples, %d threads over contiguous segment runs (2048-sample halo), '
                          'median of 3, max rel deviation from the one-thread result %.1e' % (threads, dev)}
    return single, parallel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--ramp-ms', type=float, default=150.0,
                    help='untimed device clock ramp before the warm-up steps (an idle MI355X sits at 775 MHz sclk)')
    ap.add_argument('--log2-samples', type=int, default=LOG2_SAMPLES)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    # the pool's host driver only supports dmabuf IPC; RCCL needs this before the HIP runtime starts
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    import torch.distributed as dist
    from ofdm_tools import _hip, windows

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
    if world > 1 or force:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if rehearse:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    n = 1 << args.log2_samples
    multi = world > 1 or force
    trim = 256 if multi else 0
    nbins = NFFT - 2 * trim

    # a dedicated non-blocking stream, made torch's current stream: the library's kernels, torch's fills and the
    # event dependencies of the NCCL collectives all refer to it (the legacy null stream would serialise
    # against every blocking stream in the process)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = _hip.Context(local_rank, stream=stream.cuda_stream)
    iq = torch.empty((n, 2), dtype=torch.float32, device=dev)               # the IQ ring buffer in HBM
    ctx.synth_iq(iq.data_ptr(), n, (2000 + rank) if multi else 1002, TONES, DC)
    plan = ctx.welch_plan(NFFT, window=windows.get_window('hann', NFFT), fs=1.0, fftshift=multi, trim_bins=trim)
    nseg = plan.nseg(n)
    # two sets of buffers: the all-gather of sweep i overlaps the Welch kernel of sweep i+1
    local = [torch.zeros((1, nbins), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world, nbins), dtype=torch.float32, device=dev) for _ in range(2)] if multi else None
    pending = [None, None]
    count = [0]

    def step():
        i = count[0] & 1
        count[0] += 1
        if multi and pending[i] is not None:
            pending[i].wait()                     # sweep i-2 has been gathered: its buffers are free again
            pending[i] = None
        plan.exec_dev(iq.data_ptr(), n, local[i].data_ptr())
        if multi:
            pending[i] = dist.all_gather_into_tensor(gathered[i], local[i], async_op=True)
            return gathered[i]
        return local[i][0]

    def drain():
        for i in range(2):
            if multi and pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    def fence():
        drain()
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # clock ramp (setup, untimed, not a step count): the part idles at its lowest sclk level and needs
    # some tens of ms of load before it holds its sustained clock; then the W warm-up steps
    # (time-bounded, so every rank runs its own number of iterations: local kernels only, no collective)
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(8):
            plan.exec_dev(iq.data_ptr(), n, local[0].data_ptr())
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_timing(True)
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wide = step()
    fence()
    t1 = time.perf_counter()
    kern_ms, launches = ctx.get_timing(reset=True)
    ctx.set_timing(False)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * n / (elapsed / args.steps) / 1e6
        kavg_ms = kern_ms / max(launches, 1)
        achieved = 8.0 * n / (kavg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get('log2_samples') == args.log2_samples:
                    traffic = tj.get('hbm_bytes_per_launch')
            except (OSError, ValueError):
                pass
        # sanity / parity on a prefix, outside the timed region
        from oracle import ref_cpu as R
        pre = iq[:1 << 20].cpu().numpy().view(np.complex64).reshape(-1)
        chk = ctx.welch_plan(NFFT, window=windows.get_window('hann', NFFT), fs=1.0)
        _, ref = R.welch_np(pre, fs=1.0, nperseg=NFFT, nfft=NFFT)
        err = float(np.max(np.abs(chk.exec(pre) - ref) / ref))
        probe_ms = ctx.stream_read_probe(iq.data_ptr(), n * 8, 5)
        result = {
            'metric': 'IQ Msamples/s Welch-PSD (4096-pt, 50% ovlp)',
            'value': value, 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('C4-weak: one 2^%d-sample RF segment per GPU, 4096-pt Hann Welch 50%% overlap, '
                                    'fftshift + 256-bin trim, all_gather of %d-bin rows' % (args.log2_samples, nbins))
                       if multi else
                       ('C2: 2^%d-sample complex64 stream, 4096-pt Hann Welch, 50%% overlap, detrend constant, '
                        'density' % args.log2_samples),
                       'nfft': NFFT, 'noverlap': NFFT // 2, 'window': 'hann', 'samples_per_gpu': n,
                       'segments_per_gpu': nseg, 'parallelism': 'segment-per-gpu x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'kernel': 'welch4096ws_kernel', 'kernel_avg_ms': kavg_ms, 'launches': int(launches),
                         'algorithmic_bytes_per_launch': 8 * n,
                         'read_probe_GBps': 8.0 * n / (probe_ms * 1e-3) / 1e9},
            'parity_prefix_max_rel_err': err,
            'device': ctx.device_name(),
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'], result['cpu_baseline_parallel'] = cpu_baseline(NFFT)
        else:
            result['cpu_baseline'] = None
        assert int(wide.numel()) == world * nbins
        assert bool(torch.isfinite(wide).all())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
