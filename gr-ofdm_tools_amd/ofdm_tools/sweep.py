"""Sharding of a frequency sweep over GPUs (one process per GPU).

spectrum_stitcher.run (python/spectrum_sweeper.py:207-231) walks the tune
frequencies one after another and concatenates the per-segment PSDs (:223).  The
segments are independent until that concatenate, so segment i goes to rank
i mod world, every rank runs its segments through the HIP Welch plan, and one
all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) reassembles the
wideband PSD in tune order on every rank.
"""
import torch
import torch.distributed as dist


def shard_segments(nseg_total, rank, world):
    """Indices (in tune order) of the segments rank `rank` owns."""
    return list(range(rank, nseg_total, world))


def segments_per_rank(nseg_total, world):
    return (nseg_total + world - 1) // world


def gather_wideband(local_psd, nseg_total, rank=None, world=None, group=None, out=None):
    """local_psd: [segments_per_rank, nbins] tensor (rows past this rank's share are padding).
    Returns the [nseg_total * nbins] wideband PSD in tune order, identical on every rank."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    spr = segments_per_rank(nseg_total, world)
    assert local_psd.dim() == 2 and local_psd.shape[0] == spr, (tuple(local_psd.shape), spr)
    nbins = local_psd.shape[1]
    if world == 1:
        return local_psd[:nseg_total].reshape(-1)
    if out is None:
        out = torch.empty((world * spr, nbins), dtype=local_psd.dtype, device=local_psd.device)
    dist.all_gather_into_tensor(out, local_psd.contiguous(), group=group)
    # row r*spr + j is segment r + world*j  ->  tune order
    return out.view(world, spr, nbins).permute(1, 0, 2).reshape(spr * world, nbins)[:nseg_total].reshape(-1)


def sweep_psd(segment_iq, compute_psd, nseg_total, nbins, device, rank, world, group=None):
    """Run this rank's segments and gather.  segment_iq(i) -> IQ of tune index i (any form
    compute_psd accepts); compute_psd(iq, out_row) writes nbins float32 into out_row."""
    spr = segments_per_rank(nseg_total, world)
    local = torch.zeros((spr, nbins), dtype=torch.float32, device=device)
    for j, i in enumerate(shard_segments(nseg_total, rank, world)):
        compute_psd(segment_iq(i), local[j])
    return gather_wideband(local, nseg_total, rank, world, group)


# --------------------------------------------------------------------------------------------
# Long-stream sharding: segment sums are associative, so a stream is cut
# into contiguous runs of segments, one per rank; run g needs samples
# [s0 * step, (s1 - 1) * step + nperseg) - i.e. its neighbour's first nperseg - step samples again (the
# halo).  Every rank reduces its run to raw sums (oth_welch_partial_dev: sum |X|^2, nfft floats;
# oth_csd_partial_dev: sum |X|^2, sum |Y|^2, sum conj(X) Y, 4 * nfft floats), ONE all-gather moves the
# partials + segment counts, every rank adds them in rank order (bit-identical result everywhere) and
# applies the scaling for the total segment count (oth_welch_scale_dev / oth_csd_scale_dev).
# --------------------------------------------------------------------------------------------

def time_shard(nsamples, nperseg, step, rank, world):
    """-> (first_sample, nsamples_local, first_segment, nseg_local) of rank's contiguous run of segments
    (ceil(nseg / world) segments per rank; trailing ranks may get none)."""
    nseg = (nsamples - (nperseg - step)) // step if nsamples >= nperseg else 0
    per = -(-nseg // world) if nseg else 0
    s0 = min(rank * per, nseg)
    s1 = min(s0 + per, nseg)
    if s1 <= s0:
        return s0 * step, 0, s0, 0
    return s0 * step, (s1 - s0 - 1) * step + nperseg, s0, s1 - s0


def reduce_partials(local_sums, nseg_local, rank=None, world=None, group=None, nseg_total=None):
    """local_sums: 1-D float32 tensor of raw sums (zeros when this rank has no segment).  Returns
    (sums float32 tensor, nseg_total) - the same bits on every rank (fixed summation order).
    When the caller knows the total segment count (time_shard() makes it a function of the stream length alone)
    it passes nseg_total: nothing but the sums travels and nothing comes back to the host.  Otherwise the counts
    ride along in two floats each and are read on the host AFTER the sums have been formed on the device."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_sums, int(nseg_local if nseg_total is None else nseg_total)
    n = local_sums.numel()
    extra = 0 if nseg_total is not None else 2
    packed = local_sums
    if extra:
        packed = torch.empty(n + 2, dtype=torch.float32, device=local_sums.device)
        packed[:n] = local_sums
        packed[n] = float(int(nseg_local) >> 16)          # the count travels exactly in two floats
        packed[n + 1] = float(int(nseg_local) & 0xFFFF)
    out = torch.empty(world * (n + extra), dtype=torch.float32, device=local_sums.device)
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    out = out.view(world, n + extra)
    total = torch.zeros(n, dtype=torch.float64, device=local_sums.device)
    for r in range(world):                            # rank order: deterministic, identical everywhere
        total += out[r, :n].to(torch.float64)
    sums = total.to(torch.float32)                    # enqueued before any host read below
    if nseg_total is not None:
        return sums, int(nseg_total)
    counts = out[:, n:].to(torch.float64).cpu()
    nseg = int(sum(int(counts[r, 0]) * 65536 + int(counts[r, 1]) for r in range(world)))
    return sums, nseg


def welch_time_sharded(partial, scale, nsamples, nperseg, step, nsums, device, rank, world, group=None):
    """One long stream over `world` ranks.  partial(first_sample, n, out) reduces samples
    [first_sample, first_sample + n) to raw sums in the float32 tensor `out` (nsums floats) and returns
    its segment count; scale(sums, nseg_total) turns the summed partials into the result."""
    first, n, _, nseg_local = time_shard(nsamples, nperseg, step, rank, world)
    nseg_total = (nsamples - (nperseg - step)) // step if nsamples >= nperseg else 0      # every rank can count them
    local = torch.zeros(nsums, dtype=torch.float32, device=device)
    got = partial(first, n, local) if nseg_local else 0
    assert got == nseg_local, (got, nseg_local)
    sums, nseg = reduce_partials(local, nseg_local, rank, world, group, nseg_total=nseg_total)
    return scale(sums, nseg), nseg


def gather_rows(local_rows, nrows_total, rank=None, world=None, group=None):
    """Batched scanner (8e row 3): channel stream c lives on rank c mod world; -> rows in channel order."""
    nbins = local_rows.shape[1]
    return gather_wideband(local_rows, nrows_total, rank, world, group).view(nrows_total, nbins)


# ---- the same two paths on HIP plans (device pointers; ofdm_tools._hip.WelchPlan) ---------------------------
# Stream ordering.  The library launches on its context's stream, torch (fills, casts, collectives) on torch's current
# stream.  A context created on torch's stream (Context(dev, stream=...), what bench.py does) needs nothing.  With a
# context on its own stream every hand-over is synchronised in BOTH directions: before a library call that reads or
# writes memory torch has produced (torch.zeros, the summed partials, a caller's tensor) torch's stream is drained,
# and after it the context's stream is, before torch touches the result.

def torch_then_ctx(ctx, device):
    """Call before a library call whose operands torch's current stream may still be writing."""
    if not ctx.on_torch_stream() and torch.device(device).type == 'cuda':
        torch.cuda.current_stream(device).synchronize()


def ctx_then_torch(ctx):
    """Call after a library call whose results torch's current stream will read."""
    if not ctx.on_torch_stream():
        ctx.sync()


def welch_long_stream(plan, local_dptr, local_first_sample, nsamples_total, device, rank, world, group=None):
    """Welch PSD of one stream of nsamples_total samples spread over the ranks in time order.  This rank's HBM
    buffer at local_dptr starts at stream sample local_first_sample and must cover its run + halo
    (time_shard()).  -> (psd float32 tensor [plan.out_len] on `device`, nseg_total), identical on every rank."""
    def partial(first, n, out):
        torch_then_ctx(plan.ctx, device)     # torch.zeros(out) has landed
        nseg = plan.partial_dev(local_dptr + 8 * (first - local_first_sample), n, out.data_ptr())
        ctx_then_torch(plan.ctx)             # the collective runs on torch's stream
        return nseg

    def scale(sums, nseg):
        out = torch.empty(plan.out_len, dtype=torch.float32, device=device)
        sums = sums.contiguous()
        torch_then_ctx(plan.ctx, device)     # the rank-order sum and its cast have landed
        plan.scale_dev(sums.data_ptr(), nseg, out.data_ptr())
        ctx_then_torch(plan.ctx)
        return out
    return welch_time_sharded(partial, scale, nsamples_total, plan.nperseg, plan.step, plan.nfft, device, rank, world,
                              group)


def csd_long_stream(plan, x_dptr, y_dptr, local_first_sample, nsamples_total, device, rank, world, group=None):
    """Two-channel cross spectrum / coherence of one long stream pair over the ranks (partials: sum |X|^2,
    sum |Y|^2, sum conj(X) Y = 4 * nfft floats per rank).  -> ((pxx, pyy, pxy [out_len][2], cxy), nseg_total)."""
    def partial(first, n, out):
        off = 8 * (first - local_first_sample)
        torch_then_ctx(plan.ctx, device)
        nseg = plan.csd_partial_dev(x_dptr + off, y_dptr + off, n, out.data_ptr())
        ctx_then_torch(plan.ctx)
        return nseg

    def scale(sums, nseg):
        m = plan.out_len
        pxx, pyy, cxy = (torch.empty(m, dtype=torch.float32, device=device) for _ in range(3))
        pxy = torch.empty((m, 2), dtype=torch.float32, device=device)
        sums = sums.contiguous()
        torch_then_ctx(plan.ctx, device)
        plan.csd_scale_dev(sums.data_ptr(), nseg, pxx.data_ptr(), pyy.data_ptr(), pxy.data_ptr(), cxy.data_ptr())
        ctx_then_torch(plan.ctx)
        return pxx, pyy, pxy, cxy
    return welch_time_sharded(partial, scale, nsamples_total, plan.nperseg, plan.step, 4 * plan.nfft, device, rank,
                              world, group)


class SweepPipeline(object):
    """Continuous sweeping (spectrum_stitcher.run loops forever, python/spectrum_sweeper.py:207-231): the
    all-gather of sweep i is issued asynchronously and overlaps the kernels of sweep i + 1 on `depth` sets of
    row buffers.  ``run(compute)`` computes this rank's segments (``compute(i, out_row)`` writes nbins float32
    of tune index i into the device row, asynchronously on torch's current stream) and starts the gather;
    ``wideband(slot)`` waits for it and returns the stitched PSD in tune order."""

    def __init__(self, nseg_total, nbins, device, rank, world, group=None, depth=2):
        self.nseg_total, self.nbins, self.rank, self.world, self.group = nseg_total, nbins, rank, world, group
        self.mine = shard_segments(nseg_total, rank, world)
        self.spr = segments_per_rank(nseg_total, world)
        self.local = [torch.zeros((self.spr, nbins), dtype=torch.float32, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((world * self.spr, nbins), dtype=torch.float32, device=device)
                         for _ in range(depth)] if world > 1 else None
        self.pending = [None] * depth
        self.count = 0

    def run(self, compute):
        slot = self.count % len(self.local)
        self.count += 1
        if self.pending[slot] is not None:
            self.pending[slot].wait()            # the sweep that used these buffers has been gathered
            self.pending[slot] = None
        for j, i in enumerate(self.mine):
            compute(i, self.local[slot][j])
        if self.world > 1:
            self.pending[slot] = dist.all_gather_into_tensor(self.gathered[slot], self.local[slot], group=self.group,
                                                             async_op=True)
        return slot

    def wideband(self, slot):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        if self.world == 1:
            return self.local[slot][:self.nseg_total].reshape(-1)
        g = self.gathered[slot]
        return g.view(self.world, self.spr, self.nbins).permute(1, 0, 2).reshape(-1, self.nbins)[:self.nseg_total] \
            .reshape(-1)

    def drain(self):
        for s in range(len(self.pending)):
            if self.pending[s] is not None:
                self.pending[s].wait()
                self.pending[s] = None
