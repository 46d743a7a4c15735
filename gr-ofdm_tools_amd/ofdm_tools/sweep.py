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
