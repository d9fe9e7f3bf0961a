"""Channel sharding over GPUs: one process per GPU, contiguous channel ranges, no
exchange inside the chain, one all-gather of the decimated I/Q per step.

Channels are independent (own NCO phase, own filter state), so the only
collective is the gather of the [SPEC|VOICE] int16 I/Q that every rank produced
for its own channels.  The functions here hold the host logic shared by the
benchmark and the tests; they work with any torch.distributed backend (RCCL on
the GPUs, gloo on CPU)."""
import torch


def channel_range(rank, world, total_channels):
    """Contiguous [lo, hi) of the channels rank owns; the first total % world
    ranks take one channel more."""
    if not (0 <= rank < world) or total_channels < 0:
        raise ValueError("bad rank/world/total")
    base, extra = divmod(total_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(world, total_channels):
    return [channel_range(r, world, total_channels)[1] - channel_range(r, world, total_channels)[0]
            for r in range(world)]


def gather_decimated(dist, local, total_channels, out=None):
    """All-gather per-rank outputs laid out [2 (SPEC|VOICE)][channels_local][frames][2]
    into [2][total_channels][frames][2], channel-major in rank order.

    Equal shards use one all_gather_into_tensor on a [world][...] staging tensor;
    ragged shards pad every rank to the largest shard first."""
    world = dist.get_world_size()
    sizes = shard_sizes(world, total_channels)
    if local.shape[1] != sizes[dist.get_rank()]:
        raise ValueError("local shard holds %d channels, expected %d" % (local.shape[1], sizes[dist.get_rank()]))
    big = max(sizes)
    if big != local.shape[1]:
        pad = torch.zeros((local.shape[0], big - local.shape[1]) + tuple(local.shape[2:]), dtype=local.dtype,
                          device=local.device)
        send = torch.cat([local, pad], dim=1)
    else:
        send = local.contiguous()
    # The payload travels as bytes (gloo has no int16) and concatenated along
    # dim 0, the one output layout every backend accepts.
    flat = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat.view(torch.uint8), send.view(torch.uint8))
    stage = flat.view((world,) + tuple(send.shape))
    if out is None:
        out = torch.empty((local.shape[0], total_channels) + tuple(local.shape[2:]), dtype=local.dtype,
                          device=local.device)
    lo = 0
    for r, n in enumerate(sizes):
        out[:, lo:lo + n] = stage[r][:, :n]
        lo += n
    return out
