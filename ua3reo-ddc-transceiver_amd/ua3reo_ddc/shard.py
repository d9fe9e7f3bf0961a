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


class StepGather:
    """All-gather of every step's decimated output, double buffered so that the gather of
    step k runs beside the compute of step k + 1 (equal shards: the weak-scaling layout).

    The compute side writes step k into buffer(k) = outs[k & 1]; submit(k) then gathers
    that buffer from every rank into gathered[k & 1], laid out [world][...local shape].
    On a GPU the gather runs on a stream of its own, ordered by events against the
    stream the compute is queued on (compute_stream, a torch stream); buffer(k) makes
    that stream wait for the gather that last read the buffer it hands out.  On the CPU
    (gloo) everything is synchronous.  The payload travels as bytes."""

    def __init__(self, dist, local_shape, dtype, device, compute_stream=None, timed=False):
        self.dist = dist
        self.world = dist.get_world_size()
        self.cuda = torch.device(device).type == "cuda"
        self.outs = torch.zeros((2,) + tuple(local_shape), dtype=dtype, device=device)
        self.gathered = torch.zeros((2, self.world) + tuple(local_shape), dtype=dtype, device=device)
        self.compute_stream = compute_stream
        self.times = [] if timed else None
        self.steps_submitted = 0
        if self.cuda:
            self.comm = torch.cuda.Stream(device=device)
            self.computed = [torch.cuda.Event(), torch.cuda.Event()]
            self.done = [torch.cuda.Event(), torch.cuda.Event()]
            self.pending = [False, False]

    def buffer(self, step):
        """The buffer step `step` computes into; call before queueing that step's compute."""
        b = step & 1
        if self.cuda and self.pending[b]:
            self.compute_stream.wait_event(self.done[b])
        return self.outs[b]

    def submit(self, step):
        """Queue the gather of step `step`'s buffer (after the compute queued so far)."""
        b = step & 1
        # concatenated along dim 0: the one output layout every backend accepts
        src = self.outs[b].view(torch.uint8)
        dst = self.gathered[b].view((self.world * self.outs.shape[1],) + tuple(self.outs.shape[2:])).view(torch.uint8)
        if not self.cuda:
            self.dist.all_gather_into_tensor(dst, src)
            self.steps_submitted += 1
            return
        self.computed[b].record(self.compute_stream)
        self.comm.wait_event(self.computed[b])
        with torch.cuda.stream(self.comm):
            if self.times is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.comm)
            self.dist.all_gather_into_tensor(dst, src)
            if self.times is not None:
                t1.record(self.comm)
                self.times.append((t0, t1))
            self.done[b].record(self.comm)
        self.pending[b] = True
        self.steps_submitted += 1

    def result(self, step):
        """[world][...local shape] of step `step` (one of the last two submitted); waits for its gather."""
        b = step & 1
        if self.cuda:
            self.done[b].synchronize()
        return self.gathered[b]

    def finish(self):
        if self.cuda:
            self.comm.synchronize()

    def average_ms(self):
        if not self.times:
            return None
        return sum(a.elapsed_time(b) for a, b in self.times) / len(self.times)
