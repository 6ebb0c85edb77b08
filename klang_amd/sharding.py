"""Voice sharding across GPUs: one process per GPU, one collective per render.

Voices are independent until the Mixer sum (klang/audio/mixer.py:41-49), so a
patch of N voices is cut into contiguous voice ranges, one per rank; a voice's
whole chain (oscillator, envelope, filter, reverb rings) lives on one GPU and
time is never split.  Each rank renders the UN-normalised partial bus
sum_i gain_i * x_i of its shard; one reduce (RCCL over xGMI, backend "nccl")
sums the partial buses onto the root, which applies the Mixer's division by the
global input count.  Payload: n_blocks x channels x 1 KiB float32 per render --
latency-bound, hence one call per render, never per block.
"""
import collections
import os

import torch
import torch.distributed as dist

BUFFER_SIZE = 256

# What one rank keeps of one mixer: `first`..`first+count` of its `n_total` inputs.
Shard = collections.namedtuple('Shard', 'rank world n_total first count group')


def resolve(shard=None):
    """run_klang's `shard` argument -> (rank, world, group) or None.  `shard` may be a
    (rank, world) pair, a (rank, world, group) triple, True ("whatever torch.distributed
    says") or None: then the process group decides if one is initialised, and a lone
    WORLD_SIZE > 1 in the environment (a launcher started us, nobody initialised the group
    yet) initialises it -- backend "nccl" (RCCL) on LOCAL_RANK's GPU."""
    if shard is None or shard is True:
        if not dist.is_available():
            return None
        if not dist.is_initialized():
            if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
                return None
            local = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        if dist.get_world_size() == 1 and shard is None:
            return None
        return dist.get_rank(), dist.get_world_size(), None
    shard = tuple(shard)
    rank, world = int(shard[0]), int(shard[1])
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    return rank, world, (shard[2] if len(shard) > 2 else None)


def plan(n_inputs, rank, world, group=None):
    """The Shard of a mixer with n_inputs uniform chains for `rank`."""
    first, count = shard_range(n_inputs, world, rank)
    return Shard(rank, world, int(n_inputs), first, count, group)


def shard_range(n_total, world_size, rank):
    """Contiguous voice range [first, first+count) of `rank`; sizes differ by at
    most one voice."""
    if not 0 <= rank < world_size:
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    base, extra = divmod(int(n_total), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def owner_of(voice, n_total, world_size):
    """Rank that owns global voice index `voice` (events are routed on the host)."""
    base, extra = divmod(int(n_total), int(world_size))
    edge = extra * (base + 1)
    if voice < edge:
        return voice // (base + 1)
    return extra + (voice - edge) // base


def reduce_bus(bus, n_total, world_size, dst=0, group=None, always=False):
    """Sum the per-rank partial buses onto `dst` and apply mixer.py:46-47
    (divide by nInputs when nInputs > 1).  Works on CUDA tensors (RCCL) and CPU
    tensors (gloo).  Returns the bus (meaningful on `dst` only).  `always` issues the
    collective in a world of one rank too (the same call path on a one-GPU box)."""
    if world_size > 1 or always:
        if bus.is_cuda and dist.get_backend(group) == 'gloo':
            # a gloo group over GPU buses (several ranks sharing ONE GPU, where RCCL refuses to
            # run: rehearsals and tests): the reduce goes through host copies
            host = bus.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            bus.copy_(host)
        else:
            dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if n_total > 1:
        bus /= float(n_total)
    return bus


class ShardedMixer:
    """Renders this rank's bank and reduces the bus.  With world_size == 1 the
    bank's own divisor is used and no collective is issued, unless `always_reduce` asks for
    the multi-rank path (un-normalised bus, reduce, division on the root) in a world of one."""

    def __init__(self, bank, n_total, n_blocks, world_size=1, rank=0, group=None,
                 always_reduce=False):
        self.bank = bank
        self.n_total = int(n_total)
        self.n_blocks = int(n_blocks)
        self.world_size = int(world_size)
        self.rank = rank
        self.group = group
        self.collective = self.world_size > 1 or bool(always_reduce)
        self.bus = torch.zeros((self.n_blocks, bank.channels, BUFFER_SIZE), dtype=torch.float32,
                               device='cuda')
        if self.collective:
            bank.set_mix_divisor(1.)
        else:
            bank.set_mix_divisor(float(self.n_total) if self.n_total > 1 else 1.)

    def render(self):
        # launches are ordered on the torch stream the context was created with
        self.bank.render_device(self.n_blocks, self.bus.data_ptr())
        if self.collective:
            reduce_bus(self.bus, self.n_total, self.world_size, 0, self.group, always=True)
        return self.bus
