"""The multi-GPU path on the one GPU a test box has: the RCCL reduce of the HIP
mix bus in a world of one rank (backend "nccl" is RCCL: communicator set-up, the stream
hand-over between the library's launches and the collective, the root's division), and one
131,072-voice shard of BASELINE configs[4] (1,048,576 voices over 8 GPUs) at full size."""
import socket

import numpy as np
import pytest

from klang_amd import patches as P
from klang_amd.sharding import ShardedMixer, shard_range
from tests.helpers import B, hip_bank_from_patch, library_table, oracle_bank_from_patch, report

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_rccl_reduce_of_the_hip_bus_in_a_world_of_one(hip):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0,
                            world_size=1, device_id=torch.device('cuda', 0))
    try:
        n_total, nb = 8192, 24
        first, count = shard_range(n_total, 1, 0)
        patch = P.polyphonic_saw_patch(n_voices=n_total, n_blocks=nb, off_base=6, off_span=10,
                                       first_voice=first, count=count)
        for side in (False, True):
            stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
            with torch.cuda.stream(stream):
                ctx = hip.Context(0, stream.cuda_stream)
                want = hip_bank_from_patch(hip, ctx, patch).render(nb)
                bank = hip_bank_from_patch(hip, ctx, patch)
                mixer = ShardedMixer(bank, n_total, nb, world_size=1, rank=0, always_reduce=True)
                assert mixer.collective
                host = torch.zeros((nb, 1, B), dtype=torch.float32).pin_memory()
                bus = mixer.render()          # launches, ncclReduce, division: no sync in between
                host.copy_(bus, non_blocking=True)
                stream.synchronize()
                got = host.numpy()
                assert np.abs(want).max() > 1e-3
                assert np.abs(got - want).max() <= 1e-7   # float32 division on the root
                ctx.close()
    finally:
        dist.destroy_process_group()


def test_config5_shard_of_131072_voices(hip, oracle):
    """Rank 3's shard of the 1,048,576-voice patch (voices 393,216 .. 524,287), as bench.py
    --gpus 8 renders it: un-normalised bus (divisor 1).  Against the threaded oracle, and the
    size-independent properties: shard-sum invariance (its two halves add up to it),
    launch-split invariance, linearity in the gains."""
    n_total, world, rank, nb = 1048576, 8, 3, 6
    first, count = shard_range(n_total, world, rank)
    assert (first, count) == (393216, 131072)
    kw = dict(n_voices=n_total, n_blocks=nb, off_base=2, off_span=3)
    patch = P.polyphonic_saw_patch(first_voice=first, count=count, **kw)
    ctx = hip.Context(0)
    shard = hip_bank_from_patch(hip, ctx, patch)
    shard.set_mix_divisor(1.)
    bus32 = shard.render(nb)
    bus = bus32.astype(np.float64)
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    want, _ = ob.render(nb, patch.events, n_threads=min(8, oracle.max_threads()))
    err = np.abs(bus / count - want).max()      # the oracle bank divides by its voice count
    report('chain osc+ADSR+biquad+mixer (131,072-voice shard of 1,048,576)', err)
    assert np.abs(want).max() > 1e-4 and err <= 1e-6
    np.testing.assert_array_equal(shard.get_state(hip.F_PHASE), ob.phase)
    acc = np.zeros_like(bus)
    for half in (0, 1):
        hp = P.polyphonic_saw_patch(first_voice=first + half * count // 2, count=count // 2, **kw)
        hb = hip_bank_from_patch(hip, ctx, hp)
        hb.set_mix_divisor(1.)
        acc += hb.render(nb)
        hb.close()
    assert np.abs(acc - bus).max() <= 1e-6 * count      # buses are un-normalised sums here
    again = hip_bank_from_patch(hip, ctx, patch)
    again.set_mix_divisor(1.)
    np.testing.assert_array_equal(np.concatenate([again.render(2), again.render(4)]), bus32)
    dbl = hip_bank_from_patch(hip, ctx, patch)
    dbl.set_mix_divisor(1.)
    dbl.set(hip.F_GAIN, 2. * patch.gain)
    np.testing.assert_array_equal(dbl.render(nb), 2 * bus32)
    ctx.close()
