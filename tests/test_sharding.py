"""Voice sharding on CPU: shard arithmetic, and the bus reduce over
a world_size-2 gloo group.  Each rank's partial bus comes from the CPU oracle
here (test infrastructure); on GPUs the same reduce_bus() runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from klang_amd import patches as P
from klang_amd.sharding import owner_of, reduce_bus, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_voices():
    for n, world in ((65536, 8), (1048576, 8), (10, 3), (7, 8), (1, 1)):
        spans = [shard_range(n, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
        for v in {0, n // 2, n - 1}:
            r = owner_of(v, n, world)
            assert spans[r][0] <= v < spans[r][0] + spans[r][1]
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, n_blocks, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import oracle
    from tests.helpers import oracle_bank_from_patch
    first, count = shard_range(n_total, world, rank)
    patch = P.polyphonic_saw_patch(n_voices=n_total, n_blocks=n_blocks, off_base=2, off_span=4,
                                   first_voice=first, count=count)
    table = oracle.coefficient_table()
    bank = oracle_bank_from_patch(oracle, patch, table, oracle.filter_grid())
    mix, _ = bank.render(n_blocks, patch.events)
    if count > 1:
        mix = mix * count                          # undo the bank's own division: partial SUM
    bus = torch.from_numpy(mix.astype(np.float32))
    reduce_bus(bus, n_total, world, dst=0)
    if rank == 0:
        np.save(out_path, bus.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_bank(tmp_path, oracle):
    n_total, n_blocks = 96, 10
    out = str(tmp_path / 'bus.npy')
    mp.spawn(_worker, args=(2, _free_port(), n_total, n_blocks, out), nprocs=2, join=True)
    got = np.load(out)
    from tests.helpers import oracle_bank_from_patch
    patch = P.polyphonic_saw_patch(n_voices=n_total, n_blocks=n_blocks, off_base=2, off_span=4)
    bank = oracle_bank_from_patch(oracle, patch, oracle.coefficient_table(),
                                  oracle.filter_grid())
    want, _ = bank.render(n_blocks, patch.events)
    assert np.abs(got - want).max() <= 1e-6       # shard-sum invariance


def test_shards_of_a_patch_are_slices_of_the_whole():
    whole = P.polyphonic_saw_patch(n_voices=1000, n_blocks=8)
    first, count = shard_range(1000, 8, 3)
    part = P.polyphonic_saw_patch(n_voices=1000, n_blocks=8, first_voice=first, count=count)
    np.testing.assert_array_equal(part.frequency, whole.frequency[first:first + count])
    np.testing.assert_array_equal(part.cutoff, whole.cutoff[first:first + count])
    assert part.events['voice'].max() == count - 1
