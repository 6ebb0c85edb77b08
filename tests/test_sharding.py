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


# ---- the plan behind run_klang(shard=...) (fusion.shard_groups): which chains and notes every
# ---- rank keeps.  Evaluated with the explicit host blocks here (no GPU on this box).
def _api_patch(n_voices):
    """Voice(saw, ADSR) | Filter -> Mixer with per-input gains, a note per voice queued."""
    from klang_amd.audio import ADSR, Filter, Mixer, Oscillator, Voice, sawtooth
    from klang_amd.messages import Note
    mixer = Mixer()
    voices = []
    for k in range(n_voices):
        voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.005, .02, .6, .03))
        voice.input.push(Note(60 + k, .3 + .05 * (k % 7), frequency=110. * (1 + .37 * k)))
        mixer += voice | Filter(frequency=500. + 300. * k)
        voices.append(voice)
    mixer.gains = [1. - .04 * k for k in range(n_voices)]
    return mixer, voices


def _host_shard_bus(rank, world, n_voices, n_blocks):
    """Un-normalised partial bus of `rank`: only the kept chains' blocks are evaluated."""
    from klang_amd import fusion
    from klang_amd.messages import Note
    from klang_amd.schedule import determine_execution_order
    mixer, voices = _api_patch(n_voices)
    order = fusion.open_composites(determine_execution_order([mixer]))
    groups, errors = fusion.find_fusable(order)
    assert not errors and len(groups) == 1
    (sink, kept, synths, sh, dropped), = fusion.shard_groups(groups, rank, world)
    assert sink is mixer and (sh.rank, sh.world, sh.n_total) == (rank, world, n_voices)
    assert (sh.first, sh.count) == shard_range(n_voices, world, rank)
    mine = {b for c in kept for b in c.blocks}
    theirs = {b for c in dropped for b in c.blocks}
    assert not mine & theirs and len(kept) + len(dropped) == n_voices
    assert [c.voice for c in kept] == voices[sh.first:sh.first + sh.count]
    bus = np.zeros((n_blocks, 256))
    for k in range(n_blocks):
        if k == 5:          # every rank sees every note; a rank plays the ones of its voices
            for v in voices:
                v.input.push(Note(0, 0., frequency=0.))        # note-off
        for blk in order:
            if blk in mine:
                blk.update()
        for i, c in enumerate(kept):
            bus[k] += mixer.gains[sh.first + i] * c.end.output.value
    return bus, [v for v in voices if v.input.queue], kept


def test_shard_plan_keeps_contiguous_chains_and_their_notes():
    n = 10
    kept_all = []
    for rank in range(3):
        bus, unread, kept = _host_shard_bus(rank, 3, n, 7)
        # the notes of the voices this rank does not play were never consumed by its blocks
        assert len(unread) == n - len(kept)
        kept_all += [c.voice.input.queue for c in kept]
    assert len(kept_all) == n


def _plan_worker(rank, world, port, n_voices, n_blocks, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    bus, _, _ = _host_shard_bus(rank, world, n_voices, n_blocks)
    t = torch.from_numpy(bus.astype(np.float32))
    reduce_bus(t, n_voices, world, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_plan_sums_to_the_single_mixer(tmp_path):
    from klang_amd.schedule import determine_execution_order
    from klang_amd.messages import Note
    n, nb = 11, 12
    out = str(tmp_path / 'bus.npy')
    mp.spawn(_plan_worker, args=(2, _free_port(), n, nb, out), nprocs=2, join=True)
    got = np.load(out)
    mixer, voices = _api_patch(n)
    order = determine_execution_order([mixer])
    want = np.zeros((nb, 256))
    for k in range(nb):
        if k == 5:
            for v in voices:
                v.input.push(Note(0, 0., frequency=0.))
        for blk in order:
            blk.update()
        want[k] = mixer.output.value
    assert np.abs(want).max() > 1e-2
    assert np.abs(got - want).max() <= 1e-6


def test_groups_that_do_not_shard_stay_whole():
    from klang_amd import fusion
    from klang_amd.audio import Dac, Oscillator, Reverb
    from klang_amd.schedule import determine_execution_order
    mixer, _ = _api_patch(3)
    order = fusion.open_composites(determine_execution_order([mixer | Reverb() | Dac()]))
    groups, _ = fusion.find_fusable(order)
    plan = fusion.shard_groups(groups, 1, 4)           # 3 chains, 4 ranks: not sharded
    assert all(sh is None and not dropped for _, _, _, sh, dropped in plan)
    plan = fusion.shard_groups(groups, 1, 2)
    by_sink = {type(sink).__name__: (len(kept), sh) for sink, kept, _, sh, _ in plan}
    assert by_sink['Mixer'][0] == 1 and by_sink['Mixer'][1].first == 2
    assert by_sink['Reverb'] == (1, None)              # the post-reduce stage is nobody's shard
