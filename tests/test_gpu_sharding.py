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


# ---- voice shards behind run_klang (klang_amd.klang.run_klang(shard=...), fusion.shard_groups)
def _api_patch(n_voices, stereo=False, tail=None):
    """n Voice(saw, ADSR) | Filter chains into a Mixer / StereoMixer | [tail] | Dac, one note each."""
    from klang_amd.audio import ADSR, Dac, Filter, Mixer, Oscillator, StereoMixer, Voice, sawtooth
    from klang_amd.messages import Note
    mixer = StereoMixer() if stereo else Mixer()
    for k in range(n_voices):
        voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.005, .02, .6, .03))
        voice.input.push(Note(20 + k, .3 + .05 * (k % 7), frequency=110. * (1 + .037 * k)))
        mixer += voice | Filter(frequency=500. + 30. * k)
    mixer.gains = [1. - .004 * k for k in range(n_voices)]
    if stereo:
        mixer.pannings = [(-1) ** k * k / n_voices for k in range(n_voices)]
    end = mixer if tail is None else mixer | tail()
    return end | Dac(nChannels=2 if stereo else 1)


@pytest.fixture()
def world_of_one():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0,
                            world_size=1, device_id=torch.device('cuda', 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize('stereo', [False, True])
def test_run_klang_sharded_in_a_world_of_one_equals_the_plain_bounce(hip, tmp_path, world_of_one,
                                                                     stereo):
    """run_klang(filepath=..., shard=(0, 1)) takes the multi-rank path -- shard plan, divisor 1,
    render_device, RCCL reduce, division on the root, klmi_bank_bounce_bus -- and must write the
    file the plain one-pass bounce writes.  64 inputs: the division by the input count is exact
    in float32 and in float64, so the two roundings (here: float32 partial bus, then / 64; plain:
    float64 sum / 64, then float32) agree bit for bit."""
    from klang_amd.klang import run_klang
    dur = 40 * B / 44100. + 1e-9
    files = [str(tmp_path / name) for name in ('plain.wav', 'sharded.wav')]
    plain = np.concatenate(run_klang(_api_patch(64, stereo), offline=True, duration=dur,
                                     fadeout=.05, filepath=files[0]))
    sharded = np.concatenate(run_klang(_api_patch(64, stereo), offline=True, duration=dur,
                                       fadeout=.05, filepath=files[1], shard=(0, 1)))
    assert plain.shape == (40 * B, 2 if stereo else 1) and np.abs(plain).max() > 1e-2
    np.testing.assert_array_equal(sharded, plain)
    assert open(files[0], 'rb').read() == open(files[1], 'rb').read()
    # an input count that is not a power of two: one more float32 rounding, nothing else
    plain = np.concatenate(run_klang(_api_patch(37, stereo), offline=True, duration=dur, fadeout=0.))
    sharded = np.concatenate(run_klang(_api_patch(37, stereo), offline=True, duration=dur,
                                       fadeout=0., shard=(0, 1)))
    assert np.abs(sharded - plain).max() <= 1e-7


def test_two_ranks_rehearsed_on_one_gpu_through_run_klang(hip, tmp_path, monkeypatch):
    """Rank 1 and then rank 0 of a world of two, one after the other in this process, with the
    reduce replaced by a mailbox: rank 1 (a worker: run_shard_worker) leaves its partial buses,
    rank 0 adds them to its own in the order its reduces are issued.  The sum must be the
    unsharded patch, the two ranks must issue the same sequence of reduces (count and shapes),
    and only the root writes the file.  Once as a one-pass bounce (mixer | Dac) and once with a
    post-reduce stage (mixer | Reverb | Dac: the Reverb runs on the root only)."""
    from klang_amd import sharding
    from klang_amd.audio import Reverb
    from klang_amd.klang import run_klang
    dur = 70 * B / 44100. + 1e-9
    for tail in (None, lambda: Reverb(decay=1., dryWet=.5)):
        mailbox, shapes = [], {0: [], 1: []}

        def fake_reduce(bus, n_total, world_size, dst=0, group=None, always=False, _rank=[1]):
            shapes[_rank[0]].append(tuple(bus.shape))
            if _rank[0] == 1:
                mailbox.append(bus.clone())
            else:
                bus += mailbox.pop(0)
                bus /= float(n_total)
            return bus
        monkeypatch.setattr(sharding, 'reduce_bus', fake_reduce)
        f1, f0 = str(tmp_path / 'r1.wav'), str(tmp_path / 'r0.wav')
        got1 = run_klang(_api_patch(100, tail=tail), offline=True, duration=dur, fadeout=0.,
                         filepath=f1, shard=(1, 2))
        assert got1 == [] and not (tmp_path / 'r1.wav').exists() and mailbox
        fake_reduce.__defaults__[-1][0] = 0
        got0 = np.concatenate(run_klang(_api_patch(100, tail=tail), offline=True, duration=dur,
                                        fadeout=0., filepath=f0, shard=(0, 2)))
        assert not mailbox and shapes[0] == shapes[1] and (tmp_path / 'r0.wav').exists()
        assert sum(s[0] for s in shapes[0]) == 70
        monkeypatch.undo()
        want = np.concatenate(run_klang(_api_patch(100, tail=tail), offline=True, duration=dur,
                                        fadeout=0.))
        assert np.abs(want).max() > 1e-2
        assert report('run_klang sharded over 2 ranks vs one bank', np.abs(got0 - want).max()) <= 1e-6
        (tmp_path / 'r0.wav').unlink()


def test_sharded_run_klang_routes_prescheduled_notes_to_the_owner(hip, monkeypatch):
    """A NoteScript in front of every voice: the notes are prescheduled on the device
    (klang.preschedule) by the rank that owns the voice and swallowed by the other."""
    from klang_amd import sharding
    from klang_amd.audio import ADSR, Dac, Mixer, Oscillator, Voice, sawtooth
    from klang_amd.klang import run_klang
    from klang_amd.messages import Note
    from klang_amd.notes import NoteScript

    def build():
        mixer = Mixer()
        for k in range(6):
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.005, .02, .6, .01))
            script = NoteScript({2 + k: [Note(60 + k, .5, frequency=150. * (k + 1))],
                                 9 + k: [Note(60 + k, 0., frequency=150. * (k + 1))]})
            script | voice
            mixer += voice
        return mixer | Dac()
    dur = 24 * B / 44100. + 1e-9
    mailbox = []
    rank = [1]

    def fake_reduce(bus, n_total, world_size, dst=0, group=None, always=False):
        if rank[0] == 1:
            mailbox.append(bus.clone())
        else:
            bus += mailbox.pop(0)
            bus /= float(n_total)
        return bus
    monkeypatch.setattr(sharding, 'reduce_bus', fake_reduce)
    assert run_klang(build(), offline=True, duration=dur, fadeout=0., shard=(1, 2)) == []
    rank[0] = 0
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., shard=(0, 2)))
    monkeypatch.undo()
    want = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0.))
    assert np.abs(want).max() > 1e-2 and np.abs(got - want).max() <= 1e-6


def test_sharding_needs_something_to_shard(hip):
    from klang_amd.audio import Dac, Oscillator
    from klang_amd.errors import KlangError
    from klang_amd.klang import run_klang
    with pytest.raises(KlangError, match='nothing to shard'):
        run_klang(Oscillator(440.) | Dac(), offline=True, duration=.1, shard=(0, 2))
    with pytest.raises(KlangError, match='offline'):
        run_klang(_api_patch(4), shard=(0, 2))


def _two_process_worker(rank, world, port, out_path, n_voices, dur):
    """One rank of a REAL two-process run on the one GPU of the box: gloo group (RCCL refuses two
    ranks on one device), so the HIP bus of each process meets the collective through host
    copies (sharding.reduce_bus); everything else is the production path of run_klang."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from klang_amd.klang import run_klang
    from tests.test_gpu_sharding import _api_patch
    frames = run_klang(_api_patch(n_voices), offline=True, duration=dur, fadeout=.02,
                       filepath=out_path + '.wav')       # shard: what the process group says
    if rank == 0:
        np.save(out_path, np.concatenate(frames))
    else:
        assert frames == []
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_through_run_klang(hip, tmp_path):
    """Two processes, one per rank, both on device 0: each builds the same 90-voice patch, calls
    run_klang with no shard argument (the process group decides), keeps its half of the voices,
    and the partial HIP buses are summed by a real collective.  Rank 0's frames and WAV file
    against the unsharded patch rendered here."""
    import wave

    import torch.multiprocessing as mp
    from klang_amd.klang import run_klang
    dur = 48 * B / 44100. + 1e-9
    out = str(tmp_path / 'frames.npy')
    mp.get_context('spawn')
    mp.spawn(_two_process_worker, args=(2, _free_port(), out, 90, dur), nprocs=2, join=True)
    got = np.load(out)
    want = np.concatenate(run_klang(_api_patch(90), offline=True, duration=dur, fadeout=.02,
                                    filepath=str(tmp_path / 'plain.wav')))
    assert got.shape == want.shape == (48 * B, 1) and np.abs(want).max() > 1e-2
    assert report('run_klang, 2 processes on one GPU vs one bank', np.abs(got - want).max()) <= 1e-6
    with wave.open(out + '.wav') as a, wave.open(str(tmp_path / 'plain.wav')) as b:
        pa = np.frombuffer(a.readframes(a.getnframes()), dtype=np.int16).astype(int)
        pb = np.frombuffer(b.readframes(b.getnframes()), dtype=np.int16).astype(int)
    assert pa.shape == pb.shape and np.abs(pa - pb).max() <= 2      # int16 steps of the normalised samples
