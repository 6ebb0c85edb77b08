"""Entry point: ``run_klang(*blocks, filepath=..., offline=True, duration=...)``
(klang/klang.py:16-90).

``backend='hip'`` (default) compiles the execution order with
klang_amd.fusion so that the DSP hot path runs in the HIP library; it raises if
the library or a GPU is missing, or if a hot-path block cannot be fused --
never a silent CPU evaluation.  ``backend='host'`` is the explicit NumPy
evaluation of the same blocks (plumbing / reference behaviour, no GPU).
"""
import collections
import logging

from klang_amd.audio.engine import (Adc, Dac, look_for_audio_blocks, run_audio_engine,  # noqa: F401
                                    write_pcm)
from klang_amd.composite import Composite
from klang_amd.audio.signals import INTERVAL
from klang_amd.config import BUFFER_SIZE, SAMPLING_RATE
from klang_amd.constants import INF
from klang_amd.errors import KlangError
from klang_amd.schedule import determine_execution_order

BACKENDS = ('hip', 'host')


def unravel(execOrder):
    """Flatten composites into their internal order (klang.py:16-24)."""
    for block in execOrder:
        if isinstance(block, Composite):
            yield from unravel(block.execOrder)
        else:
            yield block


def validate_global_execution_order(execOrder, logger):
    """Warn about blocks that would run more than once per buffer (klang.py:27-48)."""
    counts = collections.Counter(unravel(execOrder))
    dup = ['  - %s: %dx' % (b, n) for b, n in counts.items() if n > 1]
    if dup:
        logger.warning('Found duplicates in flattened execOrder! Following blocks will get '
                       'executed multiple times per buffer cycle!\n%s', '\n'.join(dup))


def bounce_banks(execOrder, adc, dac, duration=INF, offline=False):
    """(banks, n_cycles) when the whole offline piece can be bounced in one device pass, else
    None: exactly one device bank feeding the Dac directly -- or a chain of banks each fed by
    the one before it on the device (`mixer | Reverb | Dac`) -- and nothing else in the order.
    A pure function of the compiled order, so that every rank of a sharded run decides alike."""
    from klang_amd import fusion
    if not offline or duration == INF or adc.nChannels:
        return None
    banks = [b for b in execOrder if isinstance(b, fusion.GpuVoiceBank)]
    if not banks or any(b not in banks and b is not dac for b in execOrder):
        return None
    bank = banks[-1]
    if banks[0].external or any(b.upstream is not a for a, b in zip(banks, banks[1:])):
        return None
    n_cycles = int(duration * SAMPLING_RATE // BUFFER_SIZE)
    if n_cycles < 1:
        return None
    feeds = [i.incomingConnection for i in dac.inputs if i.incomingConnection is not None]
    if (feeds != [bank.sink.output] or dac.inputs[0].incomingConnection is None
            or bank.bank.channels != dac.nChannels):
        return None
    return banks, n_cycles


def device_bounce(execOrder, adc, dac, duration=INF, fadeout=0., filepath='', offline=False):
    """The whole offline bounce in one device pass (klmi_bank_bounce), or None when the patch
    is not of that kind (bounce_banks).  Packing, fade-out and the peak-normalised int16
    samples (klanggeber.py:81-97, 205-249, wavfile.py:54-63) then happen on the device; the
    host only writes the container."""
    from klang_amd import native
    found = bounce_banks(execOrder, adc, dac, duration, offline)
    if found is None:
        return None
    banks, n_cycles = found
    # one bank, or a chain of banks each fed by the one before it on the device
    # (`mixer | Reverb | Dac`): the upstream banks render the whole piece, the last one bounces
    bank = banks[-1]
    try:
        for upstream, downstream in zip(banks, banks[1:]):
            upstream.render_for(downstream, n_cycles)
        frames, pcm = bank.bounce(duration, fadeout, want_pcm=bool(filepath))
    except native.NativeError as err:
        if err.code == native.E_ARG and 'Zero amplitude' in str(err):
            raise ValueError(str(err)) from None     # math.py:17-19
        raise
    if filepath:
        logging.getLogger('KlangGeber').info('Writing audio dump to WAV file %r', filepath)
        write_pcm(pcm, filepath)
    return list(frames.reshape(-1, BUFFER_SIZE, dac.nChannels))


def run_shard_worker(execOrder, banks, adc, dac, duration=INF, fadeout=0., filepath='',
                     offline=False):
    """A rank other than the root of a sharded run: it renders its voice shards and takes part
    in every reduce the root issues -- the same number of render calls with the same number of
    buffers each -- and nothing else.  Blocks behind the last sharded bank (post-reduce stages:
    `mixer | Reverb | Dac`, the Dac itself) run on the root only; no file is written here."""
    found = bounce_banks(execOrder, adc, dac, duration, offline)
    if found is not None:
        found[0][-1].bounce(duration, fadeout, want_pcm=False)      # render + reduce
        return []
    last = max(i for i, b in enumerate(execOrder) if getattr(b, 'shard', None) is not None)
    order = [b for b in execOrder[:last + 1] if not isinstance(b, (Adc, Dac))]
    from klang_amd.clock import ClockMixin
    play_time = 0.
    for _ in range(int(duration * SAMPLING_RATE // BUFFER_SIZE)):
        ClockMixin.set_current_time(play_time)
        for block in order:
            block.update()
        play_time += INTERVAL
    return []


def preschedule(execOrder, banks, n_cycles):
    """Event pre-computation for an offline bounce.  The host blocks that
    feed notes to the device (NoteScript, Sequencer, Arpeggiator, note effects: anything
    upstream of a device message input) are run for the whole piece NOW, under the clock the
    engine will keep (a running sum of INTERVAL from 0, klanggeber.py:232, 249), and what
    arrives at every device message input in buffer k is queued on the device for buffer k.
    The sources are then dropped from the order: nothing on the host is left to run between
    buffers, so the banks may render ahead (or bounce the piece in one pass).

    This is only done when that set of blocks is closed: nothing in it listens to the device
    or the Adc, and nothing it sends goes anywhere but to itself or to device message inputs.
    Otherwise the order is returned as it is and the sources run buffer by buffer."""
    from klang_amd import fusion
    from klang_amd.block import input_connections, output_connections
    from klang_amd.clock import ClockMixin
    if n_cycles is None:
        return execOrder
    targets = {}
    for bank in banks:
        for inp, handle in bank.message_targets().items():
            targets[inp] = (bank, handle)
    device = (fusion.GpuVoiceBank, fusion.GpuKick, fusion.GpuChirper, fusion.GpuHiHat, Adc, Dac)
    hosts = [b for b in execOrder if not isinstance(b, device)]
    # the host blocks upstream of a device message input
    feeders = {inp.incomingConnection.owner for inp in targets if inp.incomingConnection}
    sources, frontier = set(), [b for b in hosts if b in feeders]
    while frontier:
        block = frontier.pop()
        if block in sources:
            continue
        sources.add(block)
        frontier.extend(src.owner for src, _ in input_connections(block))
    if not sources or not sources <= set(hosts):
        return execOrder        # nothing to do, or a source listens to a device block
    for block in sources:
        for _, dst in output_connections(block):
            if dst not in targets and dst.owner not in sources:
                return execOrder
    order = [b for b in execOrder if b in sources]
    play_time = 0.
    try:
        for k in range(n_cycles):
            ClockMixin.set_current_time(play_time)
            for block in order:
                block.update()
            for inp, (bank, handle) in targets.items():
                for note in inp.receive():      # the queue has kept the newest 50 of a buffer
                    bank.schedule(handle, note, k)
            play_time += INTERVAL
    finally:
        ClockMixin.set_current_time(0.)
    return [b for b in execOrder if b not in sources]


def run_klang(*blocks, backend='hip', lookahead=None, shard=None, **kwargs):
    """Run a block network.  kwargs: duration, fadeout, filepath, offline (see
    klang_amd.audio.engine.run_audio_engine).  Returns the list of rendered
    float32 buffers (BUFFER_SIZE, nChannels).

    Several GPUs: start one process per GPU (torchrun or any launcher that sets RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*), every one building the SAME patch and calling
    run_klang the same way.  With a torch.distributed group of more than one rank (or
    `shard=(rank, world[, group])`) every mixer of uniform device chains keeps only this
    rank's contiguous range of voices (klang_amd.sharding), notes reach the rank that owns
    their voice, each render ends in ONE reduce of the mix bus onto rank 0, and rank 0 alone
    runs what follows the mixers, returns the frames and writes `filepath`; the other ranks
    return []."""
    if not blocks:
        raise ValueError('No blocks to run specified!')
    if backend not in BACKENDS:
        raise ValueError('backend must be one of %s' % (BACKENDS,))
    logger = logging.getLogger('Klang')
    execOrder = determine_execution_order(blocks)
    validate_global_execution_order(execOrder, logger)
    adc, dac = look_for_audio_blocks(execOrder)
    if adc.nChannels == 0 and dac.nChannels == 0:
        raise KlangError('Did not find any audio activity (no Adc / Dac in the patch)')

    banks = []
    if shard is not None or backend == 'hip':
        from klang_amd import sharding
        shard = sharding.resolve(shard) if backend == 'hip' or shard is not None else None
    if shard is not None:
        if backend != 'hip':
            raise KlangError("voice shards run on the device: backend='hip'")
        if not kwargs.get('offline') or kwargs.get('duration', INF) == INF:
            raise KlangError('a sharded run is an offline bounce with a finite duration')
    if backend == 'hip':
        from klang_amd import fusion
        execOrder, banks = fusion.compile_order(execOrder, strict=True, shard=shard)
        if shard is not None and not any(getattr(b, 'shard', None) for b in banks):
            for bank in banks:
                bank.close()
            raise KlangError('nothing to shard: no mixer with at least %d uniform device chains '
                             'in the patch' % shard[1])
        duration = kwargs.get('duration', INF)
        n_cycles = None if duration == INF else int(duration * SAMPLING_RATE // BUFFER_SIZE)
        if kwargs.get('offline'):
            execOrder = preschedule(execOrder, banks, n_cycles)
        # rendering ahead is safe when nothing on the host can emit a note between buffers (a
        # GpuHiHat draws its noise on the host, buffer by buffer: it keeps everything at 1)
        device_blocks = (fusion.GpuVoiceBank, fusion.GpuKick, fusion.GpuChirper, Adc, Dac)
        only_io = all(isinstance(b, device_blocks) for b in execOrder)
        ahead = (64 if only_io else 1) if lookahead is None else int(lookahead)
        for bank in banks:
            bank.lookahead = ahead
            bank.remaining = n_cycles

    def execute_all_blocks():
        for block in execOrder:
            block.update()

    try:
        if shard is not None and shard[0] != 0:
            return run_shard_worker(execOrder, banks, adc, dac, **kwargs)
        if backend == 'hip':
            frames = device_bounce(execOrder, adc, dac, **kwargs)
            if frames is not None:
                return frames
        return run_audio_engine(adc, dac, execute_all_blocks, **kwargs)
    finally:
        for bank in banks:
            bank.sync_host_blocks()
            bank.close()
