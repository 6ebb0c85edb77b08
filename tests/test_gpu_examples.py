"""The example patches (examples/, rewritten from the reference's examples/) run through
run_klang on the device and must equal the explicit host evaluation of the same blocks."""
import importlib
import wave

import numpy as np
import pytest

from tests.helpers import report

from klang_amd.klang import run_klang

pytestmark = pytest.mark.gpu
B = 256


@pytest.mark.parametrize('name,duration,channels,kwargs', [
    ('hello_world', .3, 1, {}), ('haunting_envelopes', .5, 2, {}),
    ('reverberation_demo', 1.6, 1, {}), ('synthesizer_demo', 1.5, 1, {}),
    ('drum_machine', 2.2, 1, {}),
    ('sequencer_demo', 2.5, 1, {}),                   # with the hi-hat (noise drawn on the host)
    ('sequencer_demo', 2.5, 1, dict(hihat=False)),    # Sequencer run ahead, notes prescheduled
    ('arpeggiator_demo', 3., 1, {}),
    ('tempo_aware_effects', 2.5, 2, {}),
    ('micro_rhythm_demo', 3., 1, dict(rate=.5)),
    ('aeolian_arp', 3., 2, {})])                  # stereo signals through chains and the mixer        # note-value rates and times, StereoDelay opened
def test_example_matches_the_host_evaluation(hip, tmp_path, name, duration, channels, kwargs):
    mod = importlib.import_module('examples.' + name)
    out = {}
    for backend in ('hip', 'host'):
        path = str(tmp_path / ('%s_%s.wav' % (name, backend)))
        frames = run_klang(mod.build(**kwargs), backend=backend, filepath=path, offline=True,
                           duration=duration, fadeout=.1)
        with wave.open(path, 'rb') as wf:
            assert wf.getnchannels() == channels and wf.getframerate() == 44100
            pcm = np.frombuffer(wf.readframes(-1), dtype='<i2').astype(int)
        out[backend] = (np.concatenate(frames), pcm)
    got, want = out['hip'][0], out['host'][0]
    assert got.shape == want.shape == (int(duration * 44100 // B) * B, channels)
    assert np.abs(want).max() > .02
    label = name + (' ' + ','.join('%s=%s' % kv for kv in kwargs.items()) if kwargs else '')
    assert report('example patch: ' + label, np.abs(got - want).max()) <= 1e-5
    assert np.abs(out['hip'][1] - out['host'][1]).max() <= 2   # int16 steps after normalisation


def test_sequencer_is_run_ahead_for_an_offline_bounce(hip):
    """Every note of the sequencer demo goes to device blocks (kick, hi-hat envelope,
    synthesizer): the Sequencer leaves the order, its notes are queued on the device for the
    whole piece.  With a host block listening to one of its rows it stays."""
    from examples import sequencer_demo
    from klang_amd import fusion
    from klang_amd.block import Block
    from klang_amd.klang import preschedule
    from klang_amd.schedule import determine_execution_order
    from klang_amd.sequencing import Sequencer
    from klang_amd.wiring import MessageInput
    for hihat in (False, True):
        order = determine_execution_order([sequencer_demo.build(hihat=hihat)])
        order, banks = fusion.compile_order(order, strict=True)
        assert any(isinstance(b, fusion.GpuHiHat) for b in order) == hihat
        order = preschedule(order, banks, 300)
        assert not any(isinstance(b, Sequencer) for b in order)
        for bank in banks:
            bank.close()
    dac = sequencer_demo.build(hihat=False)
    order = determine_execution_order([dac])
    seq = next(b for b in order if isinstance(b, Sequencer))
    listener = Block()
    listener.inputs = [MessageInput(owner=listener)]
    seq.outputs[0].connect(listener.input)
    order = determine_execution_order([dac, listener])
    order, banks = fusion.compile_order(order, strict=True)
    assert any(isinstance(b, Sequencer) for b in preschedule(order, banks, 300))
    for bank in banks:
        bank.close()


def test_aeolian_arp_leaves_no_note_source_on_the_host(hip):
    """Four Sequencers and an Arpeggiator in front of a kick, a hi-hat and two synthesizers:
    all of them are upstream of device message inputs only, so an offline bounce runs them
    ahead and what is left in the order are device blocks and host plumbing (the Lfo that
    sweeps a cutoff is fused into its filter's chain)."""
    from examples import aeolian_arp
    from klang_amd import fusion
    from klang_amd.klang import preschedule
    from klang_amd.schedule import determine_execution_order
    from klang_amd.sequencing import Arpeggiator, Sequencer
    order = determine_execution_order([aeolian_arp.build()])
    assert sum(isinstance(b, Sequencer) for b in order) == 4
    order, banks = fusion.compile_order(order, strict=True)
    order = preschedule(order, banks, 200)
    left = sorted({type(b).__name__ for b in order})
    assert not any(isinstance(b, (Sequencer, Arpeggiator)) for b in order), left
    assert set(left) <= {'GpuVoiceBank', 'GpuKick', 'GpuHiHat', 'AudioSplitter', 'AudioCombiner',
                         'Dac'}, left
    for bank in banks:
        bank.close()


def test_banks_are_chained_on_the_device(hip, monkeypatch):
    """`mixer | Reverb | Dac` (examples/reverberation_demo.py:33-34 of the reference): the
    synthesizers' bank and the Reverb's bank are linked device to device, so the offline
    bounce is one device pass over both (klmi_bank_set_input_from_bank + klmi_bank_bounce)
    and equals the per-buffer evaluation in which every buffer crosses the host."""
    from examples import reverberation_demo
    from klang_amd import fusion
    from klang_amd.schedule import determine_execution_order
    order, banks = fusion.compile_order(determine_execution_order([reverberation_demo.build()]))
    device = [b for b in order if isinstance(b, fusion.GpuVoiceBank)]
    assert len(device) == 2 and device[1].upstream is device[0] and device[0].upstream is None
    for bank in banks:
        bank.close()
    calls = []
    real = fusion.GpuVoiceBank.bounce
    monkeypatch.setattr(fusion.GpuVoiceBank, 'bounce',
                        lambda self, *a, **k: calls.append(self) or real(self, *a, **k))
    fast = np.concatenate(run_klang(reverberation_demo.build(), offline=True, duration=2.4, fadeout=.2))
    assert len(calls) == 1 and calls[0].upstream is not None      # one pass, end to end
    # buffers rendered ahead in lock-step (no bounce: lookahead given, engine loop on the host)
    monkeypatch.setattr(fusion.GpuVoiceBank, 'bounce', real)
    monkeypatch.setattr(fusion, 'CHAIN_ON_DEVICE', False)
    slow = np.concatenate(run_klang(reverberation_demo.build(), offline=True, duration=2.4, fadeout=.2))
    assert fast.shape == slow.shape and np.abs(slow).max() > .02
    assert report('bank -> bank on the device vs per-buffer host hop', np.abs(fast - slow).max()) <= 1e-7
