"""Note sources on the host (klang_amd/sequencing.py): the reference's known answers
(tests/test_sequencer.py, tests/test_arpeggiator.py) restated, and every note the real
Sequencer / Arpeggiator sent over a few seconds (tests/golden/sequencing.npz, written by
tools/gen_golden.py from the reference under the engine's clock) reproduced row for row."""
from fractions import Fraction

import numpy as np
import pytest

from klang_amd.audio.signals import INTERVAL
from klang_amd.block import Block
from klang_amd.clock import ClockMixin
from klang_amd.constants import TAU
from klang_amd.messages import Note
from klang_amd.sequencing import (VALID_ORDERS, Arpeggiator, Arpeggio, MaxNotes, MessageMixer,
                                  NoteLengthener, PizzaSlicer, Sequence, Sequencer,
                                  arpeggio_index_permutation, atleast_1d, atleast_2d,
                                  interleave, pizza_slice_number, random_pattern)
from klang_amd.tempo import QUARTER_NOTE, SIXTEENTH_NOTE
from klang_amd.wiring import MessageInput

C, D, E, F, G, A, B = (Note(p) for p in (60, 62, 64, 65, 67, 69, 71))
SIXTEEN_STEPS = list(range(16))


def run_source(block, n_buffers, feed=None):
    """The generator's loop (tools/gen_golden.py run_source) over klang_amd blocks."""
    sinks = []
    for out in block.outputs:
        sink = MessageInput()
        out.connect(sink)
        sinks.append(sink)
    rows, play_time = [], 0.
    try:
        for k in range(n_buffers):
            ClockMixin.set_current_time(play_time)
            for note in (feed or {}).get(k, ()):
                block.input.push(note)
            block.update()
            for ch, sink in enumerate(sinks):
                for note in sink.receive():
                    rows.append([k, ch, note.pitch, note.velocity, note.frequency])
            play_time += INTERVAL
    finally:
        ClockMixin.set_current_time(0.)
    return np.array(rows, dtype=float).reshape(-1, 5)


def chord_feed(g):
    feed = {}
    for k, pitch, velocity in g['arp_feed']:
        feed.setdefault(int(k), []).append(Note(int(pitch), float(velocity)))
    return feed


# ---- fixtures from the real reference ---------------------------------------------------------

def test_sequencer_notes_vs_reference(golden):
    g = golden('sequencing')
    got = run_source(Sequencer([[60, 0, 63, 67, [60, 64], 0, 72, 65], [36, 0, 0, 43]], tempo=125),
                     700)
    np.testing.assert_array_equal(got, g['split'])
    got = run_source(Sequencer([[48, 55, 0, 51], [72, 0, 70, 0, 67, 0]], tempo=97,
                               grid=Fraction(1, 8), absNoteLength=.21, splitOutputs=False), 700)
    np.testing.assert_array_equal(got, g['merged'])
    got = run_source(Sequencer([60, 62, 64, 65], tempo=120, grid=Fraction(1, 4),
                               relNoteLength=.9), 700)
    np.testing.assert_array_equal(got, g['quarter'])
    assert len(g['split']) > 80 and set(g['split'][:, 1]) == {0., 1.}


@pytest.mark.parametrize('order', ['up', 'down', 'upDown', 'downUp', 'alternating'])
def test_arpeggiator_notes_vs_reference(golden, order):
    g = golden('sequencing')
    got = run_source(Arpeggiator(.11, order), 600, chord_feed(g))
    np.testing.assert_array_equal(got, g['arp_' + order])


def test_arpeggiator_with_a_note_value_interval(golden):
    g = golden('sequencing')
    got = run_source(Arpeggiator(Fraction(1, 16), 'up'), 300, chord_feed(g))
    np.testing.assert_array_equal(got, g['arp_note_value'])


# ---- tests/test_sequencer.py restated -----------------------------------------------------------

def test_atleast_helpers():
    assert atleast_1d(0) == [0]
    assert atleast_1d([0, 1, 2, 3]) == [0, 1, 2, 3]
    assert atleast_1d([[0, 1, 2, 3]]) == [[0, 1, 2, 3]]
    np.testing.assert_equal(atleast_1d(np.array(0)), [0])
    np.testing.assert_equal(atleast_1d(np.array([[0, 1, 2, 3]])), [[0, 1, 2, 3]])
    assert atleast_2d(0) == [[0]]
    assert atleast_2d([0, 1, 2, 3]) == [[0, 1, 2, 3]]
    assert atleast_2d([[[0, 1, 2, 3]]]) == [[[0, 1, 2, 3]]]
    np.testing.assert_equal(atleast_2d(np.array(0)), [[0]])
    np.testing.assert_equal(atleast_2d(np.array([0, 1, 2, 3])), [[0, 1, 2, 3]])
    assert atleast_2d([[1, 2, 3, 4], [5, 6], [7]]) == [[1, 2, 3, 4], [5, 6], [7]]   # ragged


def test_pizza_slicer():
    """Start index, a pattern of four with the phases actually applied, double jumps."""
    recv = MessageInput()
    pizza = PizzaSlicer(16)
    pizza | recv
    assert pizza.currentIdx == -1
    pizza.update()
    assert pizza.currentIdx == 0 and list(recv.receive()) == [0]

    recv = MessageInput()
    pizza = PizzaSlicer(4)
    pizza | recv
    for phase, want in [(.0 * TAU, [0]), (.25 * TAU - .1, []), (.25 * TAU, [1]),
                        (.25 * TAU + .1, []), (.50 * TAU, [2]), (.75 * TAU + 1e-9, [3]),
                        (1.0 * TAU - .1, []), (1.0 * TAU + .1, [0])]:
        pizza.input.set_value(phase)
        pizza.update()
        assert list(recv.receive()) == want, phase

    recv = MessageInput()
    pizza = PizzaSlicer(4)
    pizza | recv
    for phase, want in [(.0 * TAU, [0]), (.50 * TAU, [1, 2]), (1.0 * TAU, [3, 0])]:
        pizza.input.set_value(phase)
        pizza.update()
        assert list(recv.receive()) == want
    assert pizza_slice_number(-.1, 4) == 3 and pizza_slice_number(3 * TAU + .1, 4) == 0


def test_sequence_properties_and_internal_order():
    seq = Sequence(None, SIXTEEN_STEPS, tempo=120, beatValue=QUARTER_NOTE, grid=SIXTEENTH_NOTE)
    assert (seq.length, seq.duration) == (16, 2.0)
    seq = Sequence(None, SIXTEEN_STEPS, tempo=100, beatValue=QUARTER_NOTE, grid=SIXTEENTH_NOTE)
    assert (seq.length, seq.duration) == (16, 2.4)
    seq = Sequence(None, [0, 1, 2, 3], tempo=120, beatValue=QUARTER_NOTE, grid=QUARTER_NOTE)
    assert (seq.length, seq.duration) == (4, 2.0)
    seq = Sequence(None, SIXTEEN_STEPS)
    assert seq.execOrder == [seq.phasor, seq.pizza, seq.lookup, seq.lengthener]


def test_sequence_notes_come_through():
    seq = Sequence(None, [1, 2, 3, 4], tempo=120, beatValue=QUARTER_NOTE, grid=QUARTER_NOTE,
                   absNoteLength=1.)

    def set_time(time):
        seq.phasor.currentPhase = TAU / seq.duration * time
        seq.lengthener.clock = lambda: time

    recv = MessageInput()
    seq | recv
    set_time(0.)
    seq.update()
    note, = recv.receive()
    assert note.on and note.pitch == 1
    set_time(.5)
    seq.update()
    note, = recv.receive()
    assert note.on and note.pitch == 2
    set_time(1.)
    seq.update()
    off, on = recv.receive()
    assert off.off and off.pitch == 1 and on.on and on.pitch == 3


def test_sequencer_channels_and_micro_rhythm_order():
    assert Sequencer([1, 2, 3, 4]).nChannels == 1
    assert Sequencer([[1, 2, 3, 4]]).nChannels == 1
    assert Sequencer([[1, 2, 3, 4], [5, 6], [7]]).nChannels == 3
    sequencer = Sequencer([SIXTEEN_STEPS, SIXTEEN_STEPS, SIXTEEN_STEPS])
    seq0, seq1, seq2 = sequencer.sequences
    assert sequencer.execOrder == [seq0, seq1, seq2]
    mr = Block(nInputs=1, nOutputs=1)
    sequencer.apply_micro_rhythm(mr, channel=1)
    assert sequencer.execOrder == [seq0, seq1, mr, seq2]
    sequencer.reset_micro_rhythm(channel=1)
    assert sequencer.execOrder == [seq0, seq1, seq2]


# ---- tests/test_arpeggiator.py restated ---------------------------------------------------------

def take(arpeggio, n):
    return [next(arpeggio) for _ in range(n)]


def test_interleave():
    assert interleave('ab', '01') == ['a', '0', 'b', '1']
    assert interleave('abc', '01') == ['a', '0', 'b', '1', 'c']
    assert interleave('abcde', '01') == ['a', '0', 'b', '1', 'c', 'd', 'e']
    assert interleave('ab', '0123') == ['a', '0', 'b', '1', '2', '3']


def test_arpeggio_index_permutations():
    with pytest.raises(ValueError):
        arpeggio_index_permutation('notAValidOrder', 2)
    for order in VALID_ORDERS:
        assert arpeggio_index_permutation(order, 0) == []
    for order in ('up', 'down', 'upDown', 'downUp', 'alternating'):
        assert arpeggio_index_permutation(order, 1) == [0]
    even = dict(up=[0, 1, 2, 3], down=[3, 2, 1, 0], upDown=[0, 1, 2, 3, 2, 1],
                downUp=[3, 2, 1, 0, 1, 2], alternating=[0, 3, 1, 2])
    odd = dict(up=[0, 1, 2], down=[2, 1, 0], upDown=[0, 1, 2, 1], downUp=[2, 1, 0, 1],
               alternating=[0, 2, 1])
    for order in even:
        assert arpeggio_index_permutation(order, 4) == even[order]
        assert arpeggio_index_permutation(order, 3) == odd[order]
    assert sorted(arpeggio_index_permutation('random', 7)) == list(range(7))


def test_arpeggio_container():
    with pytest.raises(StopIteration):
        next(Arpeggio())
    for order in VALID_ORDERS:
        assert take(Arpeggio(order, initialNotes=[C]), 10) == [C] * 10
    with pytest.raises(ValueError):
        Arpeggio().remove_note(C)
    with pytest.raises(ValueError):
        Arpeggio('sideways')
    arp = Arpeggio(initialNotes=[C, E, G])
    for note in (C, E, G):
        arp.remove_note(note)
    assert arp.current == 0 and arp.permutation == []
    assert take(Arpeggio('up', [C, E, G]), 4) == [C, E, G, C]
    assert take(Arpeggio('down', [C, E, G]), 4) == [G, E, C, G]
    assert take(Arpeggio('upDown', [C, E, G]), 5) == [C, E, G, E, C]
    assert take(Arpeggio('downUp', [C, E, G]), 5) == [G, E, C, E, G]
    assert take(Arpeggio('alternating', [A, G, F, D, C]), 6) == [C, A, D, G, F, C]
    arp = Arpeggio('up', [C, E, G])
    assert take(arp, 2) == [C, E]
    arp.add_note(B)                      # keeps rising when a note is added at the end
    assert take(arp, 2) == [G, B]
    arp = Arpeggio('up', [C, E, G])
    assert take(arp, 2) == [C, E]
    arp.remove_note(G)                   # back to the first note when the last one goes
    assert next(arp) == C


def test_note_lengthener_scenario():
    nl = NoteLengthener(duration=2)
    recv = MessageInput()
    nl.output.connect(recv)
    try:
        nl.input.push(C)
        nl.update()
        assert list(nl.activeNotes) == [(2, C)]
        nl.set_current_time(1.)
        nl.input.push(E)
        nl.update()
        assert list(nl.activeNotes) == [(2, C), (3, E)]
        nl.set_current_time(2.)
        nl.update()
        assert list(nl.activeNotes) == [(3, E)]
        assert recv.receive_latest() == C._replace(velocity=0.)
    finally:
        # set through the instance, the time lands on NoteLengthener itself and would hide
        # the shared clock from every later NoteLengthener (as it does in the reference)
        del NoteLengthener.currentTime


def test_max_notes_message_mixer_and_random_pattern():
    limiter, recv = MaxNotes(2), MessageInput()
    limiter.output.connect(recv)
    for note in (C, E, C.silence(), G):
        limiter.input.push(note)
    limiter.update()
    assert list(recv.receive()) == [C, E, C.silence(), G]   # the third note-on drops the oldest
    mixer, recv = MessageMixer(2), MessageInput()
    mixer.output.connect(recv)
    mixer.inputs[1].push(E)
    mixer.inputs[0].push(C)
    mixer.update()
    assert list(recv.receive()) == [C, E]
    pattern = random_pattern(10, period=3, minVal=40, maxVal=50)
    assert len(pattern) == 10 and pattern[:3] == pattern[3:6] == pattern[6:9]
    assert all(40 <= p <= 50 for p in pattern)


# ---- event pre-computation (klang_amd.klang.preschedule) ---------------------------------------

class _RecordingBank:
    """Stands in for a device bank: the message inputs it owns and what was queued for them."""

    def __new__(cls, inputs):
        from klang_amd import fusion
        bank = object.__new__(type('RecordingBank', (fusion.GpuVoiceBank,), {
            'message_targets': lambda self: {inp: ch for ch, inp in enumerate(self.inputs_)},
            'schedule': lambda self, handle, note, k: self.rows.append(
                [k, handle, note.pitch, note.velocity, note.frequency]),
            'update': lambda self: None}))
        bank.inputs_, bank.rows = list(inputs), []
        return bank


def test_preschedule_hands_the_reference_event_stream_to_the_device(golden):
    """A Sequencer in front of two device message inputs: run ahead for 700 buffers, what is
    queued for the device is row for row what the real Sequencer sent (fixture `split`), and
    the Sequencer is gone from the order."""
    from klang_amd.audio import AR, MonophonicSynthesizer, Oscillator, Voice
    from klang_amd.klang import preschedule
    seq = Sequencer([[60, 0, 63, 67, [60, 64], 0, 72, 65], [36, 0, 0, 43]], tempo=125)
    synths = [MonophonicSynthesizer(Voice(Oscillator(), AR(.01, .1))) for _ in range(2)]
    for out, synth in zip(seq.outputs, synths):
        out | synth
    bank = _RecordingBank([s.input for s in synths])
    order = preschedule([seq, bank], [bank], 700)
    assert order == [bank]
    got = np.array(sorted(bank.rows, key=lambda r: (r[0], r[1])), dtype=float)
    want = golden('sequencing')['split']
    np.testing.assert_array_equal(got, want[np.lexsort((want[:, 1], want[:, 0]))])
    assert ClockMixin.clock() == 0.


def test_preschedule_follows_note_effects_and_arpeggiators(golden):
    """NoteScript -> Arpeggiator -> device input: both blocks are upstream of the device and
    are run ahead together."""
    from klang_amd.audio import AR, MonophonicSynthesizer, Oscillator, Voice
    from klang_amd.klang import preschedule
    from klang_amd.notes import NoteScript
    g = golden('sequencing')
    script = NoteScript(chord_feed(g))
    arp = Arpeggiator(.11, 'upDown')
    synth = MonophonicSynthesizer(Voice(Oscillator(), AR(.01, .1)))
    script | arp | synth
    bank = _RecordingBank([synth.input])
    assert preschedule([script, arp, bank], [bank], 600) == [bank]
    np.testing.assert_array_equal(np.array(bank.rows, dtype=float), g['arp_upDown'])


def test_preschedule_leaves_open_source_sets_alone():
    """A source that also feeds a host block, or that listens to a host block fed by the
    device, is not run ahead: the order stays as it is, nothing is queued, nothing has run."""
    from klang_amd.audio import AR, HiHat, MonophonicSynthesizer, Oscillator, Voice
    from klang_amd.klang import preschedule
    seq = Sequencer([[60, 0, 63, 67], [0, 42, 0, 42]])
    synth = MonophonicSynthesizer(Voice(Oscillator(), AR(.01, .1)))
    hat = HiHat()
    seq.outputs[0] | synth
    seq.outputs[1] | hat
    bank = _RecordingBank([synth.input])
    order = [seq, hat, bank]
    assert preschedule(order, [bank], 100) == order and bank.rows == []
    assert seq.sequences[0].pizza.currentIdx == -1
    assert preschedule(order, [bank], None) == order      # no known length: nothing to run ahead


# ---- micro rhythm (klang/music/rhythm.py) --------------------------------------------------------

def test_micro_rhythm_warp_and_euclidian_patterns_vs_reference(golden):
    from klang_amd.sequencing import MicroRhyhtm, euclidian_rhythm
    g = golden('sequencing')
    notes = [Fraction(1, 8), Fraction(1, 16), Fraction(3, 16)]
    for row, phrasing in zip(g['micro_warp_out'], (1., .35)):
        mr = MicroRhyhtm(notes, phrasing=phrasing)
        got = np.array([mr.warp(x) for x in g['micro_warp_in']])
        np.testing.assert_array_equal(got, row)
    for row in g['euclid']:
        pulses, slots = int(row[0]), int(row[1])
        assert euclidian_rhythm(pulses, slots) == [int(x) for x in row[2:2 + slots]]
    assert sum(euclidian_rhythm(5, 13)) == 5


def test_sequencer_with_a_micro_rhythm_vs_reference(golden):
    """examples/micro_rhythm_demo.py in small: the second row's phase goes through a
    MicroRhyhtm whose phrasing an Lfo moves; the Lfo has to end up in the sequencer's own
    execution order, and every note leaves in the buffer the real Sequencer sent it."""
    from klang_amd.audio import Lfo, triangle
    from klang_amd.sequencing import MicroRhyhtm
    seq = Sequencer([[60, 0, 0, 0, 60, 0, 0, 0, 60, 0, 0, 0, 60, 0, 0, 60], 4 * [60, 61, 62, 63]],
                    tempo=80)
    lfo = Lfo(frequency=.6, wave_func=triangle)
    mr = MicroRhyhtm([Fraction(1, 8), Fraction(1, 16), Fraction(1, 16), Fraction(1, 8)])
    lfo | mr.phrasing
    seq.apply_micro_rhythm(mr, channel=1)
    assert lfo in seq.execOrder and mr in seq.execOrder
    np.testing.assert_array_equal(run_source(seq, 900), golden('sequencing')['micro_rhythm'])
