"""Note sources and note effects, host side: the pattern Sequencer (klang/sequencer.py:29-301),
the Arpeggiator (klang/arpeggiator.py:24-194) and the note effects both rest on
(klang/note_effects.py:11-91).

These are the reference's control plane: small Python state machines that turn a phase or a
held chord into Note messages, one buffer at a time.  They are mirrored here so that a patch
written against the reference runs unchanged, and because their output over a whole offline
bounce is a pure function of the buffer number: `klang_amd.klang.preschedule` runs them ahead
of the audio, records which note reaches which device voice in which buffer, and hands the
whole piece to the device banks up front ("event pre-computation").
"""
import bisect
import collections
import collections.abc
import itertools
import math
import random

import numpy as np

from klang_amd.audio.filtering import blend
from klang_amd.audio.signals import Phasor
from klang_amd.block import Block
from klang_amd.clock import ClockMixin
from klang_amd.composite import Composite
from klang_amd.constants import TAU
from klang_amd.messages import Note
from klang_amd.tempo import QUARTER_NOTE, SIXTEENTH_NOTE, compute_duration, tempo_2_period
from klang_amd.wiring import MessageInput, MessageOutput, MessageRelay, Relay

__all__ = ['NoteLengthener', 'MaxNotes', 'MessageMixer', 'euclidian_rhythm', 'MicroRhyhtm',
           'random_pattern', 'pizza_slice_number',
           'atleast_1d', 'atleast_2d', 'PizzaSlicer', 'PatternLookup', 'Sequence', 'Sequencer',
           'VALID_ORDERS', 'interleave', 'arpeggio_index_permutation', 'Arpeggio',
           'Arpeggiator']


# ---- note effects (klang/note_effects.py) --------------------------------------------------

class _MessageBlock(Block):
    """One message input, one message output."""

    def __init__(self):
        super().__init__()
        self.inputs = [MessageInput(owner=self)]
        self.outputs = [MessageOutput(owner=self)]


class NoteLengthener(_MessageBlock, ClockMixin):
    """Turns note-ons into notes of `duration` seconds: the note-off follows once the clock
    has reached its end time (note_effects.py:11-47).  Ends are kept in arrival order."""

    def __init__(self, duration):
        super().__init__()
        self.duration = duration
        self.activeNotes = collections.deque()   # (end time, note)

    def outdated_notes(self, now):
        while self.activeNotes and not now < self.activeNotes[0][0]:
            yield self.activeNotes.popleft()[1]

    def update(self):
        now = self.clock()
        for note in self.outdated_notes(now):
            self.output.send(note.silence())
        for note in self.input.receive():
            if note.on:
                self.activeNotes.append((now + self.duration, note))
                self.output.send(note)


class MaxNotes(_MessageBlock):
    """At most `nNotes` sounding notes: a new note-on silences the oldest ones first
    (note_effects.py:50-71)."""

    def __init__(self, nNotes=1):
        super().__init__()
        self.nNotes = nNotes
        self.activeNotes = collections.deque([], maxlen=nNotes + 1)

    def update(self):
        for note in self.input.receive():
            if not note.on:
                continue
            self.activeNotes.append(note)
            while len(self.activeNotes) > self.nNotes:
                self.output.send(self.activeNotes.popleft().silence())
            self.output.send(note)


class MessageMixer(Block):
    """Forwards the messages of all its inputs, input by input (note_effects.py:74-91)."""

    def __init__(self, nInputs=2):
        super().__init__()
        self.outputs = [MessageOutput(owner=self)]
        for _ in range(nInputs):
            self.add_new_channel()

    def add_new_channel(self):
        self.inputs.append(MessageInput(owner=self))

    def update(self):
        for inp in self.inputs:
            for message in inp.receive():
                self.output.send(message)


# ---- rhythm (klang/music/rhythm.py) ----------------------------------------------------------

def euclidian_rhythm(nPulses, nSlots):
    """nPulses spread as evenly as possible over nSlots, as a 0/1 list (Bjorklund's algorithm
    as rhythm.py:28-81 runs it: counts and remainders of a Euclid-style division, the string
    built recursively from them, then reversed)."""
    divisor = nSlots - nPulses
    remainder, count, level = [nPulses], [], 0
    while remainder[level] > 1:
        count.append(divisor // remainder[level])
        remainder.append(divisor % remainder[level])
        divisor = remainder[level]
        level += 1
    count.append(divisor)
    bitmap = []

    def build(lvl):
        if lvl == -1:
            bitmap.append(0)
        elif lvl == -2:
            bitmap.append(1)
        else:
            for _ in range(count[lvl]):
                build(lvl - 1)
            if remainder[lvl]:
                build(lvl - 2)

    build(level)
    return bitmap[::-1]


class MicroRhyhtm(Block):
    """(Spelt as in the reference.)  Warps a bar phase so that evenly spaced steps land on an
    uneven note pattern: the pattern's note starts, as angles, are mapped onto equal angles by
    interpolation; `phrasing` (a value input) blends between the straight and the warped
    phase (rhythm.py:103-172).  Patched into a Sequence's phase insert."""

    def __init__(self, notes, phrasing=1., beatValue=QUARTER_NOTE, kind='linear', name=''):
        super().__init__(nInputs=2, nOutputs=1, name=name)
        self.notes = notes
        self.beatValue = beatValue
        self.phrasing = self.inputs[1]
        self.phrasing.set_value(phrasing)
        self.interpolator = self.create_phase_interpolator(notes, kind)

    @staticmethod
    def create_phase_interpolator(notes, kind):
        import scipy.interpolate
        duration = sum(notes)
        starts = list(itertools.accumulate(notes[:-1], initial=type(duration)()))
        angles = (TAU / duration * np.r_[starts, duration]).astype(float)
        return scipy.interpolate.interp1d(angles, np.linspace(0, TAU, len(notes) + 1), kind)

    def warp(self, phase, n=None):
        """`n`: how many times the pattern repeats on the circle (default: beats per whole note)."""
        n = n or 1 // self.beatValue
        groove = (self.interpolator(n * phase % TAU) + int(n * phase / TAU) * TAU) / n
        return blend(phase % TAU, groove, self.phrasing.value)

    def update(self):
        self.output.set_value(self.warp(self.input.value))


# ---- sequencer (klang/sequencer.py) --------------------------------------------------------

def random_pattern(length, period=None, minVal=60, maxVal=72):
    """`length` random pitches that repeat every `period` (sequencer.py:29-53; Python's
    `random` module, one randint per period entry, drawn before anything is repeated)."""
    period = length if period is None else period
    motif = [random.randint(minVal, maxVal) for _ in range(period)]
    return [motif[step % period] for step in range(length)]


def pizza_slice_number(angle, nSlices):
    """Which of nSlices equal slices of the circle `angle` lies in (sequencer.py:56-66)."""
    turn = (angle % TAU) / TAU
    return int(turn * nSlices)


def _nested(obj, depth):
    """`obj` wrapped in lists until it is `depth` levels of sequence deep (looking at first
    elements only, so ragged rows stay as they are); arrays go through NumPy's own
    atleast_*d (sequencer.py:69-93)."""
    if isinstance(obj, np.ndarray):
        return (np.atleast_1d, np.atleast_2d)[depth - 1](obj)
    levels, probe = 0, obj
    while levels < depth and isinstance(probe, collections.abc.Sequence):
        levels += 1
        probe = probe[0] if len(probe) else None
    for _ in range(depth - levels):
        obj = [obj]
    return obj


def atleast_1d(obj):
    return _nested(obj, 1)


def atleast_2d(obj):
    return _nested(obj, 2)


class PizzaSlicer(Block):
    """Phase in [0, TAU) -> step numbers.  Sends every step from the one after the last step
    sent up to the one the phase is in now, around the circle if need be, so no step is lost
    when the phase jumps over several (sequencer.py:96-128)."""

    def __init__(self, nSlices):
        super().__init__(nInputs=1)
        self.outputs = [MessageOutput(owner=self)]
        self.nSlices = nSlices
        self.currentIdx = -1

    def increment(self):
        self.currentIdx = (self.currentIdx + 1) % self.nSlices

    def update(self):
        behind = (pizza_slice_number(self.input.value, self.nSlices) - self.currentIdx) % self.nSlices
        while behind:
            self.increment()
            self.output.send(self.currentIdx)
            behind -= 1


class PatternLookup(_MessageBlock):
    """Step number -> Note(s) of the pattern; a 0 / None / empty entry is a rest, a list entry a
    chord (sequencer.py:131-151)."""

    def __init__(self, pattern):
        super().__init__()
        self.pattern = pattern

    def pitches(self, step):
        return [pitch for pitch in atleast_1d(self.pattern[step]) if pitch]

    def update(self):
        for step in self.input.receive():
            for pitch in self.pitches(step):
                self.output.send(Note(pitch=pitch, velocity=1.))


class Sequence(Composite):
    """One sequencer channel (sequencer.py:154-243): a Phasor turns once per pattern, a
    PizzaSlicer cuts the turn into steps, a PatternLookup turns steps into notes and a
    NoteLengthener ends them.  Between phasor and slicer the phase leaves the composite
    through `phaseOut` and comes back through `phaseIn` -- normally straight, or through a
    micro rhythm patched in between."""

    def __init__(self, sequencer, pattern, tempo=120, beatValue=QUARTER_NOTE,
                 grid=SIXTEENTH_NOTE, relNoteLength=.5, absNoteLength=None):
        super().__init__()
        self.sequencer = sequencer
        self.pattern = atleast_1d(pattern)
        self.tempo, self.beatValue, self.grid = tempo, beatValue, grid
        noteLength = absNoteLength
        if noteLength is None:
            noteLength = relNoteLength * self.duration / self.length
        # ports
        self.phaseIn, self.phaseOut = Relay(owner=self), Relay(owner=self)
        self.inputs = [self.phaseIn]
        self.outputs = [MessageRelay(owner=self), self.phaseOut]
        # inner blocks, in signal order
        self.phasor = Phasor(frequency=1. / self.duration)
        self.pizza = PizzaSlicer(self.length)
        self.lookup = PatternLookup(self.pattern)
        self.lengthener = NoteLengthener(noteLength)
        self.phasor | self.phaseOut
        self._bridge()
        stage = self.phaseIn
        for nxt in (self.pizza, self.lookup, self.lengthener, self.output):
            stage = stage | nxt
        self.update_internal_exec_order(self.phasor, self.pizza)

    length = property(lambda self: len(self.pattern))

    @property
    def duration(self):
        """Seconds one pass through the pattern takes."""
        beats = self.length * self.grid / self.beatValue
        return beats * tempo_2_period(self.tempo)

    def _bridge(self):
        """phaseOut straight into phaseIn."""
        self.phaseOut.connect(self.phaseIn)

    def disconnect_phase_insert(self):
        """Whatever sits between phaseOut and phaseIn is taken out on both sides."""
        for sink in list(self.phaseOut.outgoingConnections):
            self.phaseOut.disconnect(sink)
        source = self.phaseIn.incomingConnection
        if source is not None:
            source.disconnect(self.phaseIn)

    def _reorder(self):
        self.sequencer.update_internal_exec_order()

    def apply_micro_rhythm(self, microRhythm):
        self.disconnect_phase_insert()
        self.phaseOut | microRhythm | self.phaseIn
        self._reorder()

    def reset_micro_rhythm(self):
        self.disconnect_phase_insert()
        self._bridge()
        self._reorder()


class Sequencer(Composite):
    """Pattern sequencer (sequencer.py:246-301): one Sequence per pattern row.  With
    splitOutputs every row sends on a message output of its own; without, all rows go
    through one MessageMixer to a single output.  Further arguments go to every Sequence."""

    def __init__(self, pattern, *args, splitOutputs=True, **kwargs):
        super().__init__()
        self.pattern = atleast_2d(pattern)
        self.splitOutputs = splitOutputs
        self.sequences = []
        if not splitOutputs:
            self.outputs = [MessageRelay(owner=self)]
            self.msgMixer = MessageMixer(nInputs=0)
            self.msgMixer | self.output
        for row in self.pattern:
            self.add_new_channel(row, *args, **kwargs)

    def _sink_for_new_row(self):
        """Where the next row's notes go: a new output relay, or a new mixer input."""
        if self.splitOutputs:
            self.outputs.append(MessageRelay(owner=self))
            return self.outputs[-1]
        self.msgMixer.add_new_channel()
        return self.msgMixer.inputs[-1]

    def add_new_channel(self, *args, **kwargs):
        row = Sequence(self, *args, **kwargs)
        self.sequences.append(row)
        row | self._sink_for_new_row()
        self.update_internal_exec_order()

    nChannels = property(lambda self: len(self.pattern))

    def apply_micro_rhythm(self, microRhythm, channel):
        self.sequences[channel].apply_micro_rhythm(microRhythm)

    def reset_micro_rhythm(self, channel):
        self.sequences[channel].reset_micro_rhythm()


# ---- arpeggiator (klang/arpeggiator.py) ----------------------------------------------------

VALID_ORDERS = ['up', 'down', 'upDown', 'downUp', 'alternating', 'random']


def interleave(first, second):
    """a0 b0 a1 b1 ..., then whatever the longer one has left (arpeggiator.py:31-43)."""
    n = min(len(first), len(second))
    out = [item for pair in zip(first[:n], second[:n]) for item in pair]
    return out + list(first[n:]) + list(second[n:])


def arpeggio_index_permutation(order, length):
    """The order in which `length` sorted notes are played (arpeggiator.py:46-80)."""
    if order not in VALID_ORDERS:
        raise ValueError('Invalid order %r! Either: %s.'
                         % (order, ', '.join(map(repr, VALID_ORDERS))))
    if length == 1:
        return [0]
    up = list(range(length))
    down = up[::-1]
    if order == 'up':
        return up
    if order == 'down':
        return down
    if order == 'upDown':
        return up[:-1] + down[:-1]
    if order == 'downUp':
        return down[:-1] + up[:-1]
    if order == 'alternating':     # lowest, highest, second lowest, second highest ...
        half = int(math.ceil(.5 * length))
        return interleave(up[:half], up[half:][::-1])
    permutation = list(range(length))
    random.shuffle(permutation)
    return permutation


class Arpeggio(collections.abc.Sequence):
    """The held notes, sorted, and a cursor into the current order's permutation
    (arpeggiator.py:83-163)."""

    def __init__(self, order='up', initialNotes=None):
        arpeggio_index_permutation(order, 1)   # validates the order
        self.order = order
        self.notes = []
        self.permutation = []
        self.current = 0
        for note in initialNotes or ():
            self.add_note(note)

    def wrap_current(self):
        self.current = self.current % len(self.permutation) if self.permutation else 0

    def update_state(self):
        self.permutation = arpeggio_index_permutation(self.order, len(self))
        self.wrap_current()

    def add_note(self, note):
        if note not in self:
            bisect.insort(self.notes, note)
            self.update_state()

    def remove_note(self, note):
        for held in self.notes:
            if held.pitch == note.pitch:
                self.notes.remove(held)
                self.update_state()
                return
        raise ValueError('No note with pitch %d in Arpeggio' % note.pitch)

    def __getitem__(self, index):
        return self.notes[index]

    def __len__(self):
        return len(self.notes)

    def __contains__(self, note):
        return any(held.pitch == note.pitch for held in self.notes)

    def __next__(self):
        if not self.notes:
            raise StopIteration
        index = self.permutation[self.current]
        self.current += 1
        self.wrap_current()
        return self.notes[index]


class Arpeggiator(Composite):
    """Plays the held notes one after the other, one per `interval`: whenever the internal
    phasor wraps, the previous note is silenced and the next one sent
    (arpeggiator.py:166-194)."""

    def __init__(self, interval=.1, *args, **kwargs):
        super().__init__()
        interval = compute_duration(interval)
        self.inputs = [MessageInput(owner=self)]
        self.outputs = [MessageOutput(owner=self)]
        self.prevPhase = 0.
        self.prevNote = None
        self.phasor = Phasor(1. / interval)
        self.arpeggio = Arpeggio(*args, **kwargs)

    def update(self):
        for note in self.input.receive():
            if note.on:
                self.arpeggio.add_note(note)
            elif note in self.arpeggio:
                self.arpeggio.remove_note(note)
        self.phasor.update()
        phase = self.phasor.output.value
        if phase <= self.prevPhase:
            if self.prevNote:
                self.output.send(self.prevNote.silence())
            if self.arpeggio:
                self.prevNote = next(self.arpeggio)
                self.output.send(self.prevNote)
        self.prevPhase = phase
