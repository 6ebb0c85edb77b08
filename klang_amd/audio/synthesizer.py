"""Synthesizer blocks, host side (klang/audio/synthesizer.py:63-219): note
scheduling for the monophonic synthesizer, round-robin voice allocation for the
polyphonic one.  Under ``backend='hip'`` the graph compiler hands both to a
gated device bank (klang_amd/fusion.py): the polyphonic routing then runs on the
device (klmi_bank_notes), the monophonic NoteScheduler stays here because it is
a sequential heap per instrument.

``Kick`` is mirrored with the reference's behaviour as it is: its
``currentPhase`` turns into a 256-vector after the first buffer
(synthesizer.py:282 with waves.py:61).  ``HiHat``'s white noise comes from NumPy's
global RNG, so it matches the reference exactly when the caller seeds ``np.random`` the same
way; under ``backend='hip'`` its envelope runs on the device and the noise, still drawn here,
goes in as the voice's external input (fusion.GpuHiHat).
"""
import copy
import heapq
import math

import numpy as np

from klang_amd.audio.signals import DT, MONO_SILENCE, T, sample_wave
from klang_amd.config import BUFFER_SIZE
from klang_amd.constants import PI
from klang_amd.block import Block
from klang_amd.wiring import MessageInput

__all__ = ['NoteScheduler', 'Synthesizer', 'MonophonicSynthesizer', 'PolyphonicSynthesizer',
           'HiHat', 'Kick']


def duplicate_voice(voice, number):
    """`number` independent copies of a voice (synthesizer.py:58-60)."""
    return [copy.deepcopy(voice) for _ in range(number)]


class Synthesizer(Block):
    """Base class: one message input, one mono output; every note that arrives in a buffer
    goes through ``process_note`` (synthesizer.py:63-84)."""

    def __init__(self):
        super().__init__(nOutputs=1)
        self.inputs = [MessageInput(owner=self)]
        self.output.set_value(MONO_SILENCE)

    def process_note(self, note):
        raise NotImplementedError

    def play_note(self, *notes):
        """Notes handed over directly instead of through the message input."""
        for note in notes:
            self.process_note(note)

    def update(self):
        self.play_note(*self.input.receive())


# priority of a held note under each policy of the monophonic synthesizer: the note with
# the SMALLEST rank plays (`arrival` counts the note-ons seen so far)
_RANKS = {
    'newest': lambda note, arrival: -arrival,
    'oldest': lambda note, arrival: arrival,
    'highest': lambda note, arrival: -note.pitch,
    'lowest': lambda note, arrival: note.pitch,
}


class NoteScheduler:
    """Which of the held notes a monophonic synthesizer plays (synthesizer.py:87-148).

    Held note-ons live in a binary heap (``heapq``) of (rank, note) entries, the playing note
    on top.  A note-off drops the first entry of its pitch in the heap's own array order --
    which entry that is matters when a pitch is held twice, and the reference's behaviour is
    defined by exactly this container -- and restores the heap."""

    POLICIES = tuple(_RANKS)

    def __init__(self, policy='newest'):
        assert policy in _RANKS
        self.policy = policy
        self.arrivals = 0
        self.notes = []     # the heap

    def hold(self, note):
        heapq.heappush(self.notes, (_RANKS[self.policy](note, self.arrivals), note))
        self.arrivals += 1

    def release(self, pitch):
        position = next((i for i, (_, held) in enumerate(self.notes) if held.pitch == pitch), None)
        if position is not None:
            del self.notes[position]
            heapq.heapify(self.notes)

    def get_next_note(self, note):
        """The note the voice has to play now, given a new note-on / note-off: the top of
        the heap, or the note itself when nothing is held any more."""
        if note.on:
            self.hold(note)
        else:
            self.release(note.pitch)
        return self.notes[0][1] if self.notes else note

    # the reference's names
    def add_note(self, note):
        self.hold(note)

    def remove_note(self, note):
        self.release(note.pitch)


class MonophonicSynthesizer(Synthesizer):
    """One voice; the scheduler decides which held note it plays (synthesizer.py:151-181).
    The voice is only updated while active; otherwise the output is silence."""

    def __init__(self, voice, policy='newest'):
        super().__init__()
        self.voice = voice
        self.noteScheduler = NoteScheduler(policy)

    def process_note(self, note):
        self.voice.input.push(self.noteScheduler.get_next_note(note))

    def update(self):
        super().update()
        if not self.voice.active:
            self.output.set_value(MONO_SILENCE)
            return
        self.voice.update()
        self.output.set_value(self.voice.output.value)


class RoundRobin:
    """Endless iterator over the voices of a polyphonic synthesizer; ``taken`` counts how many
    it has handed out (the device bank is told ``taken % len(voices)``)."""

    def __init__(self, voices):
        self.voices = voices
        self.taken = 0

    def __iter__(self):
        return self

    def __next__(self):
        voice = self.voices[self.taken % len(self.voices)]
        self.taken += 1
        return voice


class PolyphonicSynthesizer(Synthesizer):
    """MAX_VOICES copies of a voice.  A note-on goes to the next voice in turn, whatever it is
    doing; a note-off to every voice that currently plays its pitch; the output is the sum of
    the active voices / MAX_VOICES (synthesizer.py:184-219)."""

    MAX_VOICES = 24

    def __init__(self, voice):
        super().__init__()
        self.voices = duplicate_voice(voice, self.MAX_VOICES)
        self.freeVoice = RoundRobin(self.voices)

    @property
    def allocated(self):
        """Note-ons handed out so far."""
        return self.freeVoice.taken

    def process_note(self, note):
        targets = [next(self.freeVoice)] if note.on else \
            [voice for voice in self.voices if voice.currentPitch == note.pitch]
        for voice in targets:
            voice.input.push(note)

    def update(self):
        super().update()
        bus = MONO_SILENCE.copy()
        for voice in self.voices:
            if voice.active:
                voice.update()
                bus += voice.output.value
        self.output.set_value(bus / self.MAX_VOICES)


INTERVAL = DT * BUFFER_SIZE     # seconds per buffer


def sample_exponential_decay(decay, t0=0.):
    """One buffer of exp(-PI / decay * t) starting at time `t0`, and the time the next buffer
    starts at.  Evaluated as the reference does (synthesizer.py:22-36): the value at t0
    (libm) times the curve over the buffer's own time vector (NumPy)."""
    rate = -PI / decay
    return math.exp(rate * t0) * np.exp(rate * T), t0 + INTERVAL


def sample_pitch_decay(frequency, decay, intensity, t0=0.):
    """A frequency that starts (1 + intensity) times too high and decays to `frequency`
    (synthesizer.py:39-55)."""
    curve, t1 = sample_exponential_decay(decay, t0)
    return frequency * (1. + intensity * curve), t1


class WhiteNoise:
    """Buffers of uniform noise in [-1, 1) from NumPy's GLOBAL RandomState (the stream the
    reference draws from, so a seeded run reproduces it): a fresh buffer per call, or -- with
    `looped` -- one buffer drawn at construction and repeated."""

    def __init__(self, looped=False):
        self.frozen = self.draw() if looped else None

    @staticmethod
    def draw():
        return 2 * np.random.random(BUFFER_SIZE) - 1.

    def __call__(self):
        return self.draw() if self.frozen is None else self.frozen


class HiHat(Block):
    """White noise under a decay envelope (synthesizer.py:222-250); any note that arrives
    goes to the envelope.  Random numbers are drawn in the reference's order: the looped
    buffer at construction, otherwise one buffer per update."""

    def __init__(self, decay=.05, loopedNoise=False):
        from klang_amd.audio.envelopes import D
        super().__init__(nOutputs=1)
        self.inputs = [MessageInput(self)]
        self.noise_generator = WhiteNoise(looped=loopedNoise)
        self.envelope = D(decay)

    def update(self):
        for note in self.input.receive():
            self.envelope.input.push(note)
        self.envelope.update()
        self.output.set_value(self.envelope.output.get_value() * self.noise_generator())


class Kick(Block):
    """Kick drum: a sine with exponentially decaying amplitude and pitch, re-started by every
    note-on with pitch > 0 (synthesizer.py:253-283).  As in the reference the next start
    phase is wrap(phase[-1] + TAU * frequency * DT) with the frequency ARRAY (waves.py:61):
    a vector of 256 start phases from the second buffer on."""

    def __init__(self, frequency=40., decay=.8, intensity=2, pitchDecay=.3):
        super().__init__(nInputs=0, nOutputs=1)
        self.inputs = [MessageInput(self)]
        self.frequency, self.intensity = frequency, intensity
        self.decay, self.pitchDecay = decay, pitchDecay
        self.restart()

    def restart(self):
        self.currentTime, self.currentPhase = 0., 0.

    def update(self):
        if any(note.on and note.pitch > 0 for note in list(self.input.receive())):
            self.restart()
        now = self.currentTime
        sweep, _ = sample_pitch_decay(self.frequency, self.pitchDecay, self.intensity, now)
        loudness, self.currentTime = sample_exponential_decay(self.decay, now)
        tone, self.currentPhase = sample_wave(sweep, startPhase=self.currentPhase)
        self.output.set_value(loudness * tone)
