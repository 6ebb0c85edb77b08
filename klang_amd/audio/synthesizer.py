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
import itertools
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
    """`number` copies of a voice (synthesizer.py:58-60)."""
    return [copy.deepcopy(voice, memo=None) for _ in range(number)]


class Synthesizer(Block):
    """Base class: one message input, one mono output (synthesizer.py:63-84)."""

    def __init__(self):
        super().__init__(nOutputs=1)
        self.inputs = [MessageInput(owner=self)]
        self.output.set_value(MONO_SILENCE)

    def play_note(self, *notes):
        for note in notes:
            self.process_note(note)

    def process_note(self, note):
        raise NotImplementedError

    def update(self):
        for note in self.input.receive():
            self.process_note(note)


class NoteScheduler:
    """The sounding note-ons of a monophonic synthesizer, ordered by policy
    (synthesizer.py:87-148): a min-heap over the reversed key; a note-off removes
    the first entry of that pitch."""

    POLICIES = ('newest', 'oldest', 'lowest', 'highest')

    def __init__(self, policy='newest'):
        assert policy in self.POLICIES
        self.policy = policy
        self.counter = itertools.count()
        self.notes = []

    def get_key(self, note):
        if self.policy == 'newest':
            key = next(self.counter)
        elif self.policy == 'oldest':
            key = -next(self.counter)
        elif self.policy == 'highest':
            key = note.pitch
        else:
            key = -note.pitch
        return -key

    def add_note(self, note):
        heapq.heappush(self.notes, (self.get_key(note), note))

    def remove_note(self, note):
        for item in self.notes:
            if item[1].pitch == note.pitch:
                self.notes.remove(item)
                heapq.heapify(self.notes)
                return

    def get_next_note(self, note):
        """The note the voice has to play now, given a new note-on / note-off."""
        if note.on:
            self.add_note(note)
        else:
            self.remove_note(note)
        if not self.notes:
            return note
        return self.notes[0][1]


class MonophonicSynthesizer(Synthesizer):
    """One voice; the scheduler decides which held note it plays
    (synthesizer.py:151-181).  The voice is only updated while active."""

    def __init__(self, voice, policy='newest'):
        super().__init__()
        self.voice = voice
        self.noteScheduler = NoteScheduler(policy)

    def process_note(self, note):
        self.voice.input.push(self.noteScheduler.get_next_note(note))

    def update(self):
        super().update()
        samples = MONO_SILENCE
        if self.voice.active:
            self.voice.update()
            samples = self.voice.output.value
        self.output.set_value(samples)


class PolyphonicSynthesizer(Synthesizer):
    """MAX_VOICES copies of a voice, allocated round robin; note-offs go to every
    voice currently playing that pitch; output = sum of the active voices /
    MAX_VOICES (synthesizer.py:184-219)."""

    MAX_VOICES = 24

    def __init__(self, voice):
        super().__init__()
        self.voices = duplicate_voice(voice, self.MAX_VOICES)
        self.freeVoice = itertools.cycle(self.voices)
        self.allocated = 0   # note-ons handed out so far (the cycle's position)

    def process_note(self, note):
        if note.on:
            next(self.freeVoice).input.push(note)
            self.allocated += 1
        else:
            for voice in self.voices:
                if voice.currentPitch == note.pitch:
                    voice.input.push(note)

    def update(self):
        super().update()
        samples = MONO_SILENCE.copy()
        for voice in self.voices:
            if voice.active:
                voice.update()
                samples += voice.output.value
        self.output.set_value(samples / self.MAX_VOICES)


def sample_exponential_decay(decay, t0=0.):
    """exp(-PI / decay * (t0 + T)) as amplitude-at-t0 times the curve over one buffer, and
    the next start time (synthesizer.py:22-36)."""
    amp = math.exp(-PI / decay * t0)
    return amp * np.exp(-PI / decay * T), t0 + DT * BUFFER_SIZE


def sample_pitch_decay(frequency, decay, intensity, t0=0.):
    """synthesizer.py:39-55."""
    env, t1 = sample_exponential_decay(decay, t0)
    return frequency * (1. + intensity * env), t1


class HiHat(Block):
    """White noise under a decay envelope (synthesizer.py:222-250).  The noise is drawn from
    NumPy's global RandomState once per buffer (or once for good with ``loopedNoise``), in the
    reference's order: construction first, then one draw per update."""

    def __init__(self, decay=.05, loopedNoise=False):
        from klang_amd.audio.envelopes import D
        super().__init__(nOutputs=1)
        self.inputs = [MessageInput(self)]
        if loopedNoise:
            noise = 2 * np.random.random(BUFFER_SIZE) - 1.
            self.noise_generator = lambda: noise
        else:
            self.noise_generator = lambda: 2 * np.random.random(BUFFER_SIZE) - 1.
        self.envelope = D(decay)

    def update(self):
        for note in self.input.receive():
            self.envelope.input.push(note)
        self.envelope.update()
        env = self.envelope.output.get_value()
        self.output.set_value(env * self.noise_generator())


class Kick(Block):
    """Kick drum: a sine with exponentially decaying amplitude and pitch, re-started by every
    note-on with pitch > 0 (synthesizer.py:253-283).  As in the reference the next start
    phase is wrap(phase[-1] + TAU * frequency * DT) with the frequency ARRAY (waves.py:61):
    a vector of 256 start phases from the second buffer on."""

    def __init__(self, frequency=40., decay=.8, intensity=2, pitchDecay=.3):
        super().__init__(nInputs=0, nOutputs=1)
        self.frequency, self.decay = frequency, decay
        self.intensity, self.pitchDecay = intensity, pitchDecay
        self.inputs = [MessageInput(self)]
        self.currentTime = 0.
        self.currentPhase = 0.

    def update(self):
        for note in self.input.receive():
            if note.pitch > 0 and note.on:
                self.currentTime = 0.
                self.currentPhase = 0.
        frequency, _ = sample_pitch_decay(self.frequency, self.pitchDecay, self.intensity,
                                          self.currentTime)
        env, self.currentTime = sample_exponential_decay(self.decay, self.currentTime)
        signal, self.currentPhase = sample_wave(frequency, startPhase=self.currentPhase)
        self.output.set_value(env * signal)
