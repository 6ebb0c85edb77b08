"""Synthesizer voice (the reference's klang/audio/voices.py:28-80): an oscillator under an
envelope, driven by notes.  A note-on opens the gate, sets the oscillator's frequency and
becomes the voice's amplitude and current pitch; a note-off closes the gate.  The output is
(amplitude * oscillator) * envelope, in that order of operations."""
import copy

import numpy as np

from klang_amd.composite import Composite
from klang_amd.wiring import MessageInput

__all__ = ['Voice']


class Voice(Composite):
    def __init__(self, oscillator, envelope):
        super().__init__(nOutputs=1)
        self.inputs = [MessageInput(owner=self)]
        self.oscillator = oscillator
        self.envelope = envelope
        self.amplitude, self.currentPitch = 0., 0
        self.update_internal_exec_order(oscillator, envelope)

    @property
    def active(self):
        """A note is waiting, or the envelope has not finished."""
        return len(self.input.queue) > 0 or self.envelope.active

    def take(self, note):
        """One note, as it arrives at a buffer boundary (voices.py:55-64)."""
        self.envelope.gate(note.on)
        if note.off:
            self.currentPitch = 0
            return
        self.amplitude, self.currentPitch = note.velocity, note.pitch
        self.oscillator.frequency.set_value(note.frequency)

    def process_incoming_notes(self):
        for note in self.input.receive():
            self.take(note)

    def update(self):
        self.process_incoming_notes()
        Composite.update(self)
        scaled = np.multiply(self.amplitude, self.oscillator.output.value)
        self.output.set_value(np.multiply(scaled, self.envelope.output.value))

    def __deepcopy__(self, memo):
        parts = (copy.deepcopy(part) for part in (self.oscillator, self.envelope))
        return type(self)(*parts)
