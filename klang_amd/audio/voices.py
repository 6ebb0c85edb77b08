"""Synthesizer voice: note -> gate + frequency + amplitude
(klang/audio/voices.py:28-80)."""
import copy

from klang_amd.composite import Composite
from klang_amd.schedule import execute
from klang_amd.wiring import MessageInput

__all__ = ['Voice']


class Voice(Composite):
    def __init__(self, oscillator, envelope):
        super().__init__(nOutputs=1)
        self.inputs = [MessageInput(owner=self)]
        self.oscillator, self.envelope = oscillator, envelope
        self.amplitude = 0.
        self.currentPitch = 0
        self.update_internal_exec_order(self.oscillator, self.envelope)

    @property
    def active(self):
        return bool(self.input.queue) or self.envelope.active

    def process_incoming_notes(self):
        for note in self.input.receive():
            self.envelope.gate(note.on)
            if note.on:
                self.amplitude = note.velocity
                self.oscillator.frequency.set_value(note.frequency)
                self.currentPitch = note.pitch
            else:
                self.currentPitch = 0

    def update(self):
        self.process_incoming_notes()
        execute(self.execOrder)
        osc = self.oscillator.output.value
        env = self.envelope.output.value
        self.output.set_value(self.amplitude * osc * env)

    def __deepcopy__(self, memo):
        return type(self)(copy.deepcopy(self.oscillator), copy.deepcopy(self.envelope))
