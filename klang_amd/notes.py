"""Pre-written note events: the form in which a sequence reaches the device ahead of time.

The reference's live message sources (keyboard, MIDI: klang/keyboard.py, klang/midi.py) have
no place in an offline bounce.  A ``NoteScript`` holds the notes of a whole piece by buffer
number; executed as an ordinary block it sends buffer k's notes while buffer k runs, like any
other source, and like the mirrored Sequencer / Arpeggiator (klang_amd.sequencing) it is run
ahead of the audio by klang_amd.klang.preschedule when its notes go to the
device, so that the piece renders in one pass.
"""
from klang_amd.block import Block
from klang_amd.wiring import MessageOutput

__all__ = ['NoteScript']


class NoteScript(Block):
    """Sends `script[k]` (a list of Notes) while buffer k is being executed."""

    def __init__(self, script):
        super().__init__()
        self.outputs = [MessageOutput(owner=self)]
        self.script = {int(k): list(v) for k, v in script.items()}
        self.buffer = 0

    def update(self):
        for note in self.script.get(self.buffer, ()):
            self.output.send(note)
        self.buffer += 1
