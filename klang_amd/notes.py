"""Pre-written note events: the form in which a sequence reaches the device ahead of time.

The reference's message sources (Sequencer, Arpeggiator, keyboard: klang/sequencer.py,
klang/arpeggiator.py, klang/keyboard.py) are host control plane and emit their notes one
buffer at a time.  A ``NoteScript`` holds the notes of a whole piece by buffer number, so an
offline bounce can hand all of them to the device banks up front (klang_amd.klang.preschedule,
SURVEY 8f rank 4) and render the piece in one pass; executed as an ordinary block it sends
buffer k's notes while buffer k runs, like any other source.
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
