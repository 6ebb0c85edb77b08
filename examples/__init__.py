"""Patches of the reference's examples/ directory rewritten against klang_amd: the same
Block / | / + code, `from klang...` replaced by `from klang_amd...`.  Every module has a
`build()` that returns the patch's last block and a `main()` that bounces it to a WAV file:

    python -m examples.hello_world [out.wav]

Message sources of the reference that are control plane and not part of this package
(Sequencer, MusicalKeyboard) are replaced by `NoteScript`, a host block that sends
pre-written notes at given buffer numbers."""
from klang_amd.block import Block
from klang_amd.wiring import MessageOutput


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


def bounce(build, default_path, duration, argv=None, **kwargs):
    import sys
    from klang_amd.klang import run_klang
    argv = sys.argv[1:] if argv is None else argv
    path = argv[0] if argv else default_path
    frames = run_klang(build(), filepath=path, offline=True, duration=duration, **kwargs)
    print('%s: %d buffers (%.2f s)' % (path, len(frames), len(frames) * 256 / 44100.))
