"""Patches of the reference's examples/ directory rewritten against klang_amd: the same
Block / | / + code, `from klang...` replaced by `from klang_amd...`.  Every module has a
`build()` that returns the patch's last block and a `main()` that bounces it to a WAV file:

    python -m examples.hello_world [out.wav]

The reference's Sequencer and Arpeggiator are mirrored (klang_amd.sequencing) and used as they
are; its live sources (MusicalKeyboard, MIDI) are replaced by `klang_amd.notes.NoteScript`, a
block that sends pre-written notes at given buffer numbers.  Under backend='hip' an offline
bounce runs all of these ahead of the audio and hands the whole piece's notes to the device up
front (klang_amd.klang.preschedule)."""
from klang_amd.notes import NoteScript  # noqa: F401  (the examples import it from here)


def bounce(build, default_path, duration, argv=None, **kwargs):
    import sys
    from klang_amd.klang import run_klang
    argv = sys.argv[1:] if argv is None else argv
    path = argv[0] if argv else default_path
    frames = run_klang(build(), filepath=path, offline=True, duration=duration, **kwargs)
    print('%s: %d buffers (%.2f s)' % (path, len(frames), len(frames) * 256 / 44100.))
