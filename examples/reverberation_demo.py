"""examples/reverberation_demo.py of the reference (:11-34): three rhythmically interlaced
monophonic synthesizers into a Mixer, the mix through a Reverb.  The reference's Sequencers
are replaced by pre-written note scripts (same pitches, grids and note lengths at 120 bpm)."""
from klang_amd.audio import AR, Dac, MonophonicSynthesizer, Oscillator, Reverb, Voice
from klang_amd.messages import Note

from examples import NoteScript

BUFFERS_PER_SECOND = 44100 / 256.
QUARTER = .5   # seconds at 120 bpm


def create_synthesizer(attack=.01, release=1.):
    return MonophonicSynthesizer(Voice(Oscillator(), AR(attack, release)))


def pattern_script(pattern, grid, rel_note_length, repeats):
    """pitches on a grid (0 = rest) -> {buffer: [notes]}."""
    script = {}
    for k in range(repeats * len(pattern)):
        pitch = pattern[k % len(pattern)]
        if not pitch:
            continue
        on = int(k * grid * BUFFERS_PER_SECOND)
        off = int((k + rel_note_length) * grid * BUFFERS_PER_SECOND)
        script.setdefault(on, []).append(Note(pitch))
        script.setdefault(max(off, on + 1), []).append(Note(pitch, 0.))
    return script


def build(repeats=2):
    base = NoteScript(pattern_script([60, 62, 67, 69, 60, 62, 67, 69, 60, 62, 67, 71],
                                     QUARTER, .125, repeats)) | create_synthesizer()
    upper = NoteScript(pattern_script([0, 0, 0, 0, 0, 0, 76, 0, 0, 72, 0, 0, 0, 0, 0, 0],
                                      QUARTER / 4, .5, 3 * repeats)) | create_synthesizer()
    lower = NoteScript(pattern_script([53, 60, 55, 60, 57, 60, 62, 60],
                                      .75 * QUARTER, .25, 2 * repeats)) | create_synthesizer()
    mixer = base + upper + lower
    return mixer | Reverb(decay=3., dryWet=.7) | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'reverberation_demo.wav', 12., argv, **kwargs)


if __name__ == '__main__':
    main()
