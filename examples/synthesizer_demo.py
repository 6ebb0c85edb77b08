"""examples/synthesizer_demo.py of the reference with a polyphonic synthesizer and written-out
chords instead of the computer keyboard: 24 voices, round-robin allocation on the device,
a low-pass filter, a tempo delay and a bit of grit after the synthesizer."""
from klang_amd.audio import (ADSR, Bitcrusher, Dac, Delay, Filter, Oscillator, PolyphonicSynthesizer,
                             Voice, sawtooth)
from klang_amd.messages import Note

from examples import NoteScript

CHORDS = [(48, 55, 64, 67), (45, 52, 60, 64), (41, 53, 57, 60), (43, 55, 59, 62)]


def build(buffers_per_chord=120):
    script = {}
    for k, chord in enumerate(CHORDS * 2):
        start = k * buffers_per_chord
        for j, pitch in enumerate(chord):
            script.setdefault(start + 6 * j, []).append(Note(pitch, .6 + .1 * j))
            script.setdefault(start + buffers_per_chord - 20, []).append(Note(pitch, 0.))
    synthesizer = PolyphonicSynthesizer(Voice(Oscillator(wave_func=sawtooth), ADSR(.05, .2, .6, .4)))
    keys = NoteScript(script) | synthesizer
    return keys | Filter(frequency=1800.) | Delay(time=.25, feedback=.4, drywet=.3) | Bitcrusher(2) | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'synthesizer_demo.wav', 6., argv, **kwargs)


if __name__ == '__main__':
    main()
