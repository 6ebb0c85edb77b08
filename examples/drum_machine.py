"""A kick drum, a hi-hat, a bass line on a monophonic synthesizer and chords on a polyphonic
one into one Mixer -- sources of four different kinds, so the Mixer takes their outputs as
external voices -- with a ring modulator on the bass.  The hi-hat's white noise comes from
NumPy's global RNG (as in the reference), drawn on the host under backend='hip' too;
`build` seeds that RNG to make two bounces of the patch equal."""
import numpy as np

from klang_amd.audio import (ADSR, AR, Dac, HiHat, Kick, Mixer, MonophonicSynthesizer, Oscillator,
                             PolyphonicSynthesizer, RingModulator, Voice, sawtooth, triangle)
from klang_amd.messages import Note

from examples import NoteScript

STEP = 21   # buffers per sixteenth note at ~123 bpm


def build(bars=2, seed=7):
    np.random.seed(seed)
    kick_script, hat_script, bass_script, chord_script = {}, {}, {}, {}
    for bar in range(bars):
        start = bar * 16 * STEP
        for beat in range(4):
            kick_script.setdefault(start + 4 * beat * STEP, []).append(Note(36))
            hat_script.setdefault(start + (4 * beat + 2) * STEP, []).append(Note(42))
        for k, pitch in enumerate([36, 36, 48, 36, 39, 36, 46, 43]):
            on = start + 2 * k * STEP
            bass_script.setdefault(on, []).append(Note(pitch, .9))
            bass_script.setdefault(on + STEP, []).append(Note(pitch, 0.))
        for pitch in ((60, 63, 67), (58, 62, 65))[bar % 2]:
            chord_script.setdefault(start, []).append(Note(pitch, .5))
            chord_script.setdefault(start + 12 * STEP, []).append(Note(pitch, 0.))
    kick = NoteScript(kick_script) | Kick(frequency=45., decay=.5)
    hat = NoteScript(hat_script) | HiHat(decay=.04)
    bass = (NoteScript(bass_script)
            | MonophonicSynthesizer(Voice(Oscillator(wave_func=sawtooth), AR(.005, .15)))
            | RingModulator(frequency=220., rate=1., amount=30., dryWet=.3))
    chords = NoteScript(chord_script) | PolyphonicSynthesizer(
        Voice(Oscillator(wave_func=triangle), ADSR(.2, .3, .7, .5)))
    mixer = Mixer()
    for source, gain in ((kick, 1.), (hat, .3), (bass, .6), (chords, 2.)):
        mixer += source
        mixer.gains[-1] = gain
    return mixer | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'drum_machine.wav', 4., argv, **kwargs)


if __name__ == '__main__':
    main()
