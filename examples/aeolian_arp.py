"""The reference's "Aeolian Arp" patch (examples/aeolian_arp.py there), rebuilt part by part:

    kick      32-step pattern -> Kick
    hi-hat    24-step pattern -> HiHat -> Reverb -> StereoDelay (dotted eighths) -> Tremolo
    arpeggio  four chords on a long-note grid -> Arpeggiator (random order, eighth triplets)
              -> polyphonic triangle synthesizer -> low pass swept by a slow Lfo -> make-up gain
              -> StereoDelay (eighth / quarter)
    bass      24-step pattern on eighths -> monophonic square synthesizer -> low pass

into one StereoMixer.  Every audio block runs on the device under backend='hip', the stereo
delays' outputs included.  The hi-hat's noise (NumPy RNG) and the arpeggio's random order
(Python's `random`) are seeded in `build`, so that two bounces of the patch are equal."""
import random

import numpy as np

from klang_amd import tempo
from klang_amd.audio import (AR, Dac, Filter, Gain, HiHat, Kick, Lfo, MonophonicSynthesizer,
                             Oscillator, PolyphonicSynthesizer, Reverb, StereoDelay, StereoMixer,
                             Tremolo, Voice, square, triangle)
from klang_amd.sequencing import Arpeggiator, Sequencer

CHORD_PROGRESSION = [(60, 63, 67, 70, 72), (62, 65, 67, 70, 72), (63, 65, 70, 72, 74),
                     (55, 67, 70, 72, 74)]


def steps(length, hits):
    """A step pattern of `length` rests with pitch hits[k] at step k."""
    row = [0] * length
    for k, pitch in hits.items():
        row[k] = pitch
    return row


def synthesizer(kind, attack, release, wave):
    return kind(Voice(Oscillator(wave_func=wave), AR(attack, release)))


def kick_part():
    pattern = steps(32, {k: 60 for k in (0, 8, 14, 16, 24, 31)})
    return Sequencer(pattern) | Kick(frequency=33., decay=1.2, pitchDecay=.8)


def hat_part():
    pattern = steps(24, {k: 60 for k in (2, 4, 11, 13, 20, 22)})
    delay = StereoDelay(leftTime=tempo.DOTTED_EIGHT_NOTE, rightTime=tempo.DOUBLE_DOTTED_EIGHT_NOTE,
                        leftFeedback=.9, rightFeedback=.9, drywet=.5)
    chopper = Tremolo(tempo.EIGHT_NOTE, smoothness=.01, depth=1., dutyCycle=.5)
    return Sequencer(pattern) | HiHat() | Reverb(decay=5., preDelay=.5) | delay | chopper


def arp_part():
    sweep = Lfo(.1, outputRange=(450, 2500))
    lowpass = Filter()
    sweep | lowpass.frequency
    # a relative note length of 1. lets successive chords overlap and cancel the arpeggio
    chords = Sequencer(pattern=[CHORD_PROGRESSION], grid=tempo.LONG_NOTE, relNoteLength=.99)
    arp = Arpeggiator(interval=tempo.EIGHT_NOTE * 2 / 3, order='random')
    pad = synthesizer(PolyphonicSynthesizer, .05, .4, triangle)
    echo = StereoDelay(leftTime=tempo.EIGHT_NOTE, rightTime=tempo.QUARTER_NOTE, leftFeedback=.7,
                       rightFeedback=.8, drywet=.5)
    return chords | arp | pad | lowpass | Gain(gain=PolyphonicSynthesizer.MAX_VOICES) | echo


def bass_part():
    pattern = steps(24, {0: 36, 2: 36, 8: 32, 10: 32, 16: 34, 18: 34})
    return (Sequencer(pattern, grid=tempo.EIGHT_NOTE)
            | synthesizer(MonophonicSynthesizer, .01, 2., square) | Filter(frequency=400))


def build(seed=5):
    np.random.seed(seed)
    random.seed(seed)
    mixer = StereoMixer()
    for part, gain in ((kick_part, 2), (hat_part, .2), (arp_part, 1.), (bass_part, 1.)):
        mixer += part()
        mixer.gains[-1] = gain
    return mixer | Dac(2)


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'aeolian_arp.wav', 8., argv, **kwargs)


if __name__ == '__main__':
    main()
