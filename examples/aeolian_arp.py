"""The reference's "Aeolian Arp" (examples/aeolian_arp.py): kick and hi-hat patterns, a
chord Sequencer feeding an Arpeggiator in front of a polyphonic synthesizer, a bass line --
four sources into a StereoMixer, with reverb, tempo-synced stereo delays, a tremolo on a stereo
signal and an Lfo on a filter cutoff.  Every audio block runs on the device under
backend='hip'; the hi-hat's noise (NumPy RNG) and the arpeggio's random order (Python's
`random`) are seeded in `build` so that two bounces of the patch are equal."""
import random

import numpy as np

from klang_amd.audio import (AR, Dac, Filter, Gain, HiHat, Kick, Lfo, MonophonicSynthesizer,
                             Oscillator, PolyphonicSynthesizer, Reverb, StereoDelay, StereoMixer,
                             Tremolo, Voice, sine, square, triangle)
from klang_amd.sequencing import Arpeggiator, Sequencer
from klang_amd.tempo import (DOTTED_EIGHT_NOTE, DOUBLE_DOTTED_EIGHT_NOTE, EIGHT_NOTE, LONG_NOTE,
                             QUARTER_NOTE)

CHORDS = [
    (60, 63, 67, 70, 72),
    (62, 65, 67, 70, 72),
    (63, 65, 70, 72, 74),
    (55, 67, 70, 72, 74),
]


def build_synthesizer(cls=MonophonicSynthesizer, attack=.05, release=.4, wave_func=sine):
    return cls(Voice(Oscillator(wave_func=wave_func), AR(attack, release)))


def build(seed=5):
    np.random.seed(seed)
    random.seed(seed)
    mixer = StereoMixer()
    mixer += (Sequencer([60, 0, 0, 0, 0, 0, 0, 0,
                         60, 0, 0, 0, 0, 0, 60, 0,
                         60, 0, 0, 0, 0, 0, 0, 0,
                         60, 0, 0, 0, 0, 0, 0, 60])
              | Kick(frequency=33., decay=1.2, pitchDecay=.8))
    mixer += (Sequencer([0, 0, 60, 0, 60, 0, 0, 0,
                         0, 0, 0, 60, 0, 60, 0, 0,
                         0, 0, 0, 0, 60, 0, 60, 0])
              | HiHat()
              | Reverb(decay=5., preDelay=.5)
              | StereoDelay(leftTime=DOTTED_EIGHT_NOTE, rightTime=DOUBLE_DOTTED_EIGHT_NOTE,
                            leftFeedback=.9, rightFeedback=.9, drywet=.5)
              | Tremolo(EIGHT_NOTE, smoothness=.01, depth=1., dutyCycle=.5))
    lfo = Lfo(.1, outputRange=(450, 2500))
    fil = Filter()
    lfo | fil.frequency
    mixer += (Sequencer(pattern=[CHORDS], grid=LONG_NOTE, relNoteLength=.99)
              | Arpeggiator(interval=EIGHT_NOTE * 2 / 3, order='random')
              | build_synthesizer(PolyphonicSynthesizer, .05, .4, triangle)
              | fil
              | Gain(gain=PolyphonicSynthesizer.MAX_VOICES)
              | StereoDelay(leftTime=EIGHT_NOTE, rightTime=QUARTER_NOTE, leftFeedback=.7,
                            rightFeedback=.8, drywet=.5))
    mixer += (Sequencer([36, 0, 36, 0, 0, 0, 0, 0,
                         32, 0, 32, 0, 0, 0, 0, 0,
                         34, 0, 34, 0, 0, 0, 0, 0], grid=EIGHT_NOTE)
              | build_synthesizer(MonophonicSynthesizer, .01, 2., square)
              | Filter(frequency=400))
    mixer.gains = [2, .2, 1., 1.]
    return mixer | Dac(2)


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'aeolian_arp.wav', 8., argv, **kwargs)


if __name__ == '__main__':
    main()
