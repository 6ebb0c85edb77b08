"""The reference's tempo-aware effects demo (examples/tempo_aware_effects.py): every rate and
time is a note value relative to the global tempo -- a Tremolo on quarter notes, a low pass
swept over eight whole notes, a StereoDelay on (double) dotted quarters.

The reference drives it with `Oscillator(wave_func=random)`, white noise from NumPy's global
RNG; a sawtooth stands in for it here, so that two bounces of the patch are comparable."""
from klang_amd.audio import (Dac, Filter, Lfo, Oscillator, StereoDelay, Tremolo, sawtooth,
                             triangle)
from klang_amd.constants import STEREO
from klang_amd.tempo import (DOTTED_QUARTER_NOTE, DOUBLE_DOTTED_QUARTER_NOTE, LARGE_NOTE,
                             QUARTER_NOTE)

FEEDBACK = .8
DRY_WET = .4


def build(frequency=110.):
    osc = Oscillator(frequency, wave_func=sawtooth)
    tremolo = Tremolo(rate=QUARTER_NOTE, depth=1., smoothness=0.1)
    lfo = Lfo(LARGE_NOTE, wave_func=triangle, outputRange=(110, 1100))
    fil = Filter()
    delay = StereoDelay(leftTime=DOUBLE_DOTTED_QUARTER_NOTE, rightTime=DOTTED_QUARTER_NOTE,
                        leftFeedback=FEEDBACK, rightFeedback=FEEDBACK, drywet=DRY_WET)
    dac = Dac(nChannels=STEREO)
    osc | tremolo | fil.input | delay | dac
    lfo | fil.frequency
    return dac


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'tempo_aware_effects.wav', 8., argv, **kwargs)


if __name__ == '__main__':
    main()
