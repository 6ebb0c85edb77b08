"""examples/haunting_envelopes.py of the reference, VARIATION 2 (:16-92): 20 sine voices under
looping AR envelopes of different lengths, spread over the stereo field."""
from fractions import Fraction

from klang_amd.audio import AR, Dac, Oscillator, StereoMixer, Voice
from klang_amd.messages import Note

N_VOICES = 20
BASE_FREQUENCY = 440.


def blend(a, b, x):
    return (1. - x) * a + x * b


def build():
    mixer = StereoMixer()
    for i in range(N_VOICES):
        frac = Fraction(i + 2, i + 1)
        duration = 5. + 5. / float(frac)
        voice = Voice(Oscillator(), AR(.02 * duration, .98 * duration, loop=True))
        mixer += voice
        rel = i / (N_VOICES - 1)
        mixer.pannings[-1] = rel if i % 2 else -rel
        voice.input.push(Note(0, blend(1., .5, rel), frequency=BASE_FREQUENCY / float(frac)))
    return mixer | Dac(nChannels=2)


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'haunting_envelopes.wav', 12., argv, **kwargs)


if __name__ == '__main__':
    main()
