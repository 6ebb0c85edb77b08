"""The reference's micro rhythm demo (examples/micro_rhythm_demo.py): a kick drum and a hi-hat
from one Sequencer; the hi-hat row's phase runs through a MicroRhyhtm whose phrasing a slow Lfo
moves between straight and grooved.  apply_micro_rhythm comes last, so that the Lfo lands in
the Sequencer's own execution order."""
import numpy as np

from klang_amd.audio import Dac, HiHat, Kick, Lfo, triangle
from klang_amd.sequencing import MicroRhyhtm, Sequencer
from klang_amd.tempo import EIGHT_NOTE, SIXTEENTH_NOTE

TEMPO = 80
PATTERN = [
    [60, 0, 0, 0, 60, 0, 0, 0, 60, 0, 0, 0, 60, 0, 0, 60],   # kick drum
    4 * [60, 60, 60, 60],                                    # hi-hat
]
MICRO_RHYTHM_NOTES = [EIGHT_NOTE, SIXTEENTH_NOTE, SIXTEENTH_NOTE, EIGHT_NOTE]
RATE = .05


def build(rate=RATE, seed=11):
    np.random.seed(seed)   # the hi-hat's noise (NumPy's global RNG)
    sequencer = Sequencer(PATTERN, tempo=TEMPO)
    mixer = (sequencer.outputs[0] | Kick()) + (sequencer.outputs[1] | HiHat())
    mixer.gains = [1., .2]
    lfo = Lfo(frequency=rate, wave_func=triangle)
    mr = MicroRhyhtm(MICRO_RHYTHM_NOTES)
    lfo | mr.phrasing
    sequencer.apply_micro_rhythm(mr, channel=1)
    return mixer | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'micro_rhythm_demo.wav', 6., argv, **kwargs)


if __name__ == '__main__':
    main()
