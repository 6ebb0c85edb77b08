"""The reference's micro rhythm demo (examples/micro_rhythm_demo.py there): one Sequencer plays
a kick drum row and a straight sixteenth hi-hat row; the hi-hat row's bar phase runs through
a MicroRhyhtm (eighth, two sixteenths, eighth) whose phrasing a slow triangle Lfo moves
between straight and grooved.  The micro rhythm is applied after everything else is patched,
so that the Lfo lands in the Sequencer's own execution order."""
import numpy as np

from klang_amd import tempo
from klang_amd.audio import Dac, HiHat, Kick, Lfo, triangle
from klang_amd.sequencing import MicroRhyhtm, Sequencer

BPM = 80
KICK_ROW = [60 if k in (0, 4, 8, 12, 15) else 0 for k in range(16)]
HAT_ROW = [60] * 16
GROOVE = [tempo.EIGHT_NOTE, tempo.SIXTEENTH_NOTE, tempo.SIXTEENTH_NOTE, tempo.EIGHT_NOTE]


def build(rate=.05, seed=11):
    np.random.seed(seed)   # the hi-hat's noise (NumPy's global RNG)
    sequencer = Sequencer([KICK_ROW, HAT_ROW], tempo=BPM)
    kick_row, hat_row = sequencer.outputs
    mixer = (kick_row | Kick()) + (hat_row | HiHat())
    mixer.gains = [1., .2]
    groove = MicroRhyhtm(GROOVE)
    Lfo(frequency=rate, wave_func=triangle) | groove.phrasing
    sequencer.apply_micro_rhythm(groove, channel=1)
    return mixer | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'micro_rhythm_demo.wav', 6., argv, **kwargs)


if __name__ == '__main__':
    main()
