"""The reference's sequencer demo (examples/sequencer_demo.py): a three-row pattern Sequencer
drives a kick drum, a hi-hat and a polyphonic pad with a delay on it.

Every row ends in a device block (the hi-hat's envelope runs on the device, its NumPy RNG
noise is drawn on the host buffer by buffer), so an offline bounce runs the Sequencer ahead of
the audio and hands the whole piece's notes to the device up front
(klang_amd.klang.preschedule); `build(hihat=False)` leaves the hi-hat out."""
import numpy as np

from klang_amd.audio import (AR, Dac, Delay, HiHat, Kick, Oscillator, PolyphonicSynthesizer,
                             Voice)
from klang_amd.sequencing import Sequencer

TEMPO = 120
PATTERN = [
    [60, 0, 0, 60, 0, 0, 60, 0, 60, 0, 0, 60, 0, 60, 0, 60],             # kick drum
    4 * [0, 0, 127, 0],                                                  # hi-hat
    [65, 70, 65, 68, 72, 65, 70, 65, 68, 72, 65, 70, 65, 68, 72, 65],    # pad
]


def build(hihat=True, seed=3):
    np.random.seed(seed)
    rows = PATTERN if hihat else [PATTERN[0], PATTERN[2]]
    seq = Sequencer(rows, tempo=TEMPO, relNoteLength=.8)
    pad = PolyphonicSynthesizer(Voice(envelope=AR(attack=.1, release=.02), oscillator=Oscillator()))
    mixer = seq.outputs[0] | Kick(decay=.8)
    if hihat:
        mixer = mixer + (seq.outputs[1] | HiHat())
    mixer = mixer + (seq.outputs[-1] | pad | Delay(time=.25, feedback=.25))
    mixer.gains = [.7, .1, 1.] if hihat else [.7, 1.]
    return mixer | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'sequencer_demo.wav', 4., argv, **kwargs)


if __name__ == '__main__':
    main()
