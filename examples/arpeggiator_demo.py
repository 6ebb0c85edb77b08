"""The reference's arpeggiator demo (examples/arpeggiator_demo.py): an Arpeggiator over six
held notes plays a sawtooth synthesizer through a low pass, whose cutoff a slow Lfo sweeps,
and a long delay; a Sequencer on a whole-note grid plays a sine bass under it."""
from klang_amd.audio import (ADSR, Dac, Delay, Filter, Lfo, MonophonicSynthesizer, Oscillator,
                             Voice, sawtooth, sine, triangle)
from klang_amd.messages import Note
from klang_amd.sequencing import Arpeggiator, Sequencer
from klang_amd.tempo import WHOLE_NOTE


def build_synthesizer(attack=.1, decay=.2, sustain=.3, release=.4, wave_func=sine):
    voice = Voice(Oscillator(wave_func=wave_func), ADSR(attack, decay, sustain, release))
    return MonophonicSynthesizer(voice)


def build(interval=5. / 16):
    arp = Arpeggiator(interval=interval, order='alternating',
                      initialNotes=[Note(pitch=p) for p in (57, 60, 64, 65, 67, 69)])
    synthesizer = build_synthesizer(wave_func=sawtooth)
    lfo = Lfo(frequency=.1, wave_func=triangle, outputRange=(100, 3000))
    fil = Filter(frequency=220.)
    delay = Delay(time=0.76, feedback=.9)
    sequencer = Sequencer(pattern=[[38, 41, 0, 0, 36, 34, 0, 0]], tempo=120, relNoteLength=1.,
                          grid=WHOLE_NOTE)
    bass = build_synthesizer(wave_func=sine, attack=4., sustain=1., release=4.)
    lfo | fil.frequency
    mixer = (arp | synthesizer | fil | delay) + (sequencer | bass)
    mixer.gains = [1., .3]
    return mixer | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'arpeggiator_demo.wav', 6., argv, **kwargs)


if __name__ == '__main__':
    main()
