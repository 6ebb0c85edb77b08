"""The reference's arpeggiator demo (examples/arpeggiator_demo.py there) in two parts:

    lead  six held notes -> Arpeggiator (alternating order) -> monophonic sawtooth synthesizer
          -> low pass whose cutoff a slow triangle Lfo sweeps -> long feedback delay
    bass  one-row Sequencer on a whole-note grid -> monophonic sine synthesizer with a
          four-second attack

mixed 1 : 0.3."""
from klang_amd import tempo
from klang_amd.audio import (ADSR, Dac, Delay, Filter, Lfo, MonophonicSynthesizer, Oscillator,
                             Voice, sawtooth, sine, triangle)
from klang_amd.messages import Note
from klang_amd.sequencing import Arpeggiator, Sequencer

HELD_PITCHES = (57, 60, 64, 65, 67, 69)      # A3 C4 E4 F4 G4 A4
BASS_LINE = [38, 41, 0, 0, 36, 34, 0, 0]


def mono_synth(wave, attack=.1, decay=.2, sustain=.3, release=.4):
    return MonophonicSynthesizer(Voice(Oscillator(wave_func=wave), ADSR(attack, decay, sustain, release)))


def lead_part(interval):
    held = [Note(pitch=p) for p in HELD_PITCHES]
    arp = Arpeggiator(interval=interval, order='alternating', initialNotes=held)
    lowpass = Filter(frequency=220.)
    Lfo(frequency=.1, wave_func=triangle, outputRange=(100, 3000)) | lowpass.frequency
    return arp | mono_synth(sawtooth) | lowpass | Delay(time=0.76, feedback=.9)


def bass_part():
    line = Sequencer(pattern=[BASS_LINE], tempo=120, relNoteLength=1., grid=tempo.WHOLE_NOTE)
    return line | mono_synth(sine, attack=4., sustain=1., release=4.)


def build(interval=5. / 16):
    mixer = lead_part(interval) + bass_part()
    mixer.gains = [1., .3]
    return mixer | Dac()


def main(argv=None, **kwargs):
    from examples import bounce
    bounce(build, 'arpeggiator_demo.wav', 6., argv, **kwargs)


if __name__ == '__main__':
    main()
