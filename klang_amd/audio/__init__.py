"""Audio blocks: the public names of the reference's klang.audio that lie on
the DSP hot path (klang/audio/__init__.py star-imports)."""
from klang_amd.audio.signals import (Chirper, FmOscillator, Lfo, Oscillator, Phasor, PwmOscillator, WAVE_FUNCTIONS,
                                     sample_phase, sawtooth, sine, square, triangle)
from klang_amd.audio.envelopes import ADSR, AR, D, R, Envelope
from klang_amd.audio.filtering import (BackwardCombFilter, EchoFilter, Filter,
                                       ForwardCombFilter, Reverb)
from klang_amd.audio.effects import (AudioCombiner, AudioSplitter, Bitcrusher, Delay, Gain,
                                     OctaveDistortion, RingModulator, StereoDelay, Subsampler,
                                     TanhDistortion, Transformer, Tremolo)
from klang_amd.audio.mixers import Mixer, StereoMixer
from klang_amd.audio.voices import Voice
from klang_amd.audio.synthesizer import (HiHat, Kick, MonophonicSynthesizer, NoteScheduler,
                                         PolyphonicSynthesizer, Synthesizer)
from klang_amd.audio.engine import Adc, Dac, run_audio_engine, write_wave

__all__ = ['Chirper', 'FmOscillator', 'Lfo', 'Oscillator', 'Phasor', 'PwmOscillator', 'WAVE_FUNCTIONS', 'sample_phase',
           'sawtooth', 'sine', 'square', 'triangle', 'ADSR', 'AR', 'D', 'R', 'Envelope',
           'BackwardCombFilter', 'EchoFilter', 'Filter', 'ForwardCombFilter', 'Reverb', 'Mixer',
           'StereoMixer', 'Voice', 'HiHat', 'Kick', 'MonophonicSynthesizer', 'NoteScheduler', 'PolyphonicSynthesizer',
           'Synthesizer', 'AudioCombiner', 'AudioSplitter', 'StereoDelay', 'Delay', 'Gain', 'OctaveDistortion', 'TanhDistortion',
           'Transformer', 'Tremolo', 'Bitcrusher', 'RingModulator', 'Subsampler', 'Adc', 'Dac', 'run_audio_engine', 'write_wave']
