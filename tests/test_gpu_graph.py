"""run_klang(backend='hip'): patches written against the Block / | / + API are
compiled into device banks and must reproduce the reference fixtures."""
import wave

import numpy as np
import pytest

from klang_amd import fusion, patches as P
from klang_amd.audio import (ADSR, AR, Dac, Filter, Mixer, Oscillator, Reverb, StereoMixer,
                             Voice, sawtooth, sine)
from klang_amd.errors import KlangError
from klang_amd.klang import run_klang
from klang_amd.messages import Note
from klang_amd.schedule import determine_execution_order
from tests.helpers import report

pytestmark = pytest.mark.gpu
B = 256


def test_hello_world_through_the_device(hip, tmp_path, golden):
    path = str(tmp_path / 'hello.wav')
    osc = Oscillator(frequency=440.)
    frames = run_klang(osc | Dac(), filepath=path, offline=True, duration=1.0)
    audio = np.concatenate(frames)
    assert audio.shape == (172 * B, 1)
    np.testing.assert_allclose(audio[:4 * B, 0].reshape(4, B),
                               golden('hello_world')['first_blocks_f32'], rtol=0, atol=6e-8)
    with wave.open(path, 'rb') as wav:
        assert wav.getnframes() == 44032
    # device state was copied back into the host block
    t = 172 * B / 44100.
    assert abs(osc.currentPhase - (2 * np.pi * 440. * t) % (2 * np.pi)) < 1e-6


def test_haunting_envelopes_patch_through_the_api(hip, golden):
    """examples/haunting_envelopes.py:75-92 rewritten against klang_amd: 20 looping
    AR voices into a StereoMixer into a stereo Dac."""
    g = golden('chains')
    p = P.haunting_patch()
    mixer = StereoMixer()
    for v in range(20):
        voice = Voice(Oscillator(wave_func=sine), AR(p.attack[v], p.release[v], loop=True))
        mixer += voice
        mixer.pannings[-1] = p.pan[v]
        voice.input.push(Note(0, p.velocity[v], frequency=p.frequency[v]))
    dac = mixer | Dac(nChannels=2)
    frames = run_klang(dac, offline=True, duration=130 * B / 44100. + 1e-9)
    audio = np.stack(frames)                    # [blocks][256][2]
    kept = [b for b in g['config2_kept_blocks'] if b < len(frames)]
    want = g['config2_mix'][:len(kept)]         # [k][2][256]
    got = audio[kept].transpose(0, 2, 1)
    assert len(kept) >= 6 and np.abs(got - want).max() <= 1e-6


def test_voice_filter_mixer_patch_with_live_notes(hip, golden):
    """Config-3 mini-patch through the API with notes pushed between buffers
    (lookahead 1): the compiled order has one bank and no host DSP block."""
    g = golden('chains')
    p = P.polyphonic_saw_patch(n_voices=48, n_blocks=40, off_base=6, off_span=12)
    mixer, voices = Mixer(), []
    for v in range(48):
        voice = Voice(Oscillator(wave_func=sawtooth),
                      ADSR(p.attack[v], p.decay[v], p.sustain[v], p.release[v]))
        fil = Filter(frequency=p.cutoff[v])
        voice | fil
        mixer += fil
        voices.append(voice)
    dac = mixer | Dac()
    order, banks = fusion.compile_order(determine_execution_order([dac]))
    assert [type(b).__name__ for b in order] == ['GpuVoiceBank', 'Dac'] and len(banks) == 1
    by_block = {}
    for ev in p.events:
        note = Note(0, float(ev['velocity']) if ev['kind'] == 1 else 0., float(ev['frequency']))
        by_block.setdefault(int(ev['block']), []).append((int(ev['voice']), note))
    out = []
    for b in range(40):
        for v, note in by_block.get(b, ()):
            voices[v].input.push(note)
        for blk in order:
            blk.update()
        out.append(mixer.output.value)
    assert np.abs(np.array(out) - g['config3_mini_mix']).max() <= 1e-6
    banks[0].sync_host_blocks()
    np.testing.assert_array_equal([v.oscillator.currentPhase for v in voices],
                                  g['config3_mini_final_phase'])
    banks[0].close()


def test_reverb_chain_and_parameter_edits(hip, oracle):
    voice = Voice(Oscillator(), AR(.01, .2))
    rv = Reverb(decay=3., dryWet=.7)
    mixer = Mixer()
    voice | rv
    mixer += rv
    dac = mixer | Dac()
    order, banks = fusion.compile_order(determine_execution_order([dac]))
    assert [type(b).__name__ for b in order] == ['GpuVoiceBank', 'Dac']
    voice.input.push(Note(69, 1.))
    out = []
    for b in range(12):
        if b == 6:
            mixer.gains[0] = .5                 # host-side edit is picked up
        for blk in order:
            blk.update()
        out.append(mixer.output.value)
    out = np.concatenate(out)
    bank = oracle.OracleBank(1, True, False, False)
    bank.wave[:] = 0
    bank.set_envelope(0, .01, 0., 1., .2)
    ev = P.make_events(0, 0, P.EV_NOTE_ON, 1., 440.)
    _, dry = bank.render(12, ev, want_voices=True)
    r = oracle.OracleReverb(decay=3., dry_wet=.7)
    want = np.concatenate([r.run(dry[0, b * B:(b + 1) * B]) for b in range(12)])
    want[6 * B:] *= .5
    assert np.abs(out - want).max() <= 1e-6
    banks[0].close()


def test_reverb_chains_under_a_stereo_mixer(hip):
    """(Oscillator | Reverb) x N -> StereoMixer is ONE stereo bank with a Reverb per voice
    (gain * pan per channel after the Reverb, mixer.py:95-104) and equals the host blocks."""
    def build():
        mixer = StereoMixer()
        for k in range(5):
            voice = Voice(Oscillator(wave_func=(sine, sawtooth)[k % 2]), AR(.01, .1))
            rv = Reverb(decay=2., dryWet=.6)
            voice | rv
            mixer += rv
            mixer.gains[-1] = .5 + .2 * k
            mixer.pannings[-1] = -1. + .5 * k
            voice.input.push(Note(60 + 3 * k, .9))
        return mixer | Dac(nChannels=2)
    order, banks = fusion.compile_order(determine_execution_order([build()]))
    assert [type(b).__name__ for b in order] == ['GpuVoiceBank', 'Dac']
    assert banks[0].bank.desc.stereo and banks[0].bank.desc.has_reverb
    for bank in banks:
        bank.close()
    dur = 30 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0.))
    want = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., backend='host'))
    assert got.shape == want.shape == (30 * B, 2) and np.abs(want).max() > .05
    assert np.abs(want[:, 0] - want[:, 1]).max() > 1e-3
    err = np.abs(got - want).max()
    report('chain voice+reverb+stereo mixer', err)
    assert err <= 1e-6


def test_strict_backend_refuses_unfusable_hot_path_blocks(hip):
    from klang_amd.audio import Lfo
    lfo = Lfo(frequency=2., outputRange=(200., 2000.))       # control-rate Lfo ...
    osc = Oscillator(wave_func=sawtooth)
    lfo | osc.frequency                                      # ... into an audio-rate input
    with pytest.raises(KlangError) as e:
        run_klang(osc | Dac(), offline=True, duration=.1)
    assert "backend='hip'" in str(e.value) and 'shape=256' in str(e.value)


def _modulated_patches():
    """Builders of small patches with one modulation source per voice."""
    from klang_amd.audio import FmOscillator, Lfo, PwmOscillator, triangle, square

    def arp_like():   # examples/arpeggiator_demo.py:35-49: Lfo | fil.frequency, voice | fil
        mixer = Mixer()
        for k in range(5):
            lfo = Lfo(frequency=2. + k, wave_func=triangle, outputRange=(100, 3000), startPhase=.2 * k)
            fil = Filter(frequency=220.)
            lfo | fil.frequency
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
            voice | fil
            mixer += fil
            voice.input.push(Note(0, .5 + .1 * k, frequency=110. * (k + 1)))
        return mixer | Dac()

    def fm():
        mixer = Mixer()
        for k in range(4):
            mixer += FmOscillator(220. * (k + 1), intensity=.5 + k, modFrequency=55. + 10 * k,
                                  wave_func=(sine, sawtooth, triangle, square)[k])
        return mixer | Dac()

    def vibrato():
        mixer = Mixer()
        for k in range(4):
            lfo = Lfo(4. + k, wave_func=(sine, triangle)[k % 2], shape=B, outputRange=(430 + k, 450 + k))
            osc = Oscillator(440., wave_func=(sawtooth, sine)[k // 2], startPhase=.1 * k)
            lfo | osc.frequency
            mixer += osc
        return mixer | Dac()

    def pwm():
        mixer = Mixer()
        for k in range(3):
            lfo = Lfo(1. + k, wave_func=triangle, shape=B, outputRange=(.1, .9))
            osc = PwmOscillator(110. * (k + 1))
            lfo | osc.dutyCycle
            mixer += osc
        return mixer | Dac()

    return dict(arp_like=arp_like, fm=fm, vibrato=vibrato, pwm=pwm)


@pytest.mark.parametrize('name', ['arp_like', 'fm', 'vibrato', 'pwm'])
def test_modulated_patches_match_the_host_evaluation(hip, name):
    """backend='hip' fuses the Lfo / modulator into the bank (no host DSP block is left)
    and reproduces the explicit NumPy evaluation of the same patch."""
    build = _modulated_patches()[name]
    order, banks = fusion.compile_order(determine_execution_order([build()]))
    assert [type(b).__name__ for b in order] == ['GpuVoiceBank', 'Dac']
    for bank in banks:
        bank.close()
    dur = 20 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0.))
    want = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., backend='host'))
    assert got.shape == want.shape and np.abs(want).max() > .05
    if name == 'pwm':   # +-1 edges: float32 vs float64 duty may move one by a sample
        assert np.mean(np.abs(got - want) > 1e-5) <= 2e-3
    else:
        assert np.abs(got - want).max() <= 1e-5


def test_modulator_state_is_copied_back_to_the_host_blocks(hip):
    from klang_amd.audio import Lfo, triangle
    lfo = Lfo(frequency=3., wave_func=triangle, outputRange=(100, 3000))
    fil = Filter()
    lfo | fil.frequency
    osc = Oscillator(220., wave_func=sawtooth)
    run_klang(osc | fil | Dac(), offline=True, duration=10 * B / 44100. + 1e-9)
    twin = Lfo(frequency=3., wave_func=triangle, outputRange=(100, 3000))
    for _ in range(10):
        twin.update()
    assert lfo.currentPhase == twin.currentPhase


def _effect_patches():
    from klang_amd.audio import (Delay, Gain, Lfo, OctaveDistortion, TanhDistortion, Transformer,
                                 Tremolo, triangle)

    def arpeggiator_like():   # examples/arpeggiator_demo.py:29-49: voice | fil | delay, Lfo on the cutoff
        mixer = Mixer()
        for k in range(3):
            lfo = Lfo(frequency=1. + k, wave_func=triangle, outputRange=(100, 3000))
            fil = Filter(frequency=220.)
            lfo | fil.frequency
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
            voice.input.push(Note(0, .8, frequency=110. * (k + 1)))
            mixer += voice | fil | Delay(time=.013 + .004 * k, feedback=.9)
        mixer.gains = [1., .3, .5]
        return mixer | Dac()

    def tremolo_chain():      # examples/tempo_aware_effects.py style
        osc = Oscillator(220., wave_func=sawtooth)
        return osc | Tremolo(rate=6., depth=.7, smoothness=.3) | TanhDistortion(2.) | Gain(.5) | Dac()

    def shaping():
        mixer = Mixer()
        for k in range(4):
            mixer += (Oscillator(110. * (k + 1), wave_func=sine) | OctaveDistortion()
                      | Transformer(.5, .1 * k) | Filter(frequency=800. + 300 * k))
        return mixer | Dac()

    def lofi():               # Bitcrusher / Subsampler / RingModulator in voice chains
        from klang_amd.audio import Bitcrusher, RingModulator, Subsampler
        mixer = Mixer()
        for k in range(3):
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
            voice.input.push(Note(0, .9, frequency=110. * (k + 1)))
            mixer += (voice | RingModulator(frequency=300. + 100 * k, rate=3., amount=30., dryWet=.7)
                      | Bitcrusher(4 + k) | Subsampler(2 ** (k + 1)))
        return mixer | Dac()

    def stereo_stages():      # effect stages under a StereoMixer (stereo tiles, 32-sample chunks)
        from klang_amd.audio import Bitcrusher, RingModulator
        mixer = StereoMixer()
        for k in range(5):
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
            voice.input.push(Note(0, .7, frequency=82.4 * (k + 1)))
            mixer += (voice | Filter(frequency=600. + 400 * k) | Tremolo(rate=2. + k, depth=.6)
                      | Delay(time=.011 + .003 * k, feedback=.5) | RingModulator(200., 1., 10., dryWet=.2)
                      | Bitcrusher(3) | Gain(.9))
            mixer.pannings[-1] = -1. + .5 * k
        return mixer | Dac(nChannels=2)

    def stages_into_reverb():  # stages in front of a Reverb (time-major float64 bus output)
        mixer = Mixer()
        for k in range(3):
            voice = Voice(Oscillator(wave_func=sine), ADSR(.01, .05, .6, .04))
            voice.input.push(Note(0, .7, frequency=110. * (k + 2)))
            mixer += (voice | TanhDistortion(1.5 + k) | Delay(time=.02, feedback=.3)
                      | Reverb(decay=1., dryWet=.5))
        return mixer | Dac()

    def patched_parameters():  # effect parameters driven at block rate by host Lfos
        # (effects.py:67-70, 117-133: Gain.gain, Tremolo.rate / depth are value inputs)
        mixer = Mixer()
        for k in range(3):
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
            voice.input.push(Note(0, .8, frequency=98. * (k + 1)))
            trem, gain = Tremolo(rate=4., depth=.5, smoothness=.4), Gain(.5)
            Lfo(frequency=.7 + k, outputRange=(2., 9.)) | trem.rate
            Lfo(frequency=1.3, wave_func=triangle, outputRange=(.2, .9)) | trem.depth
            Lfo(frequency=2. + k, outputRange=(.1, 1.)) | gain.gain
            mixer += voice | Filter(frequency=900.) | trem | gain
        return mixer | Dac()

    def short_delays():       # delay lines shorter than one buffer (comb / flanger territory)
        mixer = Mixer()
        for k in range(4):
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
            voice.input.push(Note(0, .8, frequency=123.5 * (k + 1)))
            mixer += voice | Filter(frequency=1200.) | Delay(time=(.0004, .002, .0057, .02)[k], feedback=.7) | Gain(.7)
        return mixer | Dac()

    return dict(arpeggiator_like=arpeggiator_like, tremolo_chain=tremolo_chain, shaping=shaping,
                lofi=lofi, stereo_stages=stereo_stages, stages_into_reverb=stages_into_reverb,
                patched_parameters=patched_parameters, short_delays=short_delays)


@pytest.mark.parametrize('name', ['arpeggiator_like', 'tremolo_chain', 'shaping', 'lofi',
                                  'stereo_stages', 'stages_into_reverb', 'patched_parameters',
                                  'short_delays'])
def test_effect_patches_match_the_host_evaluation(hip, name):
    build = _effect_patches()[name]
    order, banks = fusion.compile_order(determine_execution_order([build()]))
    names = [type(b).__name__ for b in order]
    if name == 'patched_parameters':   # the control Lfos stay host blocks, in front of the bank
        assert names == 9 * ['Lfo'] + ['GpuVoiceBank', 'Dac']
    else:
        assert names == ['GpuVoiceBank', 'Dac']
    for bank in banks:
        bank.close()
    dur = 24 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0.))
    want = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., backend='host'))
    assert got.shape == want.shape and np.abs(want).max() > .05
    assert np.abs(got - want).max() <= 1e-5


def test_block_rate_patched_parameters_vs_reference_fixture(hip, golden):
    """The reference's own output for host Lfos patched into Tremolo.rate / depth and Gain.gain
    (tests/golden/effects_patched.npz, tools/gen_golden.py) vs the device path: the Lfos stay
    host blocks, the bank follows their values buffer by buffer."""
    from tests.test_host_blocks import patched_parameter_patch
    want = golden('effects_patched')['patched_parameters']
    dur = len(want) * B / 44100. + 1e-9
    got = np.concatenate(run_klang(patched_parameter_patch() | Dac(), offline=True, duration=dur,
                                   fadeout=0.))
    assert got.shape == (want.size, 1) and np.abs(want).max() > .05
    from tests.helpers import report
    assert report('effect parameters patched at block rate', np.abs(got[:, 0] - want.ravel()).max()) <= 1e-5


def test_patched_parameters_ignore_an_explicit_lookahead(hip, golden):
    """run_klang(lookahead=8) on a patch whose effect parameters are patched at block rate: the
    reference changes the value every buffer, so the bank must keep rendering one buffer per
    call (a bank rendering 8 ahead would hold the first buffer's value for all 8)."""
    from tests.test_host_blocks import patched_parameter_patch
    want = golden('effects_patched')['patched_parameters']
    dur = len(want) * B / 44100. + 1e-9
    order, banks = fusion.compile_order(determine_execution_order([patched_parameter_patch() | Dac()]))
    assert [b.patched_params for b in banks] == [True]
    for bank in banks:
        bank.close()
    got = np.concatenate(run_klang(patched_parameter_patch() | Dac(), offline=True, duration=dur,
                                   fadeout=0., lookahead=8))
    assert np.abs(got[:, 0] - want.ravel()).max() <= 1e-5
    # a Gain alone, Lfo into Gain.gain, against the explicit host evaluation
    from klang_amd.audio import Gain, Lfo

    def build():
        gain = Gain(.5)
        Lfo(frequency=3., outputRange=(.1, 1.)) | gain.gain
        return Oscillator(220., wave_func=sawtooth) | gain | Dac()
    dur = 24 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., lookahead=8))
    host = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., backend='host'))
    assert np.abs(got - host).max() <= 1e-5 and np.abs(host).max() > .05


def test_host_side_gain_edits_reach_the_device(hip):
    from klang_amd.audio import Gain
    gain = Gain(1.)
    dac = Oscillator(220., wave_func=sawtooth) | gain | Dac()
    order, banks = fusion.compile_order(determine_execution_order([dac]))
    out = []
    for b in range(6):
        if b == 3:
            gain.gain.set_value(.25)
        for blk in order:
            blk.update()
        out.append(np.array(gain.output.value))
    banks[0].close()
    assert abs(np.abs(out[2]).max() - 1.) < 1e-2 and abs(np.abs(out[4]).max() - .25) < 1e-2


@pytest.mark.parametrize('stereo,key', [(True, 'stereo_delay'), (False, 'stereo_delay_mono_in')])
def test_stereo_delay_is_opened_and_its_delays_run_on_the_device(hip, golden, stereo, key):
    """StereoDelay (effects.py:210-236) is a Composite of splitter, two Delays and combiner:
    the compiler opens it, each Delay becomes a device chain fed from its splitter output, and
    the stereo frames equal what the real reference's StereoDelay produced."""
    from klang_amd.audio import AudioCombiner, StereoDelay

    def build():
        left, right = Oscillator(frequency=220., wave_func=sawtooth), Oscillator(311.)
        sd = StereoDelay(leftTime=.05, rightTime=.08, leftFeedback=.5, rightFeedback=.3, drywet=.4)
        if stereo:
            comb = AudioCombiner(nInputs=2)
            left | comb.inputs[0]
            right | comb.inputs[1]
            comb | sd
        else:
            left | sd
        return sd | Dac(nChannels=2)

    order, banks = fusion.compile_order(determine_execution_order([build()]))
    names = [type(b).__name__ for b in order]
    assert 'StereoDelay' not in names and 'Delay' not in names
    assert names[-5:] == ['AudioSplitter', 'GpuVoiceBank', 'GpuVoiceBank', 'AudioCombiner', 'Dac']
    for bank in banks:
        bank.close()
    want = golden('effects')[key]                       # [blocks][2][256]
    dur = len(want) * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0.))
    want = want.transpose(0, 2, 1).reshape(-1, 2)
    assert got.shape == want.shape
    assert report('StereoDelay (two device delay lines)', np.abs(got - want).max()) <= 1e-5


def _stereo_signal_patches():
    from klang_amd.audio import (AudioCombiner, Bitcrusher, Gain, StereoDelay, TanhDistortion,
                                 Tremolo)

    def stereo_source():
        comb = AudioCombiner(nInputs=2)
        Oscillator(220., wave_func=sawtooth) | comb.inputs[0]
        Oscillator(331., wave_func=sine) | comb.inputs[1]
        return comb

    def stages_on_stereo():     # every channel of a stereo buffer runs through the stages
        return (stereo_source() | Tremolo(rate=5., depth=.7, smoothness=.4) | Filter(frequency=900.)
                | TanhDistortion(1.5) | Gain(.8) | Dac(nChannels=2))

    def stereo_into_stereo_mixer():   # aeolian_arp.py:45-98: StereoDelay / Tremolo outputs + mono
        mixer = StereoMixer()
        mixer += (Oscillator(110., wave_func=sawtooth)
                  | StereoDelay(leftTime=.03, rightTime=.05, leftFeedback=.6, rightFeedback=.4))
        mixer += Oscillator(440.) | Bitcrusher(6)
        mixer += stereo_source() | Tremolo(rate=3., depth=1., smoothness=.01)
        mixer.gains = [2., .2, 1.]
        mixer.pannings = [0., -.5, .3]
        return mixer | Dac(nChannels=2)

    def single_stereo_input():        # one input: no division (mixer.py:101-102)
        mixer = StereoMixer()
        mixer += stereo_source() | Gain(.5)
        mixer.pannings = [.4]
        return mixer | Dac(nChannels=2)

    return dict(stages_on_stereo=stages_on_stereo, stereo_into_stereo_mixer=stereo_into_stereo_mixer,
                single_stereo_input=single_stereo_input)


@pytest.mark.parametrize('name', ['stages_on_stereo', 'stereo_into_stereo_mixer',
                                  'single_stereo_input'])
def test_stereo_signals_through_device_chains_and_mixers(hip, name):
    """A stereo buffer fed into effect stages runs as one device chain per channel; a
    StereoMixer takes stereo inputs as two hard-panned voices with the pan amplitudes in their
    gains (mixer.py:95-99).  Against the host evaluation of the same blocks."""
    build = _stereo_signal_patches()[name]
    dur = 40 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0.))
    want = np.concatenate(run_klang(build(), offline=True, duration=dur, fadeout=0., backend='host'))
    assert got.shape == want.shape == (40 * B, 2) and np.abs(want).max() > .05
    assert report('stereo signal: ' + name, np.abs(got - want).max()) <= 1e-5


# ---- synthesizers -------------------------------------------------------
def _synth_template(g):
    return Voice(Oscillator(wave_func=sawtooth), ADSR(*[float(x) for x in g['template']]))


def _play_compiled(order, inputs, rows_per_input, n_blocks, out_block):
    by_block = {}
    for blk_in, rows in zip(inputs, rows_per_input):
        for blk, pitch, velocity, frequency in rows:
            by_block.setdefault(int(blk), []).append((blk_in, Note(int(pitch), velocity, frequency)))
    out = []
    for b in range(n_blocks):
        for blk_in, note in by_block.get(b, ()):
            blk_in.input.push(note)
        for blk in order:
            blk.update()
        out.append(np.array(out_block.output.value, dtype=float))
    return np.array(out)


def test_polyphonic_synthesizer_through_the_api(hip, golden):
    from klang_amd.audio import PolyphonicSynthesizer
    g = golden('synthesizer')
    poly = PolyphonicSynthesizer(_synth_template(g))
    dac = poly | Dac()
    order, banks = fusion.compile_order(determine_execution_order([dac]))
    assert [type(b).__name__ for b in order] == ['GpuVoiceBank', 'Dac'] and banks[0].poly == 24
    got = _play_compiled(order, [poly], [g['poly24_notes']], len(g['poly24_out']), poly)
    assert np.abs(got - g['poly24_out']).max() <= 1e-6
    banks[0].sync_host_blocks()
    state = g['poly24_state']
    assert [v.currentPitch for v in poly.voices] == list(state[:, 1].astype(int))
    assert [v.oscillator.currentPhase for v in poly.voices] == list(state[:, 0])
    assert [v.amplitude for v in poly.voices] == list(state[:, 2])
    assert poly.allocated == int((g['poly24_notes'][:, 2] > 0).sum())
    banks[0].close()


def test_mixer_of_synthesizers_is_one_bank(hip, golden):
    """Two 3-voice synthesizers and gains into one Mixer: one device bank of 6 voices;
    mixer.py:41-49 over synthesizer.py:211-219."""
    from klang_amd.audio import PolyphonicSynthesizer
    g = golden('synthesizer')
    Poly3 = type('Poly3', (PolyphonicSynthesizer,), {'MAX_VOICES': 3})
    a, b = Poly3(_synth_template(g)), Poly3(_synth_template(g))
    mixer = Mixer()
    mixer += a
    mixer += b
    mixer.gains[:] = [.7, 1.3]
    order, banks = fusion.compile_order(determine_execution_order([mixer | Dac()]))
    assert [type(x).__name__ for x in order] == ['GpuVoiceBank', 'Dac'] and len(banks) == 1
    n = len(g['poly3_out'])
    shifted = g['poly3_notes'].copy()
    shifted[:, 0] += 3                                   # the second one starts 3 buffers later
    got = _play_compiled(order, [a, b], [g['poly3_notes'], shifted], n, mixer)
    late = np.concatenate([np.zeros((3, B)), g['poly3_out']])[:n]
    want = (.7 * g['poly3_out'] + 1.3 * late) / 2
    assert np.abs(got - want).max() <= 1e-6 and np.abs(want).max() > .05
    banks[0].close()


def test_stereo_mixer_of_synthesizers(hip, golden):
    """Two 3-voice synthesizers panned left and right under
    a StereoMixer (the gated fast kernel's stereo form): each channel is the fixture's
    synthesizer output times its gain and constant-power pan amplitude."""
    from klang_amd.audio import PolyphonicSynthesizer
    from klang_amd.audio.mixers import panning_amplitudes
    g = golden('synthesizer')
    Poly3 = type('Poly3', (PolyphonicSynthesizer,), {'MAX_VOICES': 3})
    a, b = Poly3(_synth_template(g)), Poly3(_synth_template(g))
    mixer = StereoMixer()
    mixer += a
    mixer += b
    mixer.gains[:] = [.9, 1.1]
    mixer.pannings[:] = [-.6, .8]
    order, banks = fusion.compile_order(determine_execution_order([mixer | Dac(nChannels=2)]))
    assert [type(x).__name__ for x in order] == ['GpuVoiceBank', 'Dac'] and len(banks) == 1
    n = len(g['poly3_out'])
    shifted = g['poly3_notes'].copy()
    shifted[:, 0] += 2
    got = _play_compiled(order, [a, b], [g['poly3_notes'], shifted], n, mixer)   # [n][2][256]
    late = np.concatenate([np.zeros((2, B)), g['poly3_out']])[:n]
    pa = np.ravel(panning_amplitudes(-.6, mixer.mode, mixer.panLaw))
    pb = np.ravel(panning_amplitudes(.8, mixer.mode, mixer.panLaw))
    want = np.stack([(.9 * pa[c] * g['poly3_out'] + 1.1 * pb[c] * late) / 2 for c in (0, 1)], axis=1)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6 and np.abs(want).max() > .05
    banks[0].close()


@pytest.mark.parametrize('policy', ['newest', 'lowest'])
def test_monophonic_synthesizer_through_the_api(hip, golden, policy):
    from klang_amd.audio import MonophonicSynthesizer
    g = golden('synthesizer')
    mono = MonophonicSynthesizer(_synth_template(g), policy=policy)
    order, banks = fusion.compile_order(determine_execution_order([mono | Dac()]))
    assert [type(b).__name__ for b in order] == ['GpuVoiceBank', 'Dac']
    want = g['mono_%s_out' % policy]
    got = _play_compiled(order, [mono], [g['mono_notes']], len(want), mono)
    assert np.abs(got - want).max() <= 1e-6
    banks[0].sync_host_blocks()
    assert mono.voice.oscillator.currentPhase == g['mono_%s_state' % policy][0]
    banks[0].close()


# ---- notes known ahead of time (event pre-computation) ---------------------
def test_note_scripts_are_prescheduled_for_an_offline_bounce(hip, golden, monkeypatch):
    """NoteScript | PolyphonicSynthesizer | Dac, offline: all notes reach the device before the
    first buffer and the piece is ONE device bounce; same samples as the fixture of the real
    reference (which got the notes buffer by buffer)."""
    from klang_amd.audio import PolyphonicSynthesizer
    from klang_amd.notes import NoteScript
    g = golden('synthesizer')
    script = {}
    for blk, pitch, velocity, frequency in g['poly24_notes']:
        script.setdefault(int(blk), []).append(Note(int(pitch), velocity, frequency))
    calls = []
    real = fusion.GpuVoiceBank.bounce
    monkeypatch.setattr(fusion.GpuVoiceBank, 'bounce',
                        lambda self, *a, **kw: (calls.append(1), real(self, *a, **kw))[1])
    source = NoteScript(script)
    poly = PolyphonicSynthesizer(_synth_template(g))
    n = len(g['poly24_out'])
    frames = run_klang(source | poly | Dac(), offline=True, duration=(n + .5) * B / 44100.)
    assert calls == [1] and source.buffer == n
    got = np.concatenate(frames)[:, 0].reshape(n, B)
    assert np.abs(got - g['poly24_out']).max() <= 1e-6
    assert [v.currentPitch for v in poly.voices] == list(g['poly24_state'][:, 1].astype(int))
    assert poly.allocated == int((g['poly24_notes'][:, 2] > 0).sum())


def test_prescheduled_mono_synths_and_kick_render_ahead(hip):
    """Three sources of different kinds into one Mixer (the drum machine): prescheduled, the
    banks render 64 buffers ahead; equal to the host evaluation that gets the notes live."""
    import examples.drum_machine as dm
    dur = 150 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(dm.build(), offline=True, duration=dur))
    want = np.concatenate(run_klang(dm.build(), offline=True, duration=dur, backend='host'))
    assert np.abs(got - want).max() <= 1e-6 and np.abs(want).max() > .1
    # live (buffer by buffer) on the device gives the same samples as prescheduled
    live = np.concatenate(run_klang(dm.build(), offline=True, duration=dur, lookahead=1))
    assert np.abs(live - got).max() <= 1e-7


def test_api_patch_at_baseline_size_equals_the_cabi_bank(hip, tmp_path):
    """BASELINE configs[2] through the drop-in API: 65,536 Voice(Oscillator, ADSR) | Filter
    chains built block by block into one Mixer | Dac, run_klang(filepath=...) on the device --
    against the C-ABI bank loaded with the same patch (klang.py:59-90)."""
    import time
    n, nb = 65536, 6
    p = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=2, off_span=3)
    t0 = time.perf_counter()
    mixer = Mixer()
    voices = []
    for v in range(n):
        voice = Voice(Oscillator(wave_func=sawtooth),
                      ADSR(p.attack[v], p.decay[v], p.sustain[v], p.release[v]))
        fil = Filter(frequency=p.cutoff[v])
        voice | fil
        mixer += fil
        voice.input.push(Note(0, p.velocity[v], frequency=p.frequency[v]))
        voices.append(voice)
    dac = mixer | Dac()
    t1 = time.perf_counter()
    # note-offs are not part of this comparison: the notes above are all the patch gets
    path = str(tmp_path / 'baseline.wav')
    frames = run_klang(dac, filepath=path, offline=True, duration=nb * B / 44100. + 1e-9, fadeout=0.)
    t2 = time.perf_counter()
    got = np.concatenate(frames)[:, 0].reshape(nb, B)
    ctx = hip.Context(0)
    bank = hip.Bank(ctx, n, has_env=True, has_filter=True)
    q = P.Patch(p)
    q['events'] = p.events[p.events['kind'] == P.EV_NOTE_ON]
    bank.load_patch(q)
    want = bank.render(nb)[:, 0, :]
    ctx.close()
    assert np.abs(want).max() > 1e-4
    err = np.abs(got - want).max()
    report('API patch, 65,536 chains, vs the C-ABI bank', err)
    assert err <= 1e-6
    with wave.open(path, 'rb') as wav:
        assert wav.getnframes() == nb * B
    print('api 65,536 chains: build %.1f s, run_klang %.1f s' % (t1 - t0, t2 - t1))
