"""backend='host' evaluation of the mirrored blocks against the reference
fixtures (tests/golden, made by tools/gen_golden.py), the hello_world config
through run_klang(filepath=...), and the graph compiler's pattern matching
(no device needed for the matching itself)."""
import os
import wave

import numpy as np
import pytest

from klang_amd import fusion, patches as P
from klang_amd.audio import (ADSR, AR, BackwardCombFilter, Dac, EchoFilter, Filter,
                             ForwardCombFilter, Lfo, Mixer, Oscillator, Phasor, PwmOscillator,
                             Reverb, StereoMixer, Voice, sample_phase, sawtooth, sine, square,
                             triangle)
from klang_amd.audio.envelopes import Envelope
from klang_amd.block import Block
from klang_amd.errors import KlangError
from klang_amd.klang import run_klang
from klang_amd.messages import Note
from klang_amd.schedule import determine_execution_order
from tests.helpers import CHIRPS

B = 256
DT = 1. / 44100
WAVES = {'sine': sine, 'square': square, 'sawtooth': sawtooth, 'triangle': triangle}


class Source(Block):
    def __init__(self, value):
        super().__init__(nOutputs=1)
        self.output.set_value(value)


def test_sample_phase_and_waves_vs_golden(golden):
    g = golden('phase_waves')
    for i, f in enumerate(g['frequencies']):
        for j, s in enumerate(g['starts']):
            ph, nxt = sample_phase(f, s)
            np.testing.assert_array_equal(ph, g['phase'][i, j])
            assert nxt == g['next'][i, j]
            for name, fn in WAVES.items():
                np.testing.assert_array_equal(fn(ph), g[name][i, j])
    freq = golden('inputs')['frequency_arrays']
    cur = float(g['array_start'])
    for b in range(3):
        ph, cur = sample_phase(freq[b], cur)
        np.testing.assert_array_equal(ph, g['array_phase'][b])
    ph = Phasor(frequency=3.7, startPhase=.25)
    pw = PwmOscillator(frequency=441., dutyCycle=.3)
    ls = Lfo(frequency=2., shape=1, outputRange=(100., 1000.))
    lv = Lfo(frequency=5., wave_func=triangle, shape=B, outputRange=(-2., 3.))
    for b in range(4):
        for blk in (ph, pw, ls, lv):
            blk.update()
        assert ph.output.value == g['phasor_3p7_start_0p25'][b]
        np.testing.assert_array_equal(pw.output.value, g['pwm_441_0p3'][b])
        assert ls.output.value == g['lfo_scalar'][b]
        np.testing.assert_array_equal(lv.output.value, g['lfo_vector'][b])


def test_host_envelope_vs_golden(golden):
    g = golden('envelope')
    for case in ('adsr_regate', 'ar_zero_attack', 'd_only', 'retrigger_on', 'retrigger_off',
                 'loop_gate_off', 'overshoot_clip_low', 'overshoot_large', 'double_events'):
        a, d, s, r, over, retrig, loop = g[case + '_params']
        env = Envelope(a, d, s, r, DT, over, bool(retrig), bool(loop))
        script = {}
        for blk, gate in g[case + '_script']:
            script.setdefault(int(blk), []).append(bool(gate))
        out = []
        for b in range(int(g[case + '_n_blocks'])):
            for gate in script.get(b, ()):
                env.gate(gate)
            out.append(env.sample(B))
        np.testing.assert_array_equal(np.concatenate(out), g[case + '_samples'])
        assert env.active == bool(g[case + '_active'][-1])


def test_host_filter_vs_golden(golden):
    g = golden('filter')
    x = golden('inputs')['noise'][:8 * B]
    for c in (440., 2000., 5., 30000.):
        f = Filter(frequency=c)
        ys = []
        for b in range(8):
            f.input.set_value(x[b * B:(b + 1) * B])
            f.update()
            ys.append(f.output.value)
        assert np.abs(np.concatenate(ys) - g['noise_%g' % c]).max() <= 1e-11
    f = Filter(frequency=1000.)
    f.input.set_value(np.stack([x[:B], x[B:2 * B]]))
    f.update()
    assert np.abs(f.output.value - g['stereo_1000']).max() <= 1e-11


def test_host_ring_filters_and_reverb_vs_golden(golden):
    g = golden('ring_filters')
    x = golden('inputs')['noise']
    for name, cls in (('forward', ForwardCombFilter), ('backward', BackwardCombFilter),
                      ('echo', EchoFilter)):
        for K in (1, 7, 234, 1327):
            total = 1664 if K == 1327 else 640
            f = cls(K, .9 if K != 7 else .5)
            y = np.concatenate([f.filter(x[p:p + 123]) for p in range(0, total, 123)])
            np.testing.assert_array_equal(y[:total], g['%s_K%d_bs123' % (name, K)])
    with pytest.raises(ValueError):
        BackwardCombFilter(5).filter(np.zeros((2, 4)))
    with pytest.raises(RuntimeError):
        BackwardCombFilter(60 * 44100 + 1)
    r = golden('reverb')
    rv = Reverb(decay=3., dryWet=.7)
    assert [f.length for f in rv.filters] == list(r['taps_default'])
    out = []
    for b in range(24):
        rv.input.set_value(x[b * B:(b + 1) * B])
        rv.update()
        out.append(rv.output.value)
    np.testing.assert_allclose(np.concatenate(out), r['decay3_noise'], rtol=0, atol=1e-15)


def test_host_mixers_vs_golden(golden):
    g = golden('mixer')
    x = golden('inputs')['mixer_inputs']
    for n in (1, 2, 3, 64):
        m = Mixer()
        for c in range(n):
            m += Source(x[c])
            m.gains[-1] = g['gains'][c]
        m.update()
        np.testing.assert_array_equal(m.output.value, g['mono_%d' % n])
    sm = StereoMixer()
    for c in range(20):
        sm += Source(x[c])
        sm.gains[-1] = g['gains'][c]
        sm.pannings[-1] = g['stereo_20_pannings'][c]
    sm.update()
    np.testing.assert_allclose(sm.output.value, g['stereo_20'], rtol=0, atol=2e-16)
    with pytest.raises(ValueError):
        StereoMixer(1, mode='bogus').update()


def _polyphonic_patch(n=48):
    p = P.polyphonic_saw_patch(n_voices=n, n_blocks=40, off_base=6, off_span=12)
    mixer, voices = Mixer(), []
    for v in range(n):
        voice = Voice(Oscillator(wave_func=sawtooth),
                      ADSR(p.attack[v], p.decay[v], p.sustain[v], p.release[v]))
        fil = Filter(frequency=p.cutoff[v])
        voice | fil
        mixer += fil
        voices.append(voice)
    return p, mixer, voices


def test_host_chain_config3_mini_vs_golden(golden):
    g = golden('chains')
    p, mixer, voices = _polyphonic_patch()
    order = determine_execution_order([mixer])
    by_block = {}
    for ev in p.events:
        note = Note(0, float(ev['velocity']) if ev['kind'] == 1 else 0., float(ev['frequency']))
        by_block.setdefault(int(ev['block']), []).append((int(ev['voice']), note))
    out = []
    for b in range(12):
        for v, note in by_block.get(b, ()):
            voices[v].input.push(note)
        for blk in order:
            blk.update()
        out.append(mixer.output.value)
    assert np.abs(np.array(out) - g['config3_mini_mix'][:12]).max() <= 1e-12


def test_hello_world_config_on_host(tmp_path, golden):
    """BASELINE configs[0]: examples/hello_world.py, offline to WAV, CPU only."""
    path = str(tmp_path / 'hello.wav')
    osc = Oscillator(frequency=440.)
    frames = run_klang(osc | Dac(), backend='host', filepath=path, offline=True, duration=1.0)
    audio = np.concatenate(frames)
    assert audio.shape == (172 * B, 1) and audio.dtype == np.float32
    t = np.arange(172 * B) / 44100.
    assert np.abs(audio[:, 0] - np.sin(2. * np.pi * 440. * t)).max() <= 1e-6
    np.testing.assert_array_equal(audio[:4 * B, 0].reshape(4, B),
                                  golden('hello_world')['first_blocks_f32'])
    with wave.open(path, 'rb') as wav:
        assert (wav.getnframes(), wav.getnchannels(), wav.getframerate()) == (44032, 1, 44100)
        pcm = np.frombuffer(wav.readframes(-1), dtype='<i2')
    assert np.abs(pcm).max() == 32767                      # peak-normalised, wavfile.py:60
    with pytest.raises(KlangError):
        run_klang(Oscillator() | Dac(), backend='host', offline=True)     # duration INF
    with pytest.raises(ValueError):
        run_klang()


def test_fadeout_and_stereo_dump(tmp_path):
    path = str(tmp_path / 'st.wav')
    mixer = StereoMixer()
    for f, pan in ((220., -1.), (330., 1.)):
        mixer += Oscillator(frequency=f)
        mixer.pannings[-1] = pan
    frames = run_klang(mixer | Dac(nChannels=2), backend='host', filepath=path, offline=True,
                       duration=.1, fadeout=.05)
    audio = np.concatenate(frames)
    assert audio.shape[1] == 2 and np.abs(audio[-1]).max() < 5e-2 < np.abs(audio[:B]).max()
    assert os.path.getsize(path) == 44 + audio.size * 2


# ---- graph compiler: pattern matching (no device) -------------------------------------------------
def test_find_fusable_matches_voice_filter_mixer():
    p, mixer, voices = _polyphonic_patch(6)
    dac = mixer | Dac()
    order = determine_execution_order([dac])
    groups, errors = fusion.find_fusable(order)
    assert not errors and len(groups) == 1
    sink, chains, _ = groups[0]
    assert sink is mixer and len(chains) == 6
    assert chains[0].shape == (True, True, None, False, 0, (), False)
    assert [c.voice for c in chains] == voices and chains[0].wave_id() == 2


def test_find_fusable_single_chain_and_reverb_shape():
    osc = Oscillator(440.)
    dac = osc | Dac()
    groups, errors = fusion.find_fusable(determine_execution_order([dac]))
    assert not errors and [(s, len(c)) for s, c, _ in groups] == [(osc, 1)]
    voice = Voice(Oscillator(), AR(.01, 1.))
    rv = Reverb(decay=3., dryWet=.7)
    dac = voice | rv | Dac()
    groups, errors = fusion.find_fusable(determine_execution_order([dac]))
    assert not errors and groups[0][0] is rv
    assert groups[0][1][0].shape == (True, False, (3., .03, .7, 10), False, 0, (), False)


def test_find_fusable_reports_what_it_cannot_fuse():
    lfo = Lfo(frequency=2., shape=256, outputRange=(200., 2000.))
    osc, fil = Oscillator(wave_func=sawtooth), Filter()
    lfo | fil.frequency            # an audio-rate Lfo cannot drive the per-buffer cutoff
    dac = osc | fil | Dac()
    groups, errors = fusion.find_fusable(determine_execution_order([dac]))
    assert any('shape=1' in e for e in errors)
    shared = Oscillator()
    m = Mixer()
    m += shared
    shared.output.connect(Block(1, 0).input)      # second consumer: the oscillator's stream is
    groups, errors = fusion.find_fusable(determine_execution_order([m]))   # needed outside
    assert not errors and [g[0] for g in groups] == [m, shared]
    assert groups[0][1][0].external is shared.output                # the mixer takes it as input


# ---- effects: host blocks vs the reference's fixtures ------------------------------
def _run_effect(make, n_blocks, golden_key, golden, tol=0.):
    from klang_amd.audio import Oscillator, sawtooth
    from klang_amd.schedule import determine_execution_order
    osc = Oscillator(frequency=220., wave_func=sawtooth)
    blk = make()
    osc | blk
    order = determine_execution_order([blk])
    outs = []
    for _ in range(n_blocks):
        for o in order:
            o.update()
        outs.append(np.array(blk.output.value, dtype=float))
    want = golden('effects')[golden_key]
    if tol:
        np.testing.assert_allclose(np.array(outs), want, rtol=0, atol=tol)
    else:
        np.testing.assert_array_equal(np.array(outs), want)
    return blk


def test_effect_blocks_match_reference_fixtures(golden):
    from klang_amd.audio import (Delay, Gain, OctaveDistortion, TanhDistortion, Transformer,
                                 Tremolo)
    _run_effect(lambda: Gain(.37), 24, 'gain', golden)
    _run_effect(lambda: Transformer(.3, .1), 24, 'transformer', golden)
    _run_effect(lambda: TanhDistortion(2.), 24, 'tanh', golden)
    _run_effect(lambda: OctaveDistortion(), 24, 'octave', golden)
    # the one-pole coefficients come from the library's exp (<= 1 ulp from math.exp)
    trem = _run_effect(lambda: Tremolo(rate=3., depth=.8, smoothness=.5, dutyCycle=.3), 60,
                       'tremolo', golden, tol=1e-13)
    final = golden('effects')['tremolo_final']
    assert trem.lfo.currentPhase == final[0] and abs(trem.zi[0] - final[1]) <= 1e-13
    _run_effect(lambda: Tremolo(rate=7., depth=.5, smoothness=0., dutyCycle=.5), 30,
                'tremolo_square', golden, tol=1e-13)
    dl = _run_effect(lambda: Delay(time=.01, feedback=.6, drywet=.4), 40, 'delay', golden)
    assert [dl.ring.pos, dl.ring.length] == list(golden('effects')['delay_final_pos_length'])
    _run_effect(lambda: Delay(time=.0437, feedback=.9, drywet=.5), 40, 'delay_long', golden)
    with pytest.raises(ValueError):
        Delay(time=2.5)
    from klang_amd.audio import Bitcrusher, RingModulator, Subsampler, sine, square, triangle
    _run_effect(lambda: Subsampler(4), 6, 'subsample_4', golden)
    _run_effect(lambda: Subsampler(64), 6, 'subsample_64', golden)
    for bits in (5, 0, 13):
        _run_effect(lambda: Bitcrusher(bits), 12 if bits == 5 else 6, 'bitcrush_%d' % bits, golden)
    for wname, wave in (('square', square), ('sine', sine), ('triangle', triangle)):
        rm = _run_effect(lambda: RingModulator(frequency=500., rate=2., amount=20., wave_func=wave,
                                               dryWet=.8), 40, 'ringmod_' + wname, golden)
        final = golden('effects')['ringmod_%s_final' % wname]
        assert [rm.lfo.currentPhase, rm.modulator.currentPhase] == list(final)
    with pytest.raises(AssertionError):
        Subsampler(3)
    with pytest.raises(AssertionError):
        Bitcrusher(16)


def patched_parameter_patch():
    """The patch of tests/golden/effects_patched.npz 'patched_parameters' (tools/gen_golden.py):
    host Lfos at block rate into Tremolo.rate / depth and Gain.gain of three voice chains."""
    from klang_amd.audio import (ADSR, Filter, Gain, Lfo, Mixer, Oscillator, Tremolo, Voice,
                                 sawtooth, triangle)
    from klang_amd.messages import Note
    mixer = Mixer()
    for k in range(3):
        voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .04))
        voice.input.push(Note(0, .8, frequency=98. * (k + 1)))
        trem, gain = Tremolo(rate=4., depth=.5, smoothness=.4), Gain(.5)
        Lfo(frequency=.7 + k, outputRange=(2., 9.)) | trem.rate
        Lfo(frequency=1.3, wave_func=triangle, outputRange=(.2, .9)) | trem.depth
        Lfo(frequency=2. + k, outputRange=(.1, 1.)) | gain.gain
        mixer += voice | Filter(frequency=900.) | trem | gain
    return mixer


def test_block_rate_patched_effect_parameters_vs_reference_fixture(golden):
    from klang_amd.schedule import determine_execution_order
    want = golden('effects_patched')['patched_parameters']
    mixer = patched_parameter_patch()
    order = determine_execution_order([mixer])
    outs = []
    for _ in range(len(want)):
        for blk in order:
            blk.update()
        outs.append(np.array(mixer.output.value, dtype=float))
    # closed-form Butterworth row and one-pole design vs scipy's: a few 1e-14 relative
    np.testing.assert_allclose(np.array(outs), want, rtol=0, atol=1e-12)


@pytest.mark.parametrize('length', [100, 200, 255, 17])
def test_delay_lines_shorter_than_a_buffer_vs_reference_fixture(golden, length):
    """int(time * 44100) < 256: RingBuffer.extend's tail and head copies land inside the
    window Delay.update reads back (ring_buffer.py:88-109, effects.py:166-174); the host
    mirror reproduces the reference's result.  (The device path refuses such a Delay: see
    test_find_fusable_reports_what_it_cannot_fuse / DESIGN.md.)"""
    from klang_amd.audio import Delay, Oscillator, sawtooth
    from klang_amd.schedule import determine_execution_order
    want = golden('effects_patched')['delay_short_%d' % length]
    osc = Oscillator(frequency=220., wave_func=sawtooth)
    dl = Delay(time=(length + .5) / 44100., feedback=.6, drywet=.4)
    osc | dl
    assert dl.ring.length == length
    order = determine_execution_order([dl])
    outs = []
    for _ in range(len(want)):
        for o in order:
            o.update()
        outs.append(np.array(dl.output.value, dtype=float))
    np.testing.assert_array_equal(np.array(outs), want)


# ---- synthesizers (synthesizer.py:87-219) ------------------------------------------
def _notes(*pitches, velocity=1.):
    from klang_amd.messages import Note
    return [Note(pitch=p, velocity=velocity) for p in pitches]


def test_note_scheduler_known_answers():
    """The reference's own KATs (tests/test_audio_synthesizer.py:24-74), restated."""
    from klang_amd.audio import NoteScheduler
    C_ON, E_ON, G_ON = _notes(60, 64, 67)
    C_OFF, E_OFF, G_OFF = _notes(60, 64, 67, velocity=0.)
    rising = [C_ON, E_ON, G_ON, C_OFF, E_OFF, G_OFF]
    rise_fall = [C_ON, E_ON, G_ON, G_OFF, E_OFF, C_OFF]
    falling = [G_ON, E_ON, C_ON, G_OFF, E_OFF, C_OFF]
    fall_rise = [G_ON, E_ON, C_ON, C_OFF, E_OFF, G_OFF]
    expected = {
        'newest': ([C_ON, E_ON, G_ON, G_ON, G_ON, G_OFF], [C_ON, E_ON, G_ON, E_ON, C_ON, C_OFF],
                   [G_ON, E_ON, C_ON, C_ON, C_ON, C_OFF], [G_ON, E_ON, C_ON, E_ON, G_ON, G_OFF]),
        'oldest': ([C_ON, C_ON, C_ON, E_ON, G_ON, G_OFF], [C_ON, C_ON, C_ON, C_ON, C_ON, C_OFF],
                   [G_ON, G_ON, G_ON, E_ON, C_ON, C_OFF], [G_ON, G_ON, G_ON, G_ON, G_ON, G_OFF]),
        'lowest': ([C_ON, C_ON, C_ON, E_ON, G_ON, G_OFF], [C_ON, C_ON, C_ON, C_ON, C_ON, C_OFF],
                   [G_ON, E_ON, C_ON, C_ON, C_ON, C_OFF], [G_ON, E_ON, C_ON, E_ON, G_ON, G_OFF]),
        'highest': ([C_ON, E_ON, G_ON, G_ON, G_ON, G_OFF], [C_ON, E_ON, G_ON, E_ON, C_ON, C_OFF],
                    [G_ON, G_ON, G_ON, E_ON, C_ON, C_OFF], [G_ON, G_ON, G_ON, G_ON, G_ON, G_OFF]),
    }
    for policy, wants in expected.items():
        sched = NoteScheduler(policy=policy)
        for seq, want in zip((rising, rise_fall, falling, fall_rise), wants):
            assert [sched.get_next_note(n) for n in seq] == want, policy
    sched = NoteScheduler()
    assert sched.get_next_note(C_ON) == C_ON and sched.get_next_note(C_OFF) == C_OFF


def _play(synth, rows, n_blocks):
    from klang_amd.messages import Note
    by_block = {}
    for blk, pitch, velocity, frequency in rows:
        by_block.setdefault(int(blk), []).append(Note(int(pitch), velocity, frequency))
    out = []
    for b in range(n_blocks):
        for note in by_block.get(b, ()):
            synth.input.push(note)
        synth.update()
        out.append(np.array(synth.output.value, dtype=float))
    return np.array(out)


def _template(g):
    from klang_amd.audio import ADSR, Oscillator, Voice, sawtooth
    return Voice(Oscillator(wave_func=sawtooth), ADSR(*[float(x) for x in g['template']]))


def test_host_synthesizers_vs_golden(golden):
    """The host blocks (NumPy evaluation, backend='host') against the real reference."""
    from klang_amd.audio import MonophonicSynthesizer, PolyphonicSynthesizer
    from klang_amd.messages import Note
    g = golden('synthesizer')
    # Note's default frequency is the reference's equal temperament
    for blk, pitch, velocity, frequency in g['poly24_notes']:
        assert Note(int(pitch), velocity).frequency == frequency
    poly = PolyphonicSynthesizer(_template(g))
    got = _play(poly, g['poly24_notes'], len(g['poly24_out']))
    np.testing.assert_allclose(got, g['poly24_out'], rtol=0, atol=1e-15)
    assert [v.currentPitch for v in poly.voices] == list(g['poly24_state'][:, 1].astype(int))
    Poly3 = type('Poly3', (PolyphonicSynthesizer,), {'MAX_VOICES': 3})
    got = _play(Poly3(_template(g)), g['poly3_notes'], len(g['poly3_out']))
    np.testing.assert_allclose(got, g['poly3_out'], rtol=0, atol=1e-15)
    for policy in ('newest', 'oldest', 'lowest', 'highest'):
        mono = MonophonicSynthesizer(_template(g), policy=policy)
        handed = []
        push = mono.voice.input.push
        mono.voice.input.push = lambda note, _p=push: (handed.append(note), _p(note))[1]
        got = _play(mono, g['mono_notes'], len(g['mono_%s_out' % policy]))
        np.testing.assert_allclose(got, g['mono_%s_out' % policy], rtol=0, atol=1e-15)
        want = g['mono_%s_handed' % policy]
        assert [(n.pitch, n.velocity) for n in handed] == [(int(r[1]), r[2]) for r in want]


def test_find_fusable_synthesizers():
    from klang_amd import fusion
    from klang_amd.audio import (ADSR, Dac, Filter, Mixer, MonophonicSynthesizer, Oscillator,
                                 PolyphonicSynthesizer, Voice)
    from klang_amd.schedule import determine_execution_order
    make = lambda: Voice(Oscillator(), ADSR())
    poly = PolyphonicSynthesizer(make())
    groups, errors = fusion.find_fusable(determine_execution_order([poly | Dac()]))
    assert not errors and len(groups) == 1
    sink, chains, synths = groups[0]
    assert sink is poly and len(chains) == 24 and synths == [poly]
    # a mixer of synthesizers is ONE bank; kinds must not be mixed
    mixer = Mixer()
    a, b = PolyphonicSynthesizer(make()), PolyphonicSynthesizer(make())
    mixer += a
    mixer += b
    groups, errors = fusion.find_fusable(determine_execution_order([mixer | Dac()]))
    assert not errors and len(groups) == 1 and len(groups[0][1]) == 48 and groups[0][2] == [a, b]
    mixer = Mixer()
    mixer += PolyphonicSynthesizer(make())
    mixer += MonophonicSynthesizer(make())
    groups, errors = fusion.find_fusable(determine_execution_order([mixer | Dac()]))
    # different kinds: a bank each, and the mixer sums their outputs as external voices
    assert not errors and len(groups) == 3 and groups[0][0] is mixer
    assert all(c.external is not None for c in groups[0][1])
    # effects after a synthesizer work on the summed signal: a second bank that takes the
    # synthesizer's output as external input (examples/reverberation_demo.py:33-34 shape)
    syn, fil = PolyphonicSynthesizer(make()), Filter()
    groups, errors = fusion.find_fusable(determine_execution_order([syn | fil | Dac()]))
    assert not errors and [g[0] for g in groups] == [syn, fil]
    assert groups[1][1][0].external is syn.output and groups[1][1][0].filter is fil


def test_host_chirper_and_kick_vs_golden(golden):
    """Chirper (oscillators.py:280-308, chirp_phase with its held region) and Kick
    (synthesizer.py:253-283) as host blocks: the same NumPy operations as the reference."""
    from klang_amd.audio import Chirper, Kick, sawtooth, sine, square, triangle
    from klang_amd.messages import Note
    g = golden('chirper')
    waves = dict(sine=sine, sawtooth=sawtooth, square=square, triangle=triangle)
    for name, args, wname, _ in CHIRPS:
        wave = waves[wname]
        ch = Chirper(*args, wave_func=wave)
        out = []
        for _ in range(len(g[name])):
            ch.update()
            out.append(np.array(ch.output.value))
        np.testing.assert_array_equal(np.array(out), g[name])
        np.testing.assert_array_equal(ch.t, g[name + '_t'])
        ch.reset()
        assert ch.t[0] == 0. and ch.t[1] == 1. / 44100
    with pytest.raises(ValueError):
        Chirper(100., 1., -100., 'logarithmic').update()
    with pytest.raises(ValueError):
        Chirper(100., 1., 200., 'cubic').update()
    Chirper(100., 0., -100., 'logarithmic').update()   # never reaches the sweep: no error
    Chirper(0., 1., 0., 'hyperbolic').update()
    s = golden('synthesizer')
    kick = Kick()
    hits = {0: Note(36), 30: Note(36, 0.), 70: Note(36)}
    out = []
    for b in range(100):
        if b in hits:
            kick.input.push(hits[b])
        kick.update()
        out.append(np.array(kick.output.value))
    np.testing.assert_array_equal(np.array(out), s['kick_default_out'])
    np.testing.assert_array_equal(np.r_[kick.currentTime, kick.currentPhase], s['kick_default_final'])


def test_host_hihat_vs_golden_with_the_same_seed(golden):
    """HiHat (synthesizer.py:222-250) is a D envelope times white noise from NumPy's global
    RandomState: with the seed the fixture was made with, the host block is bit-exact.  (Under
    backend='hip' its envelope runs on the device and the noise goes in as the voice's
    external input: tests/test_gpu_synth.py.)"""
    from klang_amd.audio import HiHat
    s = golden('synthesizer')
    state = np.random.get_state()
    try:
        for name, seed, kw in (('hihat_fresh', 1234, {}),
                               ('hihat_looped', 99, dict(decay=.12, loopedNoise=True))):
            np.random.seed(seed)
            hat = HiHat(**kw)
            out = []
            for b in range(60):
                if b in (2, 30):
                    hat.input.push(Note(42))
                if b == 35:
                    hat.input.push(Note(42, 0.))
                hat.update()
                out.append(np.array(hat.output.value))
            np.testing.assert_array_equal(np.array(out), s[name + '_out'])
    finally:
        np.random.set_state(state)
    assert HiHat in fusion.HOT_PATH_TYPES


def _stereo_delay_patch(stereo):
    from klang_amd.audio import AudioCombiner, StereoDelay
    left, right = Oscillator(frequency=220., wave_func=sawtooth), Oscillator(311.)
    sd = StereoDelay(leftTime=.05, rightTime=.08, leftFeedback=.5, rightFeedback=.3, drywet=.4)
    if stereo:
        comb = AudioCombiner(nInputs=2)
        left | comb.inputs[0]
        right | comb.inputs[1]
        comb | sd
    else:
        left | sd
    return sd


@pytest.mark.parametrize('stereo,key', [(True, 'stereo_delay'), (False, 'stereo_delay_mono_in')])
def test_host_stereo_delay_vs_golden(golden, stereo, key):
    """StereoDelay = AudioSplitter -> two Delays -> AudioCombiner (effects.py:177-236): the
    host composite bit for bit, on a stereo signal and on a mono one."""
    sd = _stereo_delay_patch(stereo)
    assert [type(b).__name__ for b in sd.execOrder] == ['AudioSplitter', 'Delay', 'Delay',
                                                        'AudioCombiner']
    order = determine_execution_order([sd])
    want = golden('effects')[key]
    for b in range(len(want)):
        for blk in order:
            blk.update()
        np.testing.assert_array_equal(sd.output.value, want[b])


def test_fuser_plans_for_stereo_signals():
    """A StereoMixer that takes its inputs as external voices gets two voices per input (a
    stereo input needs both); externally fed chains under a StereoMixer stay banks of their
    own, because their source may turn out to be stereo; a StereoDelay is opened."""
    from klang_amd.audio import AudioCombiner, Bitcrusher, StereoDelay, Tremolo
    comb = AudioCombiner(nInputs=2)
    Oscillator(220., wave_func=sawtooth) | comb.inputs[0]
    Oscillator(331.) | comb.inputs[1]
    mixer = StereoMixer()
    mixer += Oscillator(110., wave_func=sawtooth) | StereoDelay(leftTime=.03, rightTime=.05)
    mixer += Oscillator(440.) | Bitcrusher(6)
    mixer += comb | Tremolo(rate=3.)
    order = fusion.open_composites(determine_execution_order([mixer | Dac(nChannels=2)]))
    assert 'StereoDelay' not in [type(b).__name__ for b in order]
    groups, errors = fusion.find_fusable(order)
    assert not errors
    by_sink = {type(sink).__name__: chains for sink, chains, _ in groups}
    assert len(by_sink['StereoMixer']) == 6 and all(c.end is None for c in by_sink['StereoMixer'])
    owners = [inp.incomingConnection.owner for inp in mixer.inputs]
    assert [c.external.owner for c in by_sink['StereoMixer'][::2]] == owners
    assert by_sink['Tremolo'][0].external.owner is comb
    assert len([s for s, _, _ in groups if type(s).__name__ == 'Delay']) == 2
