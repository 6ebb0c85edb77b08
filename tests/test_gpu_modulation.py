"""HIP path vs the reference's fixtures and vs the oracle for the modulation inputs
(SURVEY 8f rank 1): Lfo -> Filter.frequency, FmOscillator, Lfo(256) -> frequency,
Lfo(256) -> PwmOscillator.dutyCycle.  Everything goes through the C-ABI."""
import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, hip_bank_from_patch, library_table, oracle_bank_from_patch, report

pytestmark = pytest.mark.gpu
WAVES = ('sine', 'square', 'sawtooth', 'triangle')


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def one_voice(hip, ctx, wave, frequency=440., phase=0., **kw):
    bank = hip.Bank(ctx, 1, has_env=False, **kw)
    bank.set(hip.F_WAVE, [wave])
    bank.set(hip.F_FREQUENCY, [frequency])
    bank.set_state(hip.F_PHASE, [phase])
    return bank


@pytest.mark.parametrize('wname', WAVES)
def test_lfo_cutoff_vs_reference(hip, ctx, golden, wname):
    g = golden('modulation')
    scale, offset = g['lfo_cutoff_scale_offset']
    bank = one_voice(hip, ctx, hip.WAVE_SAWTOOTH, 220., has_filter=True, cutoff_lfo=True)
    bank.set(hip.F_LFO_FREQUENCY, [3.7])
    bank.set(hip.F_LFO_WAVE, [WAVES.index(wname)])
    bank.set(hip.F_LFO_SCALE, [scale])
    bank.set(hip.F_LFO_OFFSET, [offset])
    bank.set_state(hip.F_LFO_PHASE, [.3])
    want = g['lfo_cutoff_%s_out' % wname]
    got = bank.render_voices(40, f64=True)[0].reshape(40, B)
    # the library's closed-form Butterworth table vs scipy's: <= 4e-15 relative per row
    assert np.abs(got - want).max() <= 1e-10
    mix = bank.render(40)[:, 0, :]
    assert report('modulation: Lfo -> Filter.frequency', np.abs(mix - want).max()) <= 1e-5
    assert abs(bank.get_state(hip.F_LFO_PHASE)[0] - g['lfo_cutoff_%s_final_phase' % wname]) <= 1e-12


@pytest.mark.parametrize('wname', WAVES)
def test_fm_oscillator_vs_reference(hip, ctx, golden, wname):
    g = golden('modulation')
    w = WAVES.index(wname)
    bank = one_voice(hip, ctx, w, 440., mod_target=hip.MOD_FM)
    bank.set(hip.F_MOD_FREQUENCY, [110.])
    bank.set(hip.F_MOD_WAVE, [w])
    bank.set(hip.F_MOD_SCALE, [2.5])
    want = g['fm_%s_out' % wname]
    got = bank.render_voices(12, f64=True)[0].reshape(12, B)
    if wname == 'sine':
        assert np.abs(got - want).max() <= 1e-14
    else:
        np.testing.assert_array_equal(got, want)
    assert report('modulation: FmOscillator', np.abs(bank.render(12)[:, 0, :] - want).max()) <= 1e-5
    np.testing.assert_array_equal(
        [bank.get_state(hip.F_PHASE)[0], bank.get_state(hip.F_MOD_PHASE)[0]],
        g['fm_%s_final_phases' % wname])


@pytest.mark.parametrize('wname', WAVES)
def test_frequency_modulated_by_audio_rate_lfo_vs_reference(hip, ctx, golden, wname):
    g = golden('modulation')
    scale, offset = g['vibrato_scale_offset']
    bank = one_voice(hip, ctx, hip.WAVE_SAWTOOTH, 440., phase=1., mod_target=hip.MOD_FREQUENCY)
    bank.set(hip.F_MOD_FREQUENCY, [5.])
    bank.set(hip.F_MOD_WAVE, [WAVES.index(wname)])
    bank.set(hip.F_MOD_SCALE, [scale])
    bank.set(hip.F_MOD_OFFSET, [offset])
    want = g['vibrato_%s_out' % wname]
    got = bank.render_voices(12, f64=True)[0].reshape(12, B)
    if wname == 'sine':   # sin ulps enter the running frequency sum
        assert np.abs(got - want).max() <= 1e-10
    else:
        np.testing.assert_array_equal(got, want)
    assert report('modulation: Lfo(256) -> frequency', np.abs(bank.render(12)[:, 0, :] - want).max()) <= 1e-5
    if wname != 'sine':
        np.testing.assert_array_equal(
            [bank.get_state(hip.F_PHASE)[0], bank.get_state(hip.F_MOD_PHASE)[0]],
            g['vibrato_%s_final_phases' % wname])


@pytest.mark.parametrize('wname', WAVES)
def test_duty_cycle_modulated_by_audio_rate_lfo_vs_reference(hip, ctx, golden, wname):
    g = golden('modulation')
    scale, offset = g['pwm_scale_offset']
    bank = one_voice(hip, ctx, hip.WAVE_PWM, 220., mod_target=hip.MOD_DUTY)
    bank.set(hip.F_MOD_FREQUENCY, [2.])
    bank.set(hip.F_MOD_WAVE, [WAVES.index(wname)])
    bank.set(hip.F_MOD_SCALE, [scale])
    bank.set(hip.F_MOD_OFFSET, [offset])
    want = g['pwm_%s_out' % wname]
    got = bank.render_voices(12, f64=True)[0].reshape(12, B)
    if wname == 'sine':   # +-1 output: an ulp of the duty moves an edge by at most a sample
        assert np.mean(got != want) <= 1e-3
    else:
        np.testing.assert_array_equal(got, want)


def _modulated_banks(hip, ctx, oracle, n, nb, target, cutoff_lfo, seed, sine=False, paired=False,
                     has_filter=True):
    """A config-3 style patch (saw -> ADSR -> biquad -> Mixer, notes on/off) with random
    modulators; returns (hip bank, oracle bank, patch).  paired (FM): carrier and modulator of
    a voice share their wave function, as FmOscillator builds them (oscillators.py:211-213),
    64 consecutive voices share it too, and intensities take either sign."""
    rng = np.random.default_rng(seed)
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=max(1, nb // 2))
    patch['has_filter'] = has_filter
    waves = rng.integers(0 if sine else 1, 4, n)
    if target == hip.MOD_DUTY:
        patch['wave'] = np.full(n, P.WAVE_PWM)
    elif target == hip.MOD_FM and paired:
        patch['wave'] = np.repeat(np.arange(-(-n // 64)) % 3 + 1, 64)[:n].astype(np.int32)
    elif target == hip.MOD_FM:
        patch['wave'] = rng.integers(0 if sine else 1, 4, n)
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    hb = hip_bank_from_patch(hip, ctx, patch, cutoff_lfo=cutoff_lfo, mod_target=target)
    if cutoff_lfo:
        lf, ls, lo = .5 + 20. * rng.random(n), 200. + 1000. * rng.random(n), 1500. + 1000. * rng.random(n)
        lp = 6. * rng.random(n)
        ob.cutoff_lfo, ob.grid = True, grid
        ob.lfo_frequency, ob.lfo_scale, ob.lfo_offset, ob.lfo_phase = lf, ls, lo, lp.copy()
        ob.lfo_wave = waves.astype(np.int32)
        hb.set(hip.F_LFO_FREQUENCY, lf), hb.set(hip.F_LFO_SCALE, ls), hb.set(hip.F_LFO_OFFSET, lo)
        hb.set(hip.F_LFO_WAVE, waves), hb.set_state(hip.F_LFO_PHASE, lp)
    if target != hip.MOD_NONE:
        mf = 1. + 200. * rng.random(n)
        if target == hip.MOD_FM:
            ms, mo = 3. * rng.random(n), np.zeros(n)
        elif target == hip.MOD_FREQUENCY:
            ms, mo = 30. * rng.random(n), patch.frequency.copy()
        else:
            ms, mo = .4 * rng.random(n), np.full(n, .5)
        mp = 6. * rng.random(n)
        mw = rng.integers(0 if sine else 1, 4, n)
        if paired:
            mw, ms = np.asarray(patch['wave']).copy(), 8. * rng.random(n) - 4.
        ob.mod_target = target
        ob.mod_frequency, ob.mod_scale, ob.mod_offset, ob.mod_phase = mf, ms, mo, mp.copy()
        ob.mod_wave = mw.astype(np.int32)
        hb.set(hip.F_MOD_FREQUENCY, mf), hb.set(hip.F_MOD_SCALE, ms), hb.set(hip.F_MOD_OFFSET, mo)
        hb.set(hip.F_MOD_WAVE, mw), hb.set_state(hip.F_MOD_PHASE, mp)
    return hb, ob, patch


@pytest.mark.parametrize('target,cutoff_lfo', [(0, True), (1, False), (2, False), (3, False),
                                               (1, True), (2, True)])
def test_modulated_chains_vs_oracle_bit_exact(hip, ctx, oracle, target, cutoff_lfo):
    """Non-sine carriers and modulators: every operation is an IEEE basic operation, so
    the float64 voice streams and all state equal the oracle's bit for bit; launches are
    split (5 + 9 blocks) to carry modulator state across calls."""
    n, nb = 700, 14
    hb, ob, patch = _modulated_banks(hip, ctx, oracle, n, nb, target, cutoff_lfo, seed=11 + target)
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    got_voices = hb.render_voices(nb, f64=True)
    np.testing.assert_array_equal(got_voices, want_voices)
    mix = np.concatenate([hb.render(5), hb.render(9)])
    assert np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-6
    np.testing.assert_array_equal(hb.get_state(hip.F_PHASE), ob.phase)
    np.testing.assert_array_equal(hb.get_state(hip.F_Z0), ob.z0)
    if cutoff_lfo:
        np.testing.assert_array_equal(hb.get_state(hip.F_LFO_PHASE), ob.lfo_phase)
    if target:
        np.testing.assert_array_equal(hb.get_state(hip.F_MOD_PHASE), ob.mod_phase)


@pytest.mark.parametrize('has_filter,kernel', [(True, 'fast'), (False, 'fast'), (True, 'general')])
def test_fm_voices_in_the_fast_kernel_vs_oracle_bit_exact(hip, ctx, oracle, monkeypatch, has_filter,
                                                          kernel):
    """FmOscillator voices whose wavefronts agree on the wave function run the fast kernel's
    speculative FM region (three remainders under one check); voice streams, phases and filter
    state equal the oracle's bit for bit, over split launches.  The last 64 voices have a
    negative modulator frequency: their wavefront takes the exact per-sample form.  'general':
    the same bank through voice_mod_kernel's FM branch (which gated, polyphonic and effect
    chains with an FmOscillator still take)."""
    if kernel == 'general':
        monkeypatch.setenv('KLMI_FM_KERNEL', '0')
    n, nb = 64 * 11, 14
    hb, ob, patch = _modulated_banks(hip, ctx, oracle, n, nb, hip.MOD_FM, False, seed=23, paired=True,
                                     has_filter=has_filter)
    mf = ob.mod_frequency.copy()
    mf[-64:] = -mf[-64:]
    ob.mod_frequency = mf
    hb.set(hip.F_MOD_FREQUENCY, mf)
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    np.testing.assert_array_equal(hb.render_voices(nb, f64=True), want_voices)
    mix = np.concatenate([hb.render(6), hb.render(8)])
    assert np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-6
    np.testing.assert_array_equal(hb.get_state(hip.F_PHASE), ob.phase)
    np.testing.assert_array_equal(hb.get_state(hip.F_MOD_PHASE), ob.mod_phase)
    if has_filter:
        np.testing.assert_array_equal(hb.get_state(hip.F_Z0), ob.z0)


def test_modulated_chains_with_sines_within_tolerance(hip, ctx, oracle):
    """Sine carriers / modulators / Lfos: libm vs polynomial sine differ by ulps; FM and the
    cutoff Lfo stay within the float32 bar (a grid row may flip only on an exact edge)."""
    hb, ob, patch = _modulated_banks(hip, ctx, oracle, 400, 10, hip.MOD_FM, True, seed=5, sine=True)
    want_mix, _ = ob.render(10, patch.events)
    assert np.abs(hb.render(10)[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-5


def test_cutoff_lfo_in_the_fast_kernel_at_scale(hip, ctx, oracle):
    """65,536 voices with per-voice cutoff Lfos run in the speculative kernel (no
    audio-rate modulator); mix vs the threaded oracle."""
    hb, ob, patch = _modulated_banks(hip, ctx, oracle, 65536, 4, hip.MOD_NONE, True, seed=3)
    want, _ = ob.render(4, patch.events, n_threads=min(8, oracle.max_threads()))
    assert np.abs(hb.render(4)[:, 0, :] - want[:, 0, :]).max() <= 1e-6
    np.testing.assert_array_equal(hb.get_state(hip.F_LFO_PHASE), ob.lfo_phase)


def test_modulation_argument_errors(hip, ctx):
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 4, has_filter=False, cutoff_lfo=True)
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 4, mod_target=7)
    plain = hip.Bank(ctx, 4, has_filter=True)
    with pytest.raises(hip.NativeError) as e:
        plain.set(hip.F_LFO_FREQUENCY, [1.])
    assert e.value.code == hip.E_STATE
    with pytest.raises(hip.NativeError):
        plain.set(hip.F_MOD_WAVE, [0.])
    fm = hip.Bank(ctx, 4, mod_target=hip.MOD_FM)
    with pytest.raises(hip.NativeError) as e:
        fm.set(hip.F_MOD_WAVE, [4.])     # PWM is not a modulator wave
    assert e.value.code == hip.E_ARG
