"""The voice kernels are different schedules of the same arithmetic: for the banks that more
than one of them can run, their float64 voice streams and the state they leave must be equal
bit for bit.  Kernels are chosen with the library's environment knobs, which
it reads on every launch.

  plain chains:   voice_kernel (one lane per voice)  vs  voice_pipe_kernel with 1..64 voices per
                  workgroup (forced), equal and two-size grids
  FmOscillator:   voice_kernel<FM>                    vs  voice_mod_kernel
(Stage programs -- voice_fx_kernel vs the stage interpreter -- are compared in
tests/test_gpu_effects.py::test_compiled_effect_programs_vs_oracle_bit_exact.)
"""
import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, hip_bank_from_patch

pytestmark = pytest.mark.gpu

STATE = ('F_PHASE', 'F_ENV_VALUE', 'F_ENV_STAGE', 'F_ENV_ENABLED', 'F_FREQUENCY', 'F_AMPLITUDE')


@pytest.fixture()
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def random_patch(rng, n, nb, has_filter, stereo):
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=2, off_span=max(1, nb // 2))
    patch['has_filter'], patch['stereo'] = has_filter, stereo
    patch['wave'] = rng.integers(0, 4, n).astype(np.int32)
    if rng.random() < .5:      # wavefronts that agree on their wave function
        patch['wave'] = np.repeat(rng.integers(0, 4, -(-n // 64)), 64)[:n].astype(np.int32)
    patch['frequency'] = 20. + 4000. * rng.random(n)
    patch['gain'] = rng.random(n)
    patch['pan'] = 2. * rng.random(n) - 1.
    patch['loop'] = rng.random(n) < .3
    patch['release'] = .001 + .05 * rng.random(n)
    return patch


def outcome(hip, ctx, patch, nb, split, **bank_kw):
    bank = hip_bank_from_patch(hip, ctx, patch, **bank_kw)
    voices = bank.render_voices(nb, f64=True)
    mix = np.concatenate([bank.render(split), bank.render(nb - split)])
    state = [bank.get_state(getattr(hip, f)) for f in STATE]
    if patch.has_filter:
        state += [bank.get_state(hip.F_Z0), bank.get_state(hip.F_Z1)]
    bank.close()
    return voices, mix, state


def assert_same(a, b):
    np.testing.assert_array_equal(a[0], b[0])                 # float64 voice streams
    assert np.abs(a[1] - b[1]).max() <= 2e-7                  # float32 bus: fold order differs
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize('seed', range(6))
def test_plain_chains_lane_per_voice_vs_pipelined(hip, ctx, monkeypatch, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([1, 20, 63, 200, 1000, 3000, 5000]))
    nb = int(rng.integers(3, 9))
    patch = random_patch(rng, n, nb, has_filter=bool(seed & 1), stereo=bool(seed & 2))
    split = int(rng.integers(1, nb))
    monkeypatch.setenv('KLMI_PIPE_VOICES_LOG2', '-1')
    want = outcome(hip, ctx, patch, nb, split)
    monkeypatch.setenv('KLMI_VOICE_WAVES', '2')    # the build for two waves per SIMD (large banks)
    assert_same(outcome(hip, ctx, patch, nb, split), want)
    monkeypatch.delenv('KLMI_VOICE_WAVES')
    for log2 in ('0', '2', '5', '6'):
        monkeypatch.setenv('KLMI_PIPE_VOICES_LOG2', log2)
        assert_same(outcome(hip, ctx, patch, nb, split), want)
    monkeypatch.delenv('KLMI_PIPE_VOICES_LOG2')
    assert_same(outcome(hip, ctx, patch, nb, split), want)      # the library's own choice
    monkeypatch.setenv('KLMI_PIPE_SPLIT', '0')                  # ... with equal workgroups
    assert_same(outcome(hip, ctx, patch, nb, split), want)


@pytest.mark.parametrize('target', ['MOD_FM', 'MOD_FREQUENCY', 'MOD_DUTY'])
@pytest.mark.parametrize('seed', range(4))
def test_modulated_fast_forms_vs_general_kernel(hip, ctx, monkeypatch, seed, target):
    """FmOscillator, Lfo(256) -> frequency and Lfo(256) -> duty cycle: the fast kernel's forms
    against voice_mod_kernel (KLMI_FM_KERNEL=0), wavefronts that agree on the Lfo's wave
    function and wavefronts that do not, a few negative Lfo frequencies."""
    rng = np.random.default_rng(200 + seed)
    n, nb = int(rng.choice([5, 130, 700])), 6
    patch = random_patch(rng, n, nb, has_filter=bool(seed & 1), stereo=bool(seed & 2))
    tgt = getattr(hip, target)
    if tgt == hip.MOD_DUTY:
        patch['wave'] = np.full(n, P.WAVE_PWM, dtype=np.int32)
    mod_wave = np.asarray(patch['wave']) if (tgt == hip.MOD_FM and rng.random() < .5) else rng.integers(0, 4, n)
    if tgt != hip.MOD_FM and seed & 1:
        mod_wave = np.repeat(rng.integers(0, 4, -(-n // 64)), 64)[:n]
    mod_freq = 1. + 300. * rng.random(n)
    mod_freq[::37] = -mod_freq[::37]
    scale, offset = {hip.MOD_FM: (8. * rng.random(n) - 4., np.zeros(n)),
                     hip.MOD_FREQUENCY: (40. * rng.random(n), np.asarray(patch['frequency'], dtype=float)),
                     hip.MOD_DUTY: (.4 * rng.random(n), np.full(n, .5))}[tgt]
    fields = [(hip.F_MOD_FREQUENCY, mod_freq), (hip.F_MOD_SCALE, scale), (hip.F_MOD_OFFSET, offset),
              (hip.F_MOD_WAVE, mod_wave)]

    def run():
        bank = hip_bank_from_patch(hip, ctx, patch, mod_target=tgt)
        for field, values in fields:
            bank.set(field, values)
        voices = bank.render_voices(nb, f64=True)
        mix = np.concatenate([bank.render(2), bank.render(nb - 2)])
        state = [bank.get_state(getattr(hip, f)) for f in STATE + ('F_MOD_PHASE',)]
        bank.close()
        return voices, mix, state
    fast = run()
    monkeypatch.setenv('KLMI_FM_KERNEL', '0')
    general = run()
    sines = (np.asarray(patch['wave']) == 0).any() or (np.asarray(mod_wave) == 0).any()
    if tgt == hip.MOD_FM and sines:
        # sine carriers: the exact form calls sin() outside [0, TAU], the speculative one the
        # wrapped polynomial -- both within an ulp of the true value
        assert np.abs(fast[0] - general[0]).max() <= 1e-13
        for x, y in zip(fast[2], general[2]):
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-12)
    else:
        assert_same(fast, general)


@pytest.mark.parametrize('poly,stereo', [(24, False), (5, True), (1, False)])
def test_gated_fm_voices_fast_form_vs_general_kernel(hip, ctx, monkeypatch, poly, stereo):
    """Synthesizer banks whose voices are FmOscillators (non-sine, so every operation is an IEEE
    basic one): the gated FM form of the fast kernel against voice_mod_kernel (KLMI_FM_KERNEL=0)
    -- the mixed bus and every piece of state a render leaves, idle voices' carrier and
    modulator phases among them."""
    rng = np.random.default_rng(300 + poly)
    n_inst, nb = 40, 16
    n = poly * n_inst
    wave = np.repeat(rng.integers(1, 4, -(-n // 64)), 64)[:n]
    if n > 128:
        wave[64:128] = rng.integers(1, 4, 64)     # one wavefront that does not agree (exact form)
    rows = []
    for inst in range(n_inst):
        for _ in range(int(rng.integers(0, 2 * poly + 3))):
            blk, pitch = int(rng.integers(0, nb - 1)), int(rng.integers(36, 96))
            f = 440. * 2 ** ((pitch - 69) / 12)
            rows.append((inst, blk, pitch, 0, .2 + .8 * rng.random(), f))
            if rng.random() < .8:
                rows.append((inst, min(nb - 1, blk + int(rng.integers(0, 5))), pitch, 0, 0., f))
    notes = np.array(rows, dtype=hip.NOTE_DTYPE)
    fields = [(hip.F_WAVE, wave), (hip.F_ATTACK, .001 + .02 * rng.random(n)),
              (hip.F_DECAY, .005 + .05 * rng.random(n)), (hip.F_SUSTAIN, .2 + .7 * rng.random(n)),
              (hip.F_RELEASE, .005 + .05 * rng.random(n)), (hip.F_MOD_FREQUENCY, 20. + 200. * rng.random(n)),
              (hip.F_MOD_SCALE, 6. * rng.random(n) - 3.), (hip.F_MOD_WAVE, wave),
              (hip.F_GAIN, .5 + rng.random(n)), (hip.F_PAN, 2. * rng.random(n) - 1.)]

    def run():
        bank = hip.Bank(ctx, n, has_env=True, poly_voices=poly, mod_target=hip.MOD_FM, stereo=stereo)
        for field, values in fields:
            bank.set(field, values)
        bank.notes(notes)
        mix = np.concatenate([bank.render(7), bank.render(nb - 7)])
        state = [bank.get_state(getattr(hip, f)) for f in STATE + ('F_MOD_PHASE', 'F_CURRENT_PITCH',
                                                                 'F_ROUND_ROBIN')]
        bank.close()
        return mix, state
    fast = run()
    monkeypatch.setenv('KLMI_FM_KERNEL', '0')
    general = run()
    assert np.abs(fast[0]).max() > 1e-3 and np.abs(fast[0] - general[0]).max() <= 2e-7
    for x, y in zip(fast[1], general[1]):
        np.testing.assert_array_equal(x, y)
