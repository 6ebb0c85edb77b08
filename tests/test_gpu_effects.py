"""HIP path vs the reference's fixtures and vs the oracle for the effect stages of a voice
chain: Gain, Transformer, TanhDistortion, OctaveDistortion, Tremolo,
Delay, alone and chained around the Filter.  Everything goes through the C-ABI."""
import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, library_table, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def saw_bank(hip, ctx, stages, n=1, **kw):
    """stages: [(kind, p0, p1, ...)] -> bank of saw oscillators at 220 Hz."""
    bank = hip.Bank(ctx, n, has_env=kw.pop('has_env', False), effects=[s[0] for s in stages], **kw)
    bank.set(hip.F_WAVE, [hip.WAVE_SAWTOOTH] * n)
    bank.set(hip.F_FREQUENCY, [220.] * n)
    for s, st in enumerate(stages):
        for k, p in enumerate(st[1:]):
            bank.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    return bank


@pytest.mark.parametrize('key,stage,n_blocks,tol', [
    ('gain', ('FX_GAIN', .37), 24, 0.),
    ('transformer', ('FX_TRANSFORM', .3, .1), 24, 0.),
    ('tanh', ('FX_TANH', 2.), 24, 1e-15),             # device tanh vs NumPy's: a few ulp
    ('octave', ('FX_OCTAVE',), 24, 0.),
    ('tremolo', ('FX_TREMOLO', 3., .8, .5, .3), 60, 0.),
    ('tremolo_square', ('FX_TREMOLO', 7., .5, 0., .5), 30, 0.),
    ('delay', ('FX_DELAY', .01, .6, .4), 40, 0.),
    ('delay_long', ('FX_DELAY', .0437, .9, .5), 40, 0.),
    ('subsample_4', ('FX_SUBSAMPLE', 4), 6, 0.),
    ('subsample_64', ('FX_SUBSAMPLE', 64), 6, 0.),
    ('bitcrush_5', ('FX_BITCRUSH', 5), 12, 0.),
    ('bitcrush_0', ('FX_BITCRUSH', 0), 6, 0.),
    ('bitcrush_13', ('FX_BITCRUSH', 13), 6, 0.),
    ('ringmod_square', ('FX_RINGMOD', 500., 2., 20., .8, 1), 40, 1e-13),   # sine AM: libm ulps
    ('ringmod_sine', ('FX_RINGMOD', 500., 2., 20., .8, 0), 40, 1e-10),     # ... inside the cumsum
    ('ringmod_triangle', ('FX_RINGMOD', 500., 2., 20., .8, 3), 40, 1e-13),
])
def test_effect_stage_vs_reference(hip, ctx, golden, key, stage, n_blocks, tol):
    want = golden('effects')[key]
    bank = saw_bank(hip, ctx, [(getattr(hip, stage[0]),) + stage[1:]])
    got = bank.render_voices(n_blocks, f64=True)[0].reshape(n_blocks, B)   # preview
    if tol:
        assert np.abs(got - want).max() <= tol
    else:
        np.testing.assert_array_equal(got, want)
    # the committing render in two launches carries the stage state and the delay line
    k = n_blocks // 3
    mix = np.concatenate([bank.render(k), bank.render(n_blocks - k)])[:, 0, :]
    assert report('effect: ' + key.split('_')[0], np.abs(mix - want).max()) <= 1e-5 * max(1., np.abs(want).max())
    if key == 'tremolo':
        final = golden('effects')['tremolo_final']
        assert bank.get_state(hip.F_FX_STATE(0, 0))[0] == final[0]
        assert bank.get_state(hip.F_FX_STATE(0, 1))[0] == final[1]
    if key == 'delay':
        assert bank.get_state(hip.F_FX_STATE(0, 0))[0] == golden('effects')['delay_final_pos_length'][0]
    if key.startswith('ringmod'):
        final = golden('effects')[key + '_final']
        assert abs(bank.get_state(hip.F_FX_STATE(0, 0))[0] - final[0]) <= 1e-12
        assert abs(bank.get_state(hip.F_FX_STATE(0, 1))[0] - final[1]) <= 1e-9


def test_effect_chain_after_a_voice_vs_reference(hip, ctx, golden):
    """voice -> Filter -> TanhDistortion -> Tremolo -> Delay -> Gain with notes."""
    want = golden('effects')['chain']
    bank = saw_bank(hip, ctx, [(hip.FX_FILTER,), (hip.FX_TANH, 1.5), (hip.FX_TREMOLO, 5., .6, .4, .4),
                               (hip.FX_DELAY, .02, .5, .3), (hip.FX_GAIN, .8)],
                    has_env=True, has_filter=True)
    for f, val in ((hip.F_ATTACK, .01), (hip.F_DECAY, .05), (hip.F_SUSTAIN, .6),
                   (hip.F_RELEASE, .04), (hip.F_CUTOFF, 1500.)):
        bank.set(f, [val])
    bank.event(0, 0, hip.EV_NOTE_ON, .8, 220.)
    bank.event(0, 10, hip.EV_NOTE_OFF)
    bank.event(0, 14, hip.EV_NOTE_ON, .5, 329.6275569128699)
    got = bank.render_voices(30, f64=True)[0].reshape(30, B)
    assert np.abs(got - want).max() <= 1e-12     # closed-form vs scipy Butterworth row, tanh ulps
    mix = np.concatenate([bank.render(7), bank.render(23)])[:, 0, :]
    assert report('effect chain after a voice', np.abs(mix - want).max()) <= 1e-5


def test_effect_chains_vs_oracle_many_voices(hip, ctx, oracle):
    """600 voices (config-3 style notes) through Filter -> Octave -> Tremolo -> Delay ->
    Transformer -> Gain with per-voice parameters: float64 streams equal the oracle's bit
    for bit (no transcendental on the device side of this chain)."""
    n, nb = 600, 12
    rng = np.random.default_rng(21)
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=6)
    stages = [(hip.FX_FILTER,), (hip.FX_OCTAVE,),
              (hip.FX_TREMOLO, 1. + 9. * rng.random(n), rng.random(n), rng.random(n), .1 + .8 * rng.random(n)),
              (hip.FX_DELAY, .006 + .05 * rng.random(n), .9 * rng.random(n), rng.random(n)),
              (hip.FX_TRANSFORM, .5 + rng.random(n), rng.random(n) - .5), (hip.FX_GAIN, rng.random(n))]
    table, grid = library_table(hip)
    from tests.helpers import oracle_bank_from_patch
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.set_effects(stages)
    hb = hip.Bank(ctx, n, has_env=True, has_filter=True, effects=[s[0] for s in stages])
    hb.load_patch(patch)
    for s, st in enumerate(stages):
        for k, p in enumerate(st[1:]):
            hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    np.testing.assert_array_equal(hb.render_voices(nb, f64=True), want_voices)
    mix = np.concatenate([hb.render(5), hb.render(7)])
    assert np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-6
    np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(2, 0)), ob.fx_state[2, 0])
    np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(2, 1)), ob.fx_state[2, 1])
    np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(3, 0)), ob.fx_state[3, 0])
    np.testing.assert_array_equal(hb.get_state(hip.F_Z0), ob.z0)


# T Tremolo, D Delay; the element-wise kinds share one compiled slot: G Gain, X Transformer,
# O OctaveDistortion, B Bitcrusher, S Subsampler
PROGRAMS = ['T', 'D', 'G', 'TG', 'DG', 'TD', 'DT', 'X', 'OG', 'GBS', 'BT', 'SD', 'XDO', 'BS']


@pytest.mark.parametrize('program,has_filter,kernel',
                         [(p, f, 'compiled') for p in PROGRAMS for f in (True, False)]
                         + [('TD', True, 'general'), ('DG', False, 'general')])
def test_compiled_effect_programs_vs_oracle_bit_exact(hip, ctx, oracle, monkeypatch, program,
                                                      has_filter, kernel):
    """The Tremolo / Delay / element-wise programs voice_fx_kernel is compiled for (fx_tu.hip:
    Gain, Transformer, OctaveDistortion, Bitcrusher and Subsampler share the element-wise slot),
    behind the Filter or without one: 64 * 9 + 20 voices with notes, per-voice stage parameters, wave
    functions that differ from wavefront to wavefront (one wavefront mixed), some negative
    Tremolo rates (the exact remainder) -- float64 streams, stage state and delay lines
    positions equal the oracle's bit for bit, the mix over split launches within 1e-6.
    'general': the same through voice_mod_kernel's stage interpreter."""
    if kernel == 'general':
        monkeypatch.setenv('KLMI_FX_KERNEL', '0')
    n, nb = 64 * 9 + 20, 12
    rng = np.random.default_rng(40 + len(program) + 7 * PROGRAMS.index(program))
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=6)
    patch['has_filter'] = has_filter
    waves = np.repeat(np.arange(-(-n // 64)) % 3 + 1, 64)[:n]     # square, saw, triangle
    waves[128:192] = rng.integers(1, 4, 64)                         # a mixed wavefront
    patch['wave'] = waves.astype(np.int32)
    rate = 1. + 9. * rng.random(n)
    rate[300:310] = -rate[300:310]
    make = {'T': lambda: (hip.FX_TREMOLO, rate, rng.random(n), rng.random(n), .1 + .8 * rng.random(n)),
            'D': lambda: (hip.FX_DELAY, .006 + .05 * rng.random(n), .9 * rng.random(n), rng.random(n)),
            'G': lambda: (hip.FX_GAIN, 1.5 * rng.random(n)),
            'X': lambda: (hip.FX_TRANSFORM, .5 + rng.random(n), rng.random(n) - .5),
            'O': lambda: (hip.FX_OCTAVE,),
            'B': lambda: (hip.FX_BITCRUSH, rng.integers(0, 16, n)),
            'S': lambda: (hip.FX_SUBSAMPLE, 2 ** rng.integers(1, 8, n))}
    stages = ([(hip.FX_FILTER,)] if has_filter else []) + [make[k]() for k in program]
    table, grid = library_table(hip)
    from tests.helpers import oracle_bank_from_patch
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.set_effects(stages)
    hb = hip.Bank(ctx, n, has_env=True, has_filter=has_filter, effects=[s[0] for s in stages])
    hb.load_patch(patch)
    for s, st in enumerate(stages):
        for k, p in enumerate(st[1:]):
            hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    np.testing.assert_array_equal(hb.render_voices(nb, f64=True), want_voices)
    mix = np.concatenate([hb.render(5), hb.render(7)])
    assert np.abs(want_mix).max() > 1e-3
    assert np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-6
    for s, st in enumerate(stages):
        if st[0] == hip.FX_TREMOLO:
            np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(s, 0)), ob.fx_state[s, 0])
            np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(s, 1)), ob.fx_state[s, 1])
        if st[0] == hip.FX_DELAY:
            np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(s, 0)), ob.fx_state[s, 0])
    np.testing.assert_array_equal(hb.get_state(hip.F_PHASE), ob.phase)
    if has_filter:
        np.testing.assert_array_equal(hb.get_state(hip.F_Z0), ob.z0)
    hb.close()


def test_compiled_effect_program_under_a_stereo_mixer(hip, ctx, oracle):
    """Filter -> Tremolo -> Gain chains into a StereoMixer (the OUT_MIX_STEREO instantiation)."""
    n, nb = 200, 8
    rng = np.random.default_rng(77)
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=4)
    patch['stereo'] = True
    patch['pan'] = 2. * rng.random(n) - 1.
    stages = [(hip.FX_FILTER,), (hip.FX_TREMOLO, 1. + 9. * rng.random(n), rng.random(n), rng.random(n),
                                 .1 + .8 * rng.random(n)), (hip.FX_GAIN, rng.random(n))]
    table, grid = library_table(hip)
    from tests.helpers import oracle_bank_from_patch
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.set_effects(stages)
    hb = hip.Bank(ctx, n, has_env=True, has_filter=True, stereo=True, effects=[s[0] for s in stages])
    hb.load_patch(patch)
    for s, st in enumerate(stages):
        for k, p in enumerate(st[1:]):
            hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    want_mix, _ = ob.render(nb, patch.events)
    mix = np.concatenate([hb.render(3), hb.render(5)])
    assert mix.shape == want_mix.shape == (nb, 2, B)
    assert np.abs(mix - want_mix).max() <= 1e-6 and np.abs(want_mix).max() > 1e-3
    hb.close()


@pytest.mark.parametrize('length', [100, 200, 255, 17])
def test_delay_lines_shorter_than_a_buffer_vs_reference(hip, ctx, golden, length):
    """int(time * 44100) < 256 (fixtures delay_short_* from the real reference): the bank runs
    the Delay block by block with the RingBuffer's own data layout (mod_kernels.h delay_block)."""
    want = golden('effects_patched')['delay_short_%d' % length]
    nb = len(want)
    bank = saw_bank(hip, ctx, [(hip.FX_DELAY, (length + .5) / 44100., .6, .4)])
    got = bank.render_voices(nb, f64=True)[0].reshape(nb, B)
    np.testing.assert_array_equal(got, want)
    mix = np.concatenate([bank.render(5), bank.render(nb - 5)])[:, 0, :]
    assert report('effect: delay (line shorter than a buffer)', np.abs(mix - want).max()) <= 1e-5
    assert bank.get_state(hip.F_FX_STATE(0, 0))[0] == (nb * B) % length
    bank.close()


@pytest.mark.parametrize('stereo', [False, True])
def test_short_and_long_delay_lines_in_one_bank_vs_oracle(hip, ctx, oracle, stereo):
    """Filter -> Tremolo -> Delay -> Gain over 300 voices whose lines run from 3 to 900 samples
    (a third of them shorter than a buffer): float64 streams, positions and filter state equal
    the oracle's bit for bit, over split launches; and a bank that has rendered with long
    lines only refuses a short one afterwards."""
    n, nb = 300, 10
    rng = np.random.default_rng(51)
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=5)
    patch['stereo'] = stereo
    patch['pan'] = 2. * rng.random(n) - 1.
    lengths = rng.integers(256, 900, n)
    lengths[::3] = rng.integers(3, 256, len(lengths[::3]))
    stages = [(hip.FX_FILTER,),
              (hip.FX_TREMOLO, 1. + 9. * rng.random(n), rng.random(n), rng.random(n), .1 + .8 * rng.random(n)),
              (hip.FX_DELAY, (lengths + .5) / 44100., .9 * rng.random(n), rng.random(n)),
              (hip.FX_GAIN, rng.random(n))]
    table, grid = library_table(hip)
    from tests.helpers import oracle_bank_from_patch
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.set_effects(stages)
    hb = hip.Bank(ctx, n, has_env=True, has_filter=True, stereo=stereo, effects=[s[0] for s in stages])
    hb.load_patch(patch)
    for s, st in enumerate(stages):
        for k, p in enumerate(st[1:]):
            hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    np.testing.assert_array_equal(hb.render_voices(nb, f64=True), want_voices)
    mix = np.concatenate([hb.render(3), hb.render(7)])
    assert np.abs(want_mix).max() > 1e-3 and np.abs(mix - want_mix).max() <= 1e-6
    np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(2, 0)), ob.fx_state[2, 0])
    np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(1, 1)), ob.fx_state[1, 1])
    np.testing.assert_array_equal(hb.get_state(hip.F_Z0), ob.z0)
    hb.close()
    if not stereo:
        # a layout switch after the first render is refused where it is asked for, and the
        # refused call changes nothing: the bank renders on as if it had not been made
        late = saw_bank(hip, ctx, [(hip.FX_DELAY, .01, .5, .5)], n=2)
        twin = saw_bank(hip, ctx, [(hip.FX_DELAY, .01, .5, .5)], n=2)
        np.testing.assert_array_equal(late.render(2), twin.render(2))
        with pytest.raises(hip.NativeError) as e:
            late.set(hip.F_FX(0, 0), [.001, .01])
        assert e.value.code == hip.E_STATE and 'before the bank' in str(e.value)
        np.testing.assert_array_equal(late.get_state(hip.F_FX_STATE(0, 0)),
                                      twin.get_state(hip.F_FX_STATE(0, 0)))
        np.testing.assert_array_equal(late.render(3), twin.render(3))
        # a longer time is fine: the voice's line starts again empty, the other keeps its own
        late.set(hip.F_FX(0, 0), [.02], first=1)
        twin.set(hip.F_FX(0, 0), [.02], first=1)
        a, b = late.render_voices(4, f64=True), twin.render_voices(4, f64=True)
        np.testing.assert_array_equal(a, b)
        late.close()
        twin.close()


def test_crusher_and_ring_modulator_vs_oracle_many_voices(hip, ctx, oracle):
    """600 voices through Gain -> Bitcrusher -> RingModulator(non-sine Lfo) -> Subsampler ->
    Filter with per-voice parameters; Gain up to 1.5 so that the int16 wrap of the crusher
    is exercised.  The modulator's sine makes it 'a few ulp', not bit-exact."""
    n, nb = 600, 10
    rng = np.random.default_rng(33)
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=5)
    stages = [(hip.FX_GAIN, 1.5 * rng.random(n)), (hip.FX_BITCRUSH, rng.integers(0, 16, n)),
              (hip.FX_RINGMOD, 100. + 900. * rng.random(n), .5 + 8. * rng.random(n),
               50. * rng.random(n), rng.random(n), rng.integers(1, 4, n)),
              (hip.FX_SUBSAMPLE, 2 ** rng.integers(1, 8, n)), (hip.FX_FILTER,)]
    table, grid = library_table(hip)
    from tests.helpers import oracle_bank_from_patch
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.set_effects(stages)
    hb = hip.Bank(ctx, n, has_env=True, has_filter=True, effects=[s[0] for s in stages])
    hb.load_patch(patch)
    for s, st in enumerate(stages):
        for k, p in enumerate(st[1:]):
            hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    got = hb.render_voices(nb, f64=True)
    assert np.abs(got - want_voices).max() <= 1e-12
    mix = np.concatenate([hb.render(4), hb.render(6)])
    assert np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-6
    np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(2, 0)), ob.fx_state[2, 0])
    assert np.abs(hb.get_state(hip.F_FX_STATE(2, 1)) - ob.fx_state[2, 1]).max() <= 1e-9


def test_crusher_without_transcendentals_is_bit_exact(hip, ctx, oracle):
    """Gain -> Bitcrusher -> Subsampler on saws: integer work in between, exact."""
    n, nb = 300, 6
    rng = np.random.default_rng(34)
    stages = [(hip.FX_GAIN, 1.4 * rng.random(n)), (hip.FX_BITCRUSH, rng.integers(0, 16, n)),
              (hip.FX_SUBSAMPLE, 2 ** rng.integers(1, 8, n))]
    ob = oracle.OracleBank(n, False, False, False)
    ob.wave[:] = oracle.WAVE_IDS['sawtooth']
    ob.frequency[:] = 50. + 2000. * rng.random(n)
    ob.set_effects(stages)
    hb = saw_bank(hip, ctx, stages, n=n)
    hb.set(hip.F_FREQUENCY, ob.frequency)
    _, want = ob.render(nb, want_voices=True)
    np.testing.assert_array_equal(hb.render_voices(nb, f64=True), want)


def test_effect_argument_errors(hip, ctx):
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 2, effects=[hip.FX_DELAY, hip.FX_DELAY])          # one Delay per chain
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 2, has_filter=True, effects=[hip.FX_GAIN])        # Filter stage missing
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 2, effects=[hip.FX_FILTER])                       # has_filter missing
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 2, effects=[11])
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 2, effects=[hip.FX_SUBSAMPLE, hip.FX_SUBSAMPLE])
    odd = hip.Bank(ctx, 2, effects=[hip.FX_SUBSAMPLE, hip.FX_BITCRUSH, hip.FX_RINGMOD])
    for field, bad in ((hip.F_FX(0, 0), [3., 2.]), (hip.F_FX(0, 0), [256., 2.]),
                       (hip.F_FX(1, 0), [16., 2.]), (hip.F_FX(1, 0), [-1., 2.]),
                       (hip.F_FX(2, 4), [4., 0.])):
        with pytest.raises(hip.NativeError):
            odd.set(field, bad)
    odd.close()
    bank = hip.Bank(ctx, 2, effects=[hip.FX_GAIN, hip.FX_DELAY])
    with pytest.raises(hip.NativeError) as e:
        bank.set(hip.F_FX(1, 0), [2.5, 1.])
    assert 'to long' in str(e.value)                                    # effects.py:160-164
    with pytest.raises(hip.NativeError):
        bank.set(hip.F_FX(1, 0), [.00001, 1.])                          # shorter than a sample
    with pytest.raises(hip.NativeError):
        bank.set(hip.F_FX(2, 0), [1., 1.])                              # no such stage
    with pytest.raises(hip.NativeError):
        bank.get_state(hip.F_FX_STATE(5, 0))


@pytest.mark.parametrize('program,has_filter', [('R', True), ('R', False), ('RG', True), ('BR', False),
                                                ('RBS', True)])
def test_compiled_ring_modulator_programs(hip, ctx, oracle, monkeypatch, program, has_filter):
    """RingModulator (effects.py:618-662) with element-wise stages around it, compiled into
    voice_fx_kernel: float64 streams and all state bit for bit equal to the stage interpreter's
    (KLMI_FX_KERNEL=0: the same operations, the same sine polynomial), and within 1e-12 of the
    oracle (libm's sine).  Lfo wave functions that differ from wavefront to wavefront, one mixed
    wavefront, some negative Lfo rates and sweeps below zero (the exact remainder)."""
    n, nb = 64 * 6 + 11, 10
    rng = np.random.default_rng(900 + len(program) + 5 * has_filter)
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=5)
    patch['has_filter'] = has_filter
    patch['wave'] = (np.repeat(np.arange(-(-n // 64)) % 3 + 1, 64)[:n]).astype(np.int32)
    lfo_wave = np.repeat(np.arange(-(-n // 64)) % 4, 64)[:n]
    lfo_wave[64:128] = rng.integers(0, 4, 64)
    rate = .5 + 6. * rng.random(n)
    rate[200:210] = -rate[200:210]
    amount = 40. * rng.random(n)
    amount[300:320] = 900.                      # sweeps the modulator's frequency below zero
    make = {'R': lambda: (hip.FX_RINGMOD, 100. + 500. * rng.random(n), rate, amount, rng.random(n), lfo_wave),
            'G': lambda: (hip.FX_GAIN, 1.5 * rng.random(n)),
            'B': lambda: (hip.FX_BITCRUSH, rng.integers(0, 16, n)),
            'S': lambda: (hip.FX_SUBSAMPLE, 2 ** rng.integers(1, 8, n))}
    stages = ([(hip.FX_FILTER,)] if has_filter else []) + [make[k]() for k in program]
    table, grid = library_table(hip)
    from tests.helpers import oracle_bank_from_patch
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.set_effects(stages)
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)

    def run():
        hb = hip.Bank(ctx, n, has_env=True, has_filter=has_filter, effects=[s[0] for s in stages])
        hb.load_patch(patch)
        for s, st in enumerate(stages):
            for k, p in enumerate(st[1:]):
                hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
        voices = hb.render_voices(nb, f64=True)
        mix = np.concatenate([hb.render(4), hb.render(nb - 4)])
        s_ring = [s for s, st in enumerate(stages) if st[0] == hip.FX_RINGMOD][0]
        state = [hb.get_state(hip.F_PHASE), hb.get_state(hip.F_FX_STATE(s_ring, 0)),
                 hb.get_state(hip.F_FX_STATE(s_ring, 1))]
        hb.close()
        return voices, mix, state, s_ring
    compiled = run()
    monkeypatch.setenv('KLMI_FX_KERNEL', '0')
    general = run()
    np.testing.assert_array_equal(compiled[0], general[0])
    np.testing.assert_array_equal(compiled[1], general[1])
    for a, b in zip(compiled[2], general[2]):
        np.testing.assert_array_equal(a, b)
    crushed = 'B' in program     # a quantiser turns an ulp at a step into a whole step
    if not crushed:
        assert np.abs(compiled[0] - want_voices).max() <= 1e-12
    assert np.abs(compiled[1][:, 0, :] - want_mix[:, 0, :]).max() <= (1e-3 if crushed else 1e-6)
    s_ring = compiled[3]
    np.testing.assert_allclose(compiled[2][1], ob.fx_state[s_ring, 0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(compiled[2][2], ob.fx_state[s_ring, 1], rtol=0, atol=1e-9)
