"""BASELINE configs[3] at its stated size: 4,096 channels of Voice(sine, AR) -> Reverb ->
Mixer with 0.46 GB of comb delay lines in HBM (effects.py:564-615, _filters.c:257-280),
checked against the oracle -- mix, delay-line contents and ring positions -- plus the
properties the kernel's launch structure must not break (passes of 256 samples with ring ends
inside them, dry-bus chunks of 8 buffers), Reverb chains under a StereoMixer, reverbs that take
the sample-per-lane kernel, and the measured cost of float32 ring storage (the default)."""
import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, hip_bank_from_patch, library_table, oracle_bank_from_patch, report

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def oracle_reverb_channels(oracle, hip, patch, n_blocks, threads=8):
    """The patch's dry voices through the oracle bank, then one oracle Reverb per channel:
    -> (wet [n][n_blocks*256] float64, the Reverb objects)."""
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    _, dry = ob.render(n_blocks, patch.events, want_voices=True,
                       n_threads=min(threads, oracle.max_threads()))
    kw = patch['reverb']     # (its 'rings' entry is the device's storage choice: not the oracle's)
    reverbs = [oracle.OracleReverb(kw['decay'], kw['pre_delay'], kw['dry_wet'], kw['n_echos'])
               for _ in range(patch.n_voices)]
    wet = np.stack([r.run(dry[v]) for v, r in enumerate(reverbs)])
    return wet, reverbs


def oracle_rings(reverb):
    c = reverb.c
    rings = [np.ctypeslib.as_array(c.filters[k].data, shape=(c.filters[k].length,)).copy()
             for k in range(c.n_echos)]
    return rings, [int(c.filters[k].position) for k in range(c.n_echos)]


@pytest.mark.parametrize('wave,rings,ring_tol', [('sine', 'f32', 2e-6), ('sine', 'f64', 1e-12),
                                                 ('sawtooth', 'f64', 0.)])
def test_config4_full_size_vs_oracle(hip, ctx, oracle, wave, rings, ring_tol):
    """float32 ring storage is the default; float64 storage reproduces the reference's delay
    lines to the sine's ulps, and bit for bit under a sawtooth (plain float64 arithmetic end
    to end)."""
    n, nb = 4096, 24
    patch = P.reverb_patch(n_channels=n, n_blocks=nb, period=10, note_blocks=4)
    patch['reverb'] = dict(patch['reverb'], rings=rings)
    if wave == 'sawtooth':
        patch['wave'] = np.full(n, P.WAVE_SAWTOOTH, dtype=np.int32)
    rng = np.random.default_rng(7)
    patch['gain'] = .5 + rng.random(n)
    wet, reverbs = oracle_reverb_channels(oracle, hip, patch, nb)
    want = (patch.gain[:, None] * wet).sum(axis=0).reshape(nb, B) / n
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = bank.render(nb)
    assert mix.shape == (nb, 1, B) and np.abs(want).max() > 1e-2
    err = np.abs(mix[:, 0, :] - want).max()
    report('chain voice+reverb+mixer (4,096 channels)', err)
    assert err <= 1e-6
    taps, _ = hip.reverb_design(3., .03, 10)
    for v in (0, 63, 64, 2047, 4095):
        rings, pos = bank.reverb_state(v)
        o_rings, o_pos = oracle_rings(reverbs[v])
        assert list(pos) == o_pos == [(nb * B) % t for t in taps]
        for got, ref in zip(rings, o_rings):
            if ring_tol == 0.:
                np.testing.assert_array_equal(got, ref)
            else:
                assert np.abs(got - ref).max() <= ring_tol
    # the same span in calls that cut the 8-buffer dry-bus chunks at other places: per-sample
    # arithmetic and fold order do not change
    again = hip_bank_from_patch(hip, ctx, patch)
    parts = np.concatenate([again.render(k) for k in (3, 6, 8, 7)])
    np.testing.assert_array_equal(parts, mix)
    for v in (5, 4000):
        for a, b in zip(again.reverb_state(v)[0], bank.reverb_state(v)[0]):
            np.testing.assert_array_equal(a, b)


def test_reverb_chains_into_a_stereo_mixer_vs_oracle(hip, ctx, oracle):
    """(voice | Reverb) x N under a StereoMixer (mixer.py:95-104): each channel's Reverb
    output goes to both buses with gain * pan; N is not a multiple of the wavefront."""
    n, nb = 150, 20
    patch = P.reverb_patch(n_channels=n, n_blocks=nb, period=9, note_blocks=3)
    patch['stereo'] = True
    rng = np.random.default_rng(3)
    patch['pan'] = 2. * rng.random(n) - 1.
    patch['gain'] = .25 + rng.random(n)
    wet, _ = oracle_reverb_channels(oracle, hip, patch, nb)
    pan = np.array([oracle.panning_amplitudes(p) for p in patch.pan])     # [n][2]
    want = np.einsum('vc,vs->cs', patch.gain[:, None] * pan, wet) / n      # [2][span]
    want = want.reshape(2, nb, B).transpose(1, 0, 2)
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = np.concatenate([bank.render(11), bank.render(9)])
    assert mix.shape == (nb, 2, B) and np.abs(want).max() > 1e-2
    err = np.abs(mix - want).max()
    report('chain voice+reverb+stereo mixer', err)
    assert err <= 1e-6
    assert np.abs(mix[:, 0] - mix[:, 1]).max() > 1e-3      # two different buses


def test_other_comb_counts_and_short_combs_vs_oracle(hip, ctx, oracle):
    """nEchos != 10 and combs shorter than 256 samples run in the sample-per-lane kernel."""
    rng = np.random.default_rng(11)
    for kw in (dict(decay=2., pre_delay=.03, dry_wet=.5, n_echos=7),
               dict(decay=1., pre_delay=.002, dry_wet=.9, n_echos=10),
               dict(decay=.7, pre_delay=.0001, dry_wet=.6, n_echos=16)):
        n, nb = 70, 12
        patch = P.reverb_patch(n_channels=n, n_blocks=nb, period=5, note_blocks=2)
        patch['reverb'] = dict(kw, rings='f64')
        patch['wave'] = np.full(n, P.WAVE_SAWTOOTH, dtype=np.int32)
        patch['gain'] = .5 + rng.random(n)
        wet, reverbs = oracle_reverb_channels(oracle, hip, patch, nb)
        want = (patch.gain[:, None] * wet).sum(axis=0).reshape(nb, B) / n
        bank = hip_bank_from_patch(hip, ctx, patch)
        mix = np.concatenate([bank.render(5), bank.render(7)])[:, 0, :]
        assert np.abs(want).max() > 1e-2 and np.abs(mix - want).max() <= 1e-6
        for v in (0, 69):
            rings, pos = bank.reverb_state(v)
            o_rings, o_pos = oracle_rings(reverbs[v])
            assert list(pos) == o_pos
            for got, ref in zip(rings, o_rings):
                np.testing.assert_array_equal(got, ref)
        bank.close()


def test_float32_ring_storage_measured(hip, ctx, oracle, golden):
    """Reverb with float32 delay-line STORAGE (float64 arithmetic, one rounding per ring
    write: 88 instead of 168 bytes per channel-sample).  Measures what that costs against
    the float64 reference and holds it to the 1e-5 bound with margin: on the reference's own
    impulse / noise fixtures and on a 512-buffer config-4 run, per channel (a one-hot gain
    picks one channel's Reverb output out of a 64-channel bank) and on the comb outputs
    themselves (the delay lines)."""
    g = golden('reverb')
    noise = golden('inputs')['noise']
    impulse = np.zeros(24 * B)
    impulse[0] = 1.
    worst = 0.
    for pname, kw in (('default', {}), ('decay3', dict(decay=3., dry_wet=.7))):
        for iname, x in (('impulse', impulse), ('noise', noise)):
            want = g['%s_%s' % (pname, iname)]
            bank = hip.Bank(ctx, 1, has_env=False, has_reverb=True, reverb=dict(kw, rings='f32'))
            bank.set(hip.F_WAVE, [hip.WAVE_EXTERNAL])
            bank.set_input(x[None, :want.size])
            got = bank.render(want.size // B)[:, 0, :].ravel()
            worst = max(worst, np.abs(got - want).max())
    report('reverb (float32 ring storage, fixtures)', worst)
    assert worst <= 2e-6
    # config 4, 512 buffers (the envelopes are re-gated every 86), 64 channels
    n, nb = 64, 512
    patch = P.reverb_patch(n_channels=n, n_blocks=nb)
    wet, reverbs = oracle_reverb_channels(oracle, hip, patch, nb)
    patch['reverb'] = dict(patch['reverb'], rings='f32')
    out_err = ring_err = 0.
    for v in (0, 17, 42, 63):
        onehot = np.zeros(n)
        onehot[v] = 1.
        patch['gain'] = onehot
        bank = hip_bank_from_patch(hip, ctx, patch)
        bank.set_mix_divisor(1.)
        got = bank.render(nb)[:, 0, :].ravel()
        out_err = max(out_err, np.abs(got - wet[v]).max())
        rings, pos = bank.reverb_state(v)
        o_rings, o_pos = oracle_rings(reverbs[v])
        assert list(pos) == o_pos
        ring_err = max(ring_err, max(np.abs(a - b).max() for a, b in zip(rings, o_rings)))
        bank.close()
    report('reverb (float32 ring storage, config 4 x 512 buffers, per channel)', out_err)
    report('reverb (float32 ring storage, comb outputs in the delay lines)', ring_err)
    assert np.abs(wet).max() > .3
    assert out_err <= 2e-6 and ring_err <= 5e-6


@pytest.mark.parametrize('stereo,has_env', [(False, True), (True, True), (False, False)])
def test_fused_voice_reverb_kernel_equals_the_two_kernel_path(hip, ctx, monkeypatch, stereo, has_env):
    """reverb_voice_kernel (dry voice made inside the reverb wave, 1 / 2 / 4 voices per wave)
    against the voice kernel + reverb_kernel pair on a random bank with float64 rings: delay
    lines, ring positions and every voice state bit for bit, the mix to the float32 rounding of
    the different fold (per workgroup instead of per wave)."""
    n, spans = 1501, (3, 1, 7, 6)
    rng = np.random.default_rng(11)
    patch = P.reverb_patch(n_channels=n, n_blocks=sum(spans), period=5, note_blocks=2)
    patch['reverb'] = dict(patch['reverb'], rings='f64')
    patch['wave'] = rng.integers(0, 5, n).astype(np.int32)       # sine .. PWM
    patch['attack'] = .001 + .02 * rng.random(n)
    patch['decay'] = .05 * rng.random(n)
    patch['sustain'] = .2 + .8 * rng.random(n)
    patch['release'] = .002 + .05 * rng.random(n)
    patch['loop'] = rng.random(n) < .3
    patch['retrigger'] = rng.random(n) < .5
    patch['gain'] = .5 + rng.random(n)
    patch['stereo'] = stereo
    patch['has_env'] = has_env          # False: plain oscillators in front of the Reverbs
    patch['pan'] = rng.uniform(-1., 1., n)
    ev = patch.events.copy()
    ev['frequency'] *= 1. + .5 * rng.random(len(ev))
    ev['velocity'] = np.where(ev['kind'] == hip.EV_NOTE_ON, .2 + .8 * rng.random(len(ev)), 0.)
    patch['events'] = ev

    def run(fused, vpw):
        monkeypatch.setenv('KLMI_REVERB_FUSED', '1' if fused else '0')
        monkeypatch.setenv('KLMI_REVERB_VPW', str(vpw))
        bank = hip_bank_from_patch(hip, ctx, patch)
        bank.set(hip.F_DUTY, .1 + .8 * np.random.default_rng(3).random(n))
        mix = np.concatenate([bank.render(k) for k in spans])
        state = [bank.get_state(f) for f in (hip.F_PHASE, hip.F_ENV_VALUE, hip.F_ENV_STAGE,
                                             hip.F_ENV_ENABLED, hip.F_FREQUENCY, hip.F_AMPLITUDE)]
        rings = [bank.reverb_state(v) for v in (0, 1, 2, 3, 700, n - 2, n - 1)]
        bank.close()
        return mix, state, rings
    want_mix, want_state, want_rings = run(False, 1)
    assert np.abs(want_mix).max() > 1e-3
    for vpw in (1, 2, 4):
        mix, state, rings = run(True, vpw)
        assert np.abs(mix - want_mix).max() <= 2e-7
        for got, ref in zip(state, want_state):
            np.testing.assert_array_equal(got, ref)
        for (got, gpos), (ref, rpos) in zip(rings, want_rings):
            assert list(gpos) == list(rpos)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('fused', ['1', '0'])
def test_combs_barely_longer_than_a_pass_vs_oracle_bit_exact(hip, ctx, oracle, monkeypatch, fused):
    """pre_delay = 6 ms: the shortest comb is 269 samples, so a 256-sample pass reads slots that
    the SAME wave stored in the pass before, through another lane (and through the mirror /
    second-store slots at the ring's end).  Float64 rings under a sawtooth are plain float64
    arithmetic end to end: delay lines and positions must equal the oracle's bit for bit, in
    the fused kernel and in the two-kernel path, one voice per wave."""
    n, nb = 70, 9
    monkeypatch.setenv('KLMI_REVERB_FUSED', fused)
    monkeypatch.setenv('KLMI_REVERB_VPW', '1')
    patch = P.reverb_patch(n_channels=n, n_blocks=nb, period=4, note_blocks=2)
    patch['reverb'] = dict(decay=2., pre_delay=.006, dry_wet=.6, n_echos=10, rings='f64')
    patch['wave'] = np.full(n, P.WAVE_SAWTOOTH, dtype=np.int32)
    taps, _ = hip.reverb_design(2., .006, 10)
    assert 256 < min(taps) < 512
    wet, reverbs = oracle_reverb_channels(oracle, hip, patch, nb)
    want = wet.sum(axis=0).reshape(nb, B) / n
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = np.concatenate([bank.render(k) for k in (4, 5)])
    assert np.abs(want).max() > 1e-2 and np.abs(mix[:, 0, :] - want).max() <= 1e-6
    for v in (0, 1, 33, n - 1):
        rings, pos = bank.reverb_state(v)
        o_rings, o_pos = oracle_rings(reverbs[v])
        assert list(pos) == o_pos
        for got, ref in zip(rings, o_rings):
            np.testing.assert_array_equal(got, ref)
    bank.close()
