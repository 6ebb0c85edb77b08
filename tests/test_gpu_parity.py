"""GPU parity tests: the HIP library (through its C-ABI, via ctypes) against the
CPU oracle on the same seeded inputs and against the reference-generated
fixtures.  Bars (north_star): max-abs <= 1e-5 on float32 output vs the
reference's float64, reported per block type; the float64 diagnostics channel
(`render_voices(f64=True)`, envelope / comb batches) is held to bit-exactness
wherever the arithmetic is IEEE basic operations."""
import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import (B, REPORT, hip_bank_from_patch, library_table, oracle_bank_from_patch,
                           reference_table, report)

pytestmark = pytest.mark.gpu

TOL = 1e-5          # north_star bound, float32 output vs float64 reference
F32 = 6e-8          # one float32 rounding of a value <= 1
DT = 1. / 44100
WAVES = {'sine': 0, 'square': 1, 'sawtooth': 2, 'triangle': 3}

@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    print('\nparity report (max abs error per block type):')
    for k in sorted(REPORT):
        print('  %-28s %.3e' % (k, REPORT[k]))
    c.close()


# ---- oscillator ---------------------------------------------------------------------
@pytest.mark.parametrize('name', list(WAVES))
def test_oscillator_blocks_vs_golden(hip, ctx, golden, name):
    g = golden('oscillator_blocks')
    kept = list(g['kept_blocks'])
    bank = hip.Bank(ctx, 2, has_env=False)
    bank.set(hip.F_WAVE, [WAVES[name]] * 2)
    bank.set(hip.F_FREQUENCY, [440., 19999.])
    f64 = bank.render_voices(64, f64=True)
    f32 = bank.render_voices(64)
    for row, freq in enumerate((440, 19999)):
        want = g['%s_%d' % (name, freq)]
        got64 = f64[row].reshape(64, B)[kept]
        got32 = f32[row].reshape(64, B)[kept]
        if name != 'sine':   # IEEE basic operations only: bit-exact
            np.testing.assert_array_equal(got64, want)
        else:
            assert np.abs(got64 - want).max() <= 4e-15
        err = np.abs(got32 - want).max()
        report('oscillator/' + name, err)
        assert err <= F32
    bank.render(64)  # commit
    want_phase = [float(g['%s_%d_final_phase' % (name, f)]) for f in (440, 19999)]
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), want_phase)


def test_oscillator_many_frequencies_vs_oracle(hip, ctx, oracle):
    """Discontinuous waves: the wrap must fall on the same sample as NumPy's."""
    rng = np.random.default_rng(7)
    n = 512
    freq = np.concatenate([27.5 * 2. ** (9.5 * rng.random(n - 6)),
                           [0., 0.5, 44100. / 256, 22050., 19999., 4186.01]])
    start = np.concatenate([TAU_RAND(rng, n - 2), [0., 6.283185307179586]])
    for name, wid in list(WAVES.items()) + [('pwm', 4)]:
        bank = hip.Bank(ctx, n, has_env=False)
        bank.set(hip.F_WAVE, np.full(n, wid))
        bank.set(hip.F_FREQUENCY, freq)
        bank.set(hip.F_DUTY, np.full(n, .3))
        bank.set_state(hip.F_PHASE, start)
        got = bank.render_voices(6, f64=True)
        bank.render(6)
        phase = bank.get_state(hip.F_PHASE)
        for v in range(n):
            osc = oracle.OscillatorState(freq[v], name, start[v], duty=.3)
            want = np.concatenate([osc.block() for _ in range(6)])
            if name != 'sine':
                np.testing.assert_array_equal(got[v], want)
            else:
                assert np.abs(got[v] - want).max() <= 4e-15, (name, v, freq[v])
            assert phase[v] == osc.phase, (name, v, freq[v])


def TAU_RAND(rng, n):
    return 2. * np.pi * rng.random(n)


def test_negative_frequency_matches_numpy_mod(hip, ctx, oracle):
    bank = hip.Bank(ctx, 3, has_env=False)
    bank.set(hip.F_WAVE, [2, 2, 2])
    freq = [-440., -0.25, -12345.678]
    bank.set(hip.F_FREQUENCY, freq)
    got = bank.render_voices(3, f64=True)
    for v in range(3):
        osc = oracle.OscillatorState(freq[v], 'sawtooth')
        want = np.concatenate([osc.block() for _ in range(3)])
        np.testing.assert_array_equal(got[v], want)


# ---- envelope -------------------------------------------------------------------------
ENV_CASES = ['adsr_default', 'adsr_regate', 'ar_loop_5s', 'ar_zero_attack', 'd_only', 'r_only',
             'retrigger_on', 'retrigger_off', 'loop_gate_off', 'overshoot_clip_low',
             'overshoot_large', 'double_events']


@pytest.mark.parametrize('case', ENV_CASES)
def test_envelope_batch_bit_exact_vs_golden(hip, ctx, golden, case):
    """klmi_env_* (the _envelope.Envelope seam) block by block, float64, ==."""
    g = golden('envelope')
    a, d, s, r, over, retrig, loop = g[case + '_params']
    env = hip.EnvelopeBatch(ctx, 3, a, d, s, r, DT, over, bool(retrig), bool(loop))
    script = {}
    for blk, gate in g[case + '_script']:
        script.setdefault(int(blk), []).append(bool(gate))
    n_blocks = int(g[case + '_n_blocks'])
    step = 16 if n_blocks > 100 else 1   # several blocks per call where no gate falls inside
    samples = []
    b = 0
    while b < n_blocks:
        for gate in script.get(b, ()):
            env.gate(gate)
        nb = 1
        while nb < step and b + nb < n_blocks and (b + nb) not in script:
            nb += 1
        out = env.sample(nb * B)
        np.testing.assert_array_equal(out[0], out[2])
        samples.append(out[1])
        b += nb
    got = np.concatenate(samples)
    np.testing.assert_array_equal(got[::int(g[case + '_decim'])], g[case + '_samples'])
    np.testing.assert_array_equal(got[B - 1::B], g[case + '_block_last'])
    assert bool(env.active()[0]) == bool(g[case + '_active'][-1])
    report('envelope', 0.)


@pytest.mark.parametrize('case', ENV_CASES)
def test_envelope_in_bank_vs_golden(hip, ctx, golden, case):
    """The fused kernel's envelope path (wave ONE, amplitude 1): float64 == and
    float32 within one rounding."""
    g = golden('envelope')
    a, d, s, r, over, retrig, loop = g[case + '_params']
    n_blocks = int(g[case + '_n_blocks'])
    bank = hip.Bank(ctx, 1, has_env=True)
    bank.set(hip.F_WAVE, [hip.WAVE_ONE])
    for f, v in ((hip.F_ATTACK, a), (hip.F_DECAY, d), (hip.F_SUSTAIN, s), (hip.F_RELEASE, r),
                 (hip.F_OVERSHOOT, over), (hip.F_RETRIGGER, retrig), (hip.F_LOOP, loop)):
        bank.set(f, [v])
    for blk, gate in g[case + '_script']:
        bank.event(0, int(blk), hip.EV_NOTE_ON if gate else hip.EV_NOTE_OFF, 1., 440.)
    got = bank.render_voices(n_blocks, f64=True)[0]
    dec = int(g[case + '_decim'])
    np.testing.assert_array_equal(got[::dec], g[case + '_samples'])
    got32 = bank.render_voices(n_blocks)[0]
    err = np.abs(got32[::dec] - g[case + '_samples']).max()
    report('envelope', err)
    assert err <= F32
    # commit in two calls: state carried across launches, queue re-based
    first = n_blocks // 3
    bank.render(first)
    bank.render(n_blocks - first)
    assert bank.get_state(hip.F_ENV_VALUE)[0] == g[case + '_block_last'][-1]
    assert bool(bank.get_state(hip.F_ENV_STAGE)[0]) == bool(g[case + '_active'][-1])


def test_reference_envelope_kats(hip, ctx):
    """tests/test_audio__envelope.py:8-78 of the reference, on klmi_env_*."""
    env = hip.EnvelopeBatch(ctx, 1, 1., 1., .5, 1., dt=.1)
    assert env.sample(0).shape == (1, 0)
    with pytest.raises(hip.NativeError):
        env.sample(-1)
    np.testing.assert_array_equal(env.sample(10)[0], np.zeros(10))
    env.gate(True)
    assert env.sample(10)[0, -1] == 1.
    assert env.sample(10)[0, -1] == .5
    np.testing.assert_array_equal(env.sample(10)[0], np.full(10, .5))
    env.gate(False)
    assert env.sample(10)[0, -1] == 0.
    np.testing.assert_array_equal(env.sample(10)[0], np.zeros(10))
    assert not env.active()[0]
    for field in (hip.F_ATTACK, hip.F_DECAY, hip.F_RELEASE, hip.F_OVERSHOOT):
        with pytest.raises(hip.NativeError) as e:
            env.set(field, [-1.])
        assert e.value.code == hip.E_ARG
    for bad in (-.1, 1.1):
        with pytest.raises(hip.NativeError):
            env.set(hip.F_SUSTAIN, [bad])
    env.set(hip.F_OVERSHOOT, [0.])
    assert env.get(hip.F_OVERSHOOT)[0] == 1e-9       # clipped, _envelope.c:374
    env.set(hip.F_ATTACK, [.25])
    assert env.get(hip.F_ATTACK)[0] == .25


# ---- Filter (biquad) ----------------------------------------------------------------------
def test_filter_design_helpers(hip, golden):
    g = golden('filter')
    table, grid = library_table(hip)
    np.testing.assert_allclose(grid, g['grid'], rtol=2.3e-16, atol=0)
    ref = g['table'][:, [0, 1, 2, 4, 5]]
    np.testing.assert_allclose(table, ref, rtol=4e-15, atol=1e-15)
    for req, want in zip(g['index_requests'], g['index_results']):
        assert hip.filter_grid_index(req) == want


@pytest.mark.parametrize('src', ['noise', 'saw'])
def test_filter_block_vs_golden_and_oracle(hip, ctx, oracle, golden, src):
    g = golden('filter')
    x = golden('inputs')['noise'][:8 * B] if src == 'noise' else g['saw_input']
    cutoffs = g['cutoffs']
    n = len(cutoffs)
    bank = hip.Bank(ctx, n, has_env=False, has_filter=True)
    bank.set(hip.F_WAVE, np.full(n, hip.WAVE_EXTERNAL))
    bank.set(hip.F_CUTOFF, cutoffs)
    bank.set_input(np.tile(x, (n, 1)))
    got64 = bank.render_voices(8, f64=True)
    got32 = bank.render_voices(8)
    table, grid = library_table(hip)
    for v, c in enumerate(cutoffs):
        row = table[hip.filter_grid_index(c)]
        want, z = oracle.lfilter2(row[:3], np.r_[1., row[3:]], x, np.zeros(2))
        np.testing.assert_array_equal(got64[v], want)          # same coefficients: bit-exact
        ref = g['%s_%g' % (src, c)]
        assert np.abs(got64[v] - ref).max() <= 1e-11             # scipy's design: ulp-level coefs
        err = np.abs(got32[v] - ref).max()
        report('biquad', err)
        assert err <= 2 * F32 * max(1., np.abs(ref).max())
    bank.render(8)
    for v, c in enumerate(cutoffs):
        z = g['%s_%g_state' % (src, c)]
        np.testing.assert_allclose(bank.get_state(hip.F_Z0)[v], z[0], rtol=0, atol=1e-11)
        np.testing.assert_allclose(bank.get_state(hip.F_Z1)[v], z[1], rtol=0, atol=1e-11)


def test_filter_cutoff_sweep_keeps_state(hip, ctx, golden):
    g = golden('filter')
    x = golden('inputs')['noise'][:8 * B]
    bank = hip.Bank(ctx, 1, has_env=False, has_filter=True)
    bank.set(hip.F_WAVE, [hip.WAVE_EXTERNAL])
    for blk, f in g['sweep_schedule']:
        bank.event(0, int(blk), hip.EV_SET_CUTOFF, 0., float(f))
    bank.set_input(x[None, :])
    got = bank.render_voices(8, f64=True)[0]
    assert np.abs(got - g['sweep_noise']).max() <= 1e-11
    report('biquad', np.abs(got.astype(np.float32) - g['sweep_noise']).max())


# ---- ring-buffer filters ---------------------------------------------------------------------
@pytest.mark.parametrize('kind', ['forward', 'backward', 'echo'])
def test_ring_filters_bit_exact_vs_golden(hip, ctx, golden, kind):
    g = golden('ring_filters')
    x = golden('inputs')['noise']
    kid = {'forward': 0, 'backward': 1, 'echo': 2}[kind]
    for K in (1, 7, 234, 1327):
        total = 1664 if K == 1327 else 640
        for bs in (256, 123):
            f = hip.CombBatch(ctx, kid, 2, K, .9 if K != 7 else .5)
            ys, pos = [], 0
            while pos < total:
                n = min(bs, total - pos)
                y = f.filter(np.tile(x[pos:pos + n], (2, 1)))
                np.testing.assert_array_equal(y[0], y[1])
                ys.append(y[0])
                pos += n
            np.testing.assert_array_equal(np.concatenate(ys), g['%s_K%d_bs%d' % (kind, K, bs)])
    report('comb/' + kind, 0.)


def test_reference_ring_filter_kats(hip, ctx):
    """tests/test_audio_filters.py:24-73 of the reference, on klmi_comb_*."""
    K, alpha, bs = 234, .9, 123
    x = np.zeros(3 * K)
    x[0] = 1.

    def run(kind):
        f = hip.CombBatch(ctx, kind, 1, K, alpha)
        return np.concatenate([f.filter(x[None, p:p + bs])[0] for p in range(0, x.size, bs)])

    y = run(hip.COMB_FORWARD)
    assert y[0] == 1. and y[K] == alpha and y[2 * K] == 0.
    y = run(hip.COMB_BACKWARD)
    assert y[0] == 1. and y[K] == alpha and y[2 * K] == alpha * alpha
    y = run(hip.COMB_ECHO)
    assert y[0] == 0. and y[K] == 1. and y[2 * K] == alpha
    with pytest.raises(hip.NativeError) as e:
        hip.CombBatch(ctx, 0, 1, 60 * 44100 + 1)
    assert e.value.code == hip.E_MEMORY


# ---- mixers ------------------------------------------------------------------------------------
def test_mixer_vs_golden(hip, ctx, golden):
    g = golden('mixer')
    x = golden('inputs')['mixer_inputs']
    for n in (1, 2, 3, 64):
        bank = hip.Bank(ctx, n, has_env=False)
        bank.set(hip.F_WAVE, np.full(n, hip.WAVE_EXTERNAL))
        bank.set(hip.F_GAIN, g['gains'][:n])
        bank.set_input(x[:n])
        got = bank.render(1)[0, 0]
        err = np.abs(got - g['mono_%d' % n]).max()
        report('mixer', err)
        assert err <= 4 * F32 * np.abs(g['gains']).max()
    bank = hip.Bank(ctx, 20, has_env=False, stereo=True)
    bank.set(hip.F_WAVE, np.full(20, hip.WAVE_EXTERNAL))
    bank.set(hip.F_GAIN, g['gains'][:20])
    bank.set(hip.F_PAN, g['stereo_20_pannings'])
    bank.set_input(x[:20])
    got = bank.render(1)[0]
    err = np.abs(got - g['stereo_20']).max()
    report('stereo_mixer', err)
    assert err <= 4 * F32 * np.abs(g['gains']).max()
    for mode in ('linear', 'constant_power', 'mixture'):
        for p, want, want_law in zip(g['pan_levels'], g['pan_' + mode],
                                     g['pan_%s_law_m3' % mode]):
            np.testing.assert_allclose(hip.panning_amplitudes(p, mode), want, rtol=0, atol=2e-16)
            np.testing.assert_allclose(hip.panning_amplitudes(p, mode, -3.), want_law, rtol=0,
                                       atol=4e-16)


# ---- chains ----------------------------------------------------------------------------------------
def test_single_voice_chain_vs_golden(hip, ctx, golden):
    g = golden('chains')
    bank = hip.Bank(ctx, 1, has_env=True)
    bank.set(hip.F_WAVE, [hip.WAVE_SAWTOOTH])
    for f, v in ((hip.F_ATTACK, .01), (hip.F_DECAY, .05), (hip.F_SUSTAIN, .6),
                 (hip.F_RELEASE, .04)):
        bank.set(f, [v])
    bank.event(0, 0, hip.EV_NOTE_ON, .8, 220.)
    bank.event(0, 6, hip.EV_NOTE_OFF, 0., 220.)
    bank.event(0, 9, hip.EV_NOTE_ON, .5, 329.6275569128699)
    got64 = bank.render_voices(16, f64=True)[0].reshape(16, B)
    np.testing.assert_array_equal(got64, g['voice_saw_adsr'])
    got = bank.render(16)[:, 0, :]
    err = np.abs(got - g['voice_saw_adsr']).max()
    report('voice', err)
    assert err <= F32


def test_config3_mini_vs_golden(hip, ctx, golden):
    g = golden('chains')
    patch = P.polyphonic_saw_patch(n_voices=48, n_blocks=40, off_base=6, off_span=12)
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = bank.render(patch.n_blocks)[:, 0, :]
    err = np.abs(mix - g['config3_mini_mix']).max()
    report('chain osc+ADSR+biquad+mixer', err)
    assert err <= 1e-6
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), g['config3_mini_final_phase'])


def test_config2_haunting_vs_golden(hip, ctx, golden):
    g = golden('chains')
    patch = P.haunting_patch()
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = np.concatenate([bank.render(1024), bank.render(1024)])
    err = np.abs(mix[g['config2_kept_blocks']] - g['config2_mix']).max()
    report('chain looping AR + stereo mixer', err)
    assert err <= 1e-6


def test_config4_mini_reverb_vs_golden(hip, ctx, golden):
    g = golden('chains')
    patch = P.reverb_patch(n_channels=6, n_blocks=72, period=24, note_blocks=5)
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = np.concatenate([bank.render(30), bank.render(42)])[:, 0, :]
    err = np.abs(mix[g['config4_kept_blocks']] - g['config4_mini_mix']).max()
    report('chain voice+reverb+mixer', err)
    assert err <= 1e-6


def test_reverb_block_vs_golden(hip, ctx, golden):
    g = golden('reverb')
    noise = golden('inputs')['noise']
    impulse = np.zeros(24 * B)
    impulse[0] = 1.
    for pname, kw in (('default', {}), ('decay3', dict(decay=3., dry_wet=.7))):
        for iname, x in (('impulse', impulse), ('noise', noise)):
            bank = hip.Bank(ctx, 1, has_env=False, has_reverb=True, reverb=kw)
            bank.set(hip.F_WAVE, [hip.WAVE_EXTERNAL])
            bank.set_input(x[None, :])
            got = np.concatenate([bank.render(10), bank.render(14)])[:, 0, :].ravel()
            err = np.abs(got - g['%s_%s' % (pname, iname)]).max()
            report('reverb', err)
            assert err <= 4 * F32
    # taps shorter than a block (time slabs < 256 samples) and dryWet clipping
    bank = hip.Bank(ctx, 1, has_env=False, has_reverb=True,
                    reverb=dict(decay=.5, pre_delay=.001, dry_wet=1.5, n_echos=4))
    bank.set(hip.F_WAVE, [hip.WAVE_EXTERNAL])
    bank.set_input(noise[None, :6 * B])
    got = bank.render(6)[:, 0, :].ravel()
    err = np.abs(got - g['short_noise']).max()
    report('reverb', err)
    assert err <= 4 * F32 * np.abs(g['short_noise']).max()
    taps, alphas = hip.reverb_design(1.5, .03, 10)
    np.testing.assert_array_equal(taps, g['taps_default'])
    np.testing.assert_allclose(alphas, g['default_alphas'], rtol=2e-16)
    rings, pos = bank.reverb_state(0)
    assert [len(r) for r in rings] == list(g['short_taps'])
    assert list(pos) == [(6 * B) % t for t in g['short_taps']]


# ---- larger cases vs the oracle, and size-independent properties ------------------------------------
def test_polyphonic_patch_4096_vs_oracle(hip, ctx, oracle):
    patch = P.polyphonic_saw_patch(n_voices=4096, n_blocks=24, off_base=4, off_span=12)
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    want_mix, want_voices = ob.render(patch.n_blocks, patch.events, want_voices=True)
    bank = hip_bank_from_patch(hip, ctx, patch)
    got_voices = bank.render_voices(patch.n_blocks, first=1000, count=300, f64=True)
    np.testing.assert_array_equal(got_voices, want_voices[1000:1300])   # saw chain: bit-exact
    mix = bank.render(patch.n_blocks)
    err = np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max()
    report('chain osc+ADSR+biquad+mixer', err)
    assert err <= 1e-6
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), ob.phase)
    stage, value, enabled = ob.env_state()
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_VALUE), value)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_STAGE), stage)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_ENABLED), enabled)
    np.testing.assert_array_equal(bank.get_state(hip.F_Z0), ob.z0)
    np.testing.assert_array_equal(bank.get_state(hip.F_Z1), ob.z1)


@pytest.mark.parametrize('n,has_filter', [(5000, True), (20001, False)])
def test_pipelined_kernel_with_two_group_sizes_vs_oracle(hip, ctx, oracle, n, has_filter):
    """Bank sizes for which the pipelined kernel's grid is not a whole number of workgroups
    per compute unit: it then runs workgroups of two sizes (pipe_tu.hip pipe_groups; on 256
    CUs 5,000 voices are 482 x 8 + 286 x 4, 20,001 are 483 x 32 + 285 x 16 with a ragged
    last one).  Voice streams from both kinds of workgroup, mix and state vs the oracle."""
    nb = 10
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=3, off_span=5)
    patch['has_filter'] = has_filter
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True,
                                      n_threads=min(8, oracle.max_threads()))
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = np.concatenate([bank.render(4), bank.render(6)])
    assert np.abs(mix[:, 0, :] - want_mix[:, 0, :]).max() <= 1e-6
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), ob.phase)
    stage, value, enabled = ob.env_state()
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_VALUE), value)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_STAGE), stage)
    if has_filter:
        np.testing.assert_array_equal(bank.get_state(hip.F_Z0), ob.z0)
    # per-voice rows are a launch over the whole bank too (preview from the initial state)
    fresh = hip_bank_from_patch(hip, ctx, patch)
    got = fresh.render_voices(nb, f64=True)
    np.testing.assert_array_equal(got, want_voices)


def test_full_size_config3_properties(hip, ctx, oracle):
    """BASELINE config 3 at its full 65,536 voices (few blocks): shard-sum
    invariance, launch-split invariance, linearity in the gains, and the mix
    against the (threaded) oracle."""
    n, nb = 65536, 6
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=2, off_span=3)
    full = hip_bank_from_patch(hip, ctx, patch)
    mix32 = full.render(nb)
    mix = mix32.astype(np.float64)
    # (1) two shards with divisor 1, summed and divided by N == the full bank
    acc = np.zeros_like(mix)
    for first in (0, n // 2):
        shard = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=2, off_span=3,
                                       first_voice=first, count=n // 2)
        sb = hip_bank_from_patch(hip, ctx, shard)
        sb.set_mix_divisor(1.)
        acc += sb.render(nb)
    assert np.abs(acc / n - mix).max() <= 1e-6
    # (2) the same span rendered as 1+2+3 blocks carries state exactly
    again = hip_bank_from_patch(hip, ctx, patch)
    parts = np.concatenate([again.render(1), again.render(2), again.render(3)])
    np.testing.assert_array_equal(parts, mix32)
    np.testing.assert_array_equal(again.get_state(hip.F_Z0), full.get_state(hip.F_Z0))
    # (3) linearity: doubling every gain doubles the bus exactly (power of two)
    dbl = hip_bank_from_patch(hip, ctx, patch)
    dbl.set(hip.F_GAIN, 2. * patch.gain)
    np.testing.assert_array_equal(dbl.render(nb), 2 * mix32)
    # (4) against the oracle (OpenMP over voices; bus re-association only)
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    want, _ = ob.render(nb, patch.events, n_threads=min(8, oracle.max_threads()))
    err = np.abs(mix - want).max()
    report('chain osc+ADSR+biquad+mixer (65,536 voices)', err)
    assert err <= 1e-6
    np.testing.assert_array_equal(full.get_state(hip.F_PHASE), ob.phase)


def test_full_size_config3_long_run_vs_oracle(hip, ctx, oracle):
    """65,536 voices x 48 buffers (every envelope stage visited: notes end between buffers 8
    and 40) in launches of 1 + 20 + 27 buffers: the mix and every voice's final phase, envelope
    and filter state against the threaded oracle."""
    n, nb = 65536, 48
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=8, off_span=32)
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    want, _ = ob.render(nb, patch.events, n_threads=min(16, oracle.max_threads()))
    bank = hip_bank_from_patch(hip, ctx, patch)
    mix = np.concatenate([bank.render(1), bank.render(20), bank.render(27)])
    err = np.abs(mix[:, 0, :] - want[:, 0, :]).max()
    report('chain osc+ADSR+biquad+mixer (65,536 voices, 48 buffers)', err)
    assert err <= 1e-6 and np.abs(want).max() > 1e-3
    stage, value, enabled = ob.env_state()
    assert len(set(stage.tolist())) >= 2          # some voices still sound, others are off
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), ob.phase)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_VALUE), value)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_STAGE), stage)
    np.testing.assert_array_equal(bank.get_state(hip.F_Z0), ob.z0)
    np.testing.assert_array_equal(bank.get_state(hip.F_Z1), ob.z1)


def test_render_device_is_ordered_on_the_callers_stream(hip, ctx):
    """klmi_ctx_create(stream=NULL) is the default stream: a torch copy issued
    right after render_device, with no synchronisation in between, sees the bus."""
    import torch
    patch = P.polyphonic_saw_patch(n_voices=20000, n_blocks=64, off_base=20, off_span=30)
    want = hip_bank_from_patch(hip, ctx, patch).render(64)
    busy = torch.zeros(1 << 27, dtype=torch.float32, device='cuda')
    for side in (False, True):
        stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
        if side:
            # keep the NULL stream busy while the bank is created and rendered on the
            # (non-blocking) side stream: the creation-time clears are ordered on the context's
            # stream, so nothing of the bank may wait for -- or race with -- this work
            for _ in range(40):
                busy.add_(1.)
        with torch.cuda.stream(stream):
            c2 = hip.Context(0, stream.cuda_stream)
            bank = hip_bank_from_patch(hip, c2, patch)
            bus = torch.zeros((64, 1, B), dtype=torch.float32, device='cuda')
            host = torch.zeros((64, 1, B), dtype=torch.float32).pin_memory()
            bank.render_device(64, bus.data_ptr())
            host.copy_(bus, non_blocking=True)
            stream.synchronize()
            np.testing.assert_array_equal(host.numpy(), want)
            bank.close()
            c2.close()


def test_preview_leaves_state_and_queue_untouched(hip, ctx):
    patch = P.polyphonic_saw_patch(n_voices=300, n_blocks=12, off_base=3, off_span=6)
    a = hip_bank_from_patch(hip, ctx, patch)
    b = hip_bank_from_patch(hip, ctx, patch)
    v1 = a.render_voices(12)
    v2 = a.render_voices(12)
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_array_equal(a.render(12), b.render(12))
    # the streams previewed are the ones mixed: sum in float64 / N
    want = v1.astype(np.float64).sum(axis=0).reshape(12, B) / 300
    got = hip_bank_from_patch(hip, ctx, patch).render(12)[:, 0, :]
    assert np.abs(got - want).max() <= 1e-6


def test_argument_errors(hip, ctx):
    bank = hip.Bank(ctx, 4)
    for field, bad in ((hip.F_ATTACK, -1.), (hip.F_DECAY, -1.), (hip.F_RELEASE, -1.),
                       (hip.F_SUSTAIN, 1.5), (hip.F_WAVE, 9.)):
        with pytest.raises(hip.NativeError) as e:
            bank.set(field, [bad], first=1)
        assert e.value.code == hip.E_ARG
    with pytest.raises(hip.NativeError):
        bank.set(hip.F_GAIN, np.ones(5))              # range outside the bank
    with pytest.raises(hip.NativeError):
        bank.event(4, 0, hip.EV_NOTE_ON, 1., 440.)     # voice out of range
    with pytest.raises(hip.NativeError):
        bank.render(0)
    bank.set(hip.F_WAVE, [hip.WAVE_EXTERNAL] * 4)
    with pytest.raises(hip.NativeError) as e:
        bank.render(1)                                 # external input never supplied
    assert e.value.code == hip.E_STATE
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 0)
    with pytest.raises(hip.NativeError):
        hip.Context(99)


def test_voice_rows_on_aligned_and_unaligned_device_buffers(hip, ctx):
    """Per-voice output takes 16-byte stores (4 voice rows per instruction) when the rows are
    16-byte aligned and an element-wise loop when they are not; a voice count that is no
    multiple of 64 leaves a partial tile.  All of them must write the same rows: a caller's
    device buffer at offsets of 0, 1 and 2 floats, 200 voices (3 full tiles + 8 voices),
    rows compared bit for bit with the host-buffer call, guard words around the buffer intact."""
    import torch
    n, nb = 200, 3
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=1, off_span=2)
    want = hip_bank_from_patch(hip, ctx, patch).render_voices(nb)
    assert want.shape == (n, nb * B) and np.abs(want).max() > .1
    for offset in (0, 1, 2):
        bank = hip_bank_from_patch(hip, ctx, patch)
        buf = torch.full((n * nb * B + 8,), 7., dtype=torch.float32, device='cuda')
        view = buf[offset:offset + n * nb * B]
        bank.render_voices_device(nb, view.data_ptr())
        ctx.synchronize()
        host = buf.cpu().numpy()
        np.testing.assert_array_equal(host[offset:offset + n * nb * B].reshape(n, nb * B), want)
        assert (host[:offset] == 7.).all() and (host[offset + n * nb * B:] == 7.).all()
    # a sub-range of voices that starts inside a tile
    bank = hip_bank_from_patch(hip, ctx, patch)
    np.testing.assert_array_equal(bank.render_voices(nb, first=37, count=100), want[37:137])
    np.testing.assert_array_equal(bank.render_voices(nb, first=0, count=n, f64=True).astype(np.float32),
                                  want)
