"""Seeded random chains against the oracle: every draw combines features that the other test
files cover one at a time -- bank size, envelope / filter / stereo, wave functions per voice,
an audio-rate modulation input, the block-rate cutoff Lfo, a stage program of up to three
effects (delay lines on either side of one buffer), the kernel family (the pipelined kernel for
small banks or one lane per voice) and the way the span is cut into launches.  Only IEEE basic
operations are drawn (no sine, no tanh), so float64 voice streams and all state must equal the
oracle's bit for bit; the float32 mix within 1e-6.  Everything goes through the C-ABI."""
import os

import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, library_table, oracle_bank_from_patch

pytestmark = pytest.mark.gpu

SIZES = (1, 37, 64, 65, 200, 333, 513)


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def draw(hip, rng):
    n = int(rng.choice(SIZES))
    nb = int(rng.integers(3, 11))
    cfg = dict(n=n, nb=nb, has_env=bool(rng.random() < .85), has_filter=bool(rng.random() < .6),
               stereo=bool(rng.random() < .3))
    target = int(rng.choice([hip.MOD_NONE, hip.MOD_NONE, hip.MOD_NONE, hip.MOD_FM, hip.MOD_FREQUENCY,
                             hip.MOD_DUTY]))
    cfg['target'] = target
    cfg['cutoff_lfo'] = bool(cfg['has_filter'] and rng.random() < .25)
    if target == hip.MOD_DUTY:
        wave = np.full(n, P.WAVE_PWM)
    elif rng.random() < .5:          # wavefronts that agree on their wave function
        wave = np.repeat(rng.integers(1, 5, -(-n // 64)), 64)[:n]
    else:
        wave = rng.integers(1, 5, n)
    cfg['wave'] = wave.astype(np.int32)
    cfg['duty'] = .1 + .8 * rng.random(n)
    # a stage program: at most one stage of each kind that keeps state
    kinds = []
    pool = ['T', 'D', 'G', 'X', 'O', 'B', 'S']
    for _ in range(int(rng.choice([0, 0, 1, 2, 3]))):
        k = str(rng.choice(pool))
        kinds.append(k)
        if k in 'TDS':
            pool.remove(k)
    cfg['program'] = ''.join(kinds)
    cfg['short_delay'] = bool(rng.random() < .3)
    cfg['pipe_off'] = bool(rng.random() < .5)
    cuts = sorted(set(int(c) for c in rng.integers(1, nb, int(rng.integers(0, 3)))))
    cfg['launches'] = [b - a for a, b in zip([0] + cuts, cuts + [nb])]
    return cfg


def stage_of(hip, rng, k, n, short_delay):
    if k == 'T':
        rate = 1. + 9. * rng.random(n)
        rate[rng.random(n) < .05] *= -1.
        return (hip.FX_TREMOLO, rate, rng.random(n), rng.random(n), .1 + .8 * rng.random(n))
    if k == 'D':
        time = (3 + rng.integers(0, 250, n)) / 44100. + 1e-7 if short_delay else .006 + .04 * rng.random(n)
        return (hip.FX_DELAY, time, .9 * rng.random(n), rng.random(n))
    if k == 'G':
        return (hip.FX_GAIN, 1.5 * rng.random(n))
    if k == 'X':
        return (hip.FX_TRANSFORM, .5 + rng.random(n), rng.random(n) - .5)
    if k == 'O':
        return (hip.FX_OCTAVE,)
    if k == 'B':
        return (hip.FX_BITCRUSH, rng.integers(0, 16, n))
    return (hip.FX_SUBSAMPLE, 2 ** rng.integers(1, 8, n))


# KLANG_FUZZ_SEEDS widens a hunt; the default keeps the file at a few seconds
@pytest.mark.parametrize('seed', range(int(os.environ.get('KLANG_FUZZ_SEEDS', '60'))))
def test_random_chain_vs_oracle_bit_exact(hip, ctx, oracle, monkeypatch, seed):
    rng = np.random.default_rng(9000 + seed)
    cfg = draw(hip, rng)
    n, nb = cfg['n'], cfg['nb']
    if cfg['pipe_off']:
        monkeypatch.setenv('KLMI_PIPE_VOICES_LOG2', '-1')
    patch = P.polyphonic_saw_patch(n_voices=n, n_blocks=nb, seed=seed, off_base=2, off_span=max(1, nb // 2))
    patch['has_env'], patch['has_filter'], patch['stereo'] = cfg['has_env'], cfg['has_filter'], cfg['stereo']
    patch['wave'] = cfg['wave']
    patch['pan'] = 2. * rng.random(n) - 1.
    patch['gain'] = .2 + rng.random(n)
    stages = ([(hip.FX_FILTER,)] if cfg['has_filter'] and cfg['program'] else []) + \
        [stage_of(hip, rng, k, n, cfg['short_delay']) for k in cfg['program']]
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    ob.duty[:] = cfg['duty']
    hb = hip.Bank(ctx, n, has_env=cfg['has_env'], has_filter=cfg['has_filter'], stereo=cfg['stereo'],
                  cutoff_lfo=cfg['cutoff_lfo'], mod_target=cfg['target'], effects=[s[0] for s in stages])
    hb.load_patch(patch)
    hb.set(hip.F_DUTY, cfg['duty'])
    if stages:
        ob.set_effects(stages)
        for s, st in enumerate(stages):
            for k, p in enumerate(st[1:]):
                hb.set(hip.F_FX(s, k), np.broadcast_to(np.asarray(p, dtype=float), (n,)))
    wv = rng.integers(1, 4, n)
    if cfg['cutoff_lfo']:
        lf, ls, lo = .5 + 20. * rng.random(n), 200. + 1000. * rng.random(n), 1500. + 1000. * rng.random(n)
        lp = 6. * rng.random(n)
        ob.cutoff_lfo, ob.grid = True, grid
        ob.lfo_frequency, ob.lfo_scale, ob.lfo_offset, ob.lfo_phase = lf, ls, lo, lp.copy()
        ob.lfo_wave = wv.astype(np.int32)
        hb.set(hip.F_LFO_FREQUENCY, lf), hb.set(hip.F_LFO_SCALE, ls), hb.set(hip.F_LFO_OFFSET, lo)
        hb.set(hip.F_LFO_WAVE, wv), hb.set_state(hip.F_LFO_PHASE, lp)
    target = cfg['target']
    if target != hip.MOD_NONE:
        mf = 1. + 200. * rng.random(n)
        if target == hip.MOD_FM:
            ms, mo = 3. * rng.random(n), np.zeros(n)
        elif target == hip.MOD_FREQUENCY:
            ms, mo = 30. * rng.random(n), patch.frequency.copy()
        else:
            ms, mo = .4 * rng.random(n), np.full(n, .5)
        mp = 6. * rng.random(n)
        mw = rng.integers(1, 4, n)
        if target == hip.MOD_FM and rng.random() < .5:     # paired as FmOscillator builds them
            mw = np.minimum(cfg['wave'], 3)
        ob.mod_target = target
        ob.mod_frequency, ob.mod_scale, ob.mod_offset, ob.mod_phase = mf, ms, mo, mp.copy()
        ob.mod_wave = mw.astype(np.int32)
        hb.set(hip.F_MOD_FREQUENCY, mf), hb.set(hip.F_MOD_SCALE, ms), hb.set(hip.F_MOD_OFFSET, mo)
        hb.set(hip.F_MOD_WAVE, mw), hb.set_state(hip.F_MOD_PHASE, mp)

    want_mix, want_voices = ob.render(nb, patch.events, want_voices=True)
    note = 'draw %r' % {k: v for k, v in cfg.items() if k not in ('wave', 'duty')}
    np.testing.assert_array_equal(hb.render_voices(nb, f64=True), want_voices, err_msg=note)
    mix = np.concatenate([hb.render(k) for k in cfg['launches']])
    assert mix.shape == want_mix.shape == (nb, 2 if cfg['stereo'] else 1, B), note
    assert np.abs(mix - want_mix).max() <= 1e-6 * max(1., np.abs(want_mix).max()), note
    np.testing.assert_array_equal(hb.get_state(hip.F_PHASE), ob.phase, err_msg=note)
    if cfg['has_filter']:
        np.testing.assert_array_equal(hb.get_state(hip.F_Z0), ob.z0, err_msg=note)
        np.testing.assert_array_equal(hb.get_state(hip.F_Z1), ob.z1, err_msg=note)
    if cfg['cutoff_lfo']:
        np.testing.assert_array_equal(hb.get_state(hip.F_LFO_PHASE), ob.lfo_phase, err_msg=note)
    if target != hip.MOD_NONE:
        np.testing.assert_array_equal(hb.get_state(hip.F_MOD_PHASE), ob.mod_phase, err_msg=note)
    for s, st in enumerate(stages):
        if st[0] in (hip.FX_TREMOLO, hip.FX_DELAY):
            np.testing.assert_array_equal(hb.get_state(hip.F_FX_STATE(s, 0)), ob.fx_state[s, 0], err_msg=note)
    hb.close()


def oracle_rings(reverb):
    c = reverb.c
    rings = [np.ctypeslib.as_array(c.filters[k].data, shape=(c.filters[k].length,)).copy()
             for k in range(c.n_echos)]
    return rings, [int(c.filters[k].position) for k in range(c.n_echos)]


@pytest.mark.parametrize('seed', range(int(os.environ.get('KLANG_FUZZ_SEEDS', '30'))))
def test_random_reverb_bank_vs_oracle(hip, ctx, oracle, monkeypatch, seed):
    """Voice -> [Filter] -> Reverb -> Mixer / StereoMixer banks with drawn comb counts, pre-delays on
    either side of the 256-sample pass, the fused or the two-kernel path, voices per wavefront
    and launch cuts; float64 rings under non-sine wave functions: the delay lines must equal the
    oracle's bit for bit, the mix within 1e-6."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 5, 64, 100, 257, 300]))
    nb = int(rng.integers(4, 15))
    kw = dict(decay=float(.5 + 2.5 * rng.random()),
              pre_delay=float(rng.choice([.0001, .002, .006, .01, .03, .05])),
              dry_wet=float(rng.random()), n_echos=int(rng.choice([10, 10, 10, 7, 16])), rings='f64')
    stereo, has_env, has_filter = bool(rng.random() < .3), bool(rng.random() < .8), bool(rng.random() < .3)
    if rng.random() < .5:
        monkeypatch.setenv('KLMI_REVERB_FUSED', '0')
    vpw = str(rng.choice(['', '1', '2', '4']))
    if vpw:
        monkeypatch.setenv('KLMI_REVERB_VPW', vpw)
    patch = P.reverb_patch(n_channels=n, n_blocks=nb, seed=seed, period=int(rng.integers(3, 9)),
                           note_blocks=int(rng.integers(1, 4)))
    patch['reverb'] = kw
    patch['stereo'], patch['has_env'], patch['has_filter'] = stereo, has_env, has_filter
    patch['wave'] = rng.integers(1, 4, n).astype(np.int32)
    patch['gain'] = .5 + rng.random(n)
    patch['pan'] = 2. * rng.random(n) - 1.
    table, grid = library_table(hip)
    ob = oracle_bank_from_patch(oracle, patch, table, grid)
    _, dry = ob.render(nb, patch.events, want_voices=True)
    reverbs = [oracle.OracleReverb(kw['decay'], kw['pre_delay'], kw['dry_wet'], kw['n_echos'])
               for _ in range(n)]
    wet = np.stack([r.run(dry[v]) for v, r in enumerate(reverbs)])
    if stereo:
        pan = np.array([oracle.panning_amplitudes(p) for p in patch.pan])
        want = np.einsum('vc,vs->cs', patch.gain[:, None] * pan, wet) / n
        want = want.reshape(2, nb, B).transpose(1, 0, 2)
    else:
        want = ((patch.gain[:, None] * wet).sum(axis=0) / n).reshape(nb, 1, B)
    bank = hip.Bank(ctx, n, has_env=has_env, has_filter=has_filter, stereo=stereo, has_reverb=True,
                    reverb=kw)
    bank.load_patch(patch)
    cuts = sorted(set(int(c) for c in rng.integers(1, nb, int(rng.integers(0, 3)))))
    launches = [b - a for a, b in zip([0] + cuts, cuts + [nb])]
    mix = np.concatenate([bank.render(k) for k in launches])
    note = 'draw n=%d nb=%d %r stereo=%s env=%s filter=%s launches=%r vpw=%r' % (
        n, nb, kw, stereo, has_env, has_filter, launches, vpw)
    assert mix.shape == want.shape, note
    assert np.abs(mix - want).max() <= 1e-6 * max(1., np.abs(want).max()), note
    for v in sorted(set([0, n // 2, n - 1])):
        rings, pos = bank.reverb_state(v)
        o_rings, o_pos = oracle_rings(reverbs[v])
        assert list(pos) == o_pos, note
        for got, ref in zip(rings, o_rings):
            np.testing.assert_array_equal(got, ref, err_msg=note)
    bank.close()


@pytest.mark.parametrize('seed', range(int(os.environ.get('KLANG_FUZZ_SEEDS', '30'))))
def test_random_synthesizer_bank_vs_oracle(hip, ctx, oracle, seed):
    """PolyphonicSynthesizer banks with drawn voice counts, instrument counts (instrument
    boundaries inside wavefronts), per-voice envelopes and wave functions, optional FM, random
    chords with note-offs and launch cuts: voice streams and all routing / envelope / phase state
    equal the oracle's bit for bit."""
    rng = np.random.default_rng(5000 + seed)
    poly = int(rng.choice([1, 2, 3, 5, 8, 24]))
    n_inst = int(rng.choice([1, 7, 30, 100]))
    nb = int(rng.integers(6, 21))
    n = poly * n_inst
    attack, decay = .001 + .02 * rng.random(n), .005 + .05 * rng.random(n)
    sustain, release = .2 + .7 * rng.random(n), .005 + .05 * rng.random(n)
    wave = rng.integers(1, 4, n).astype(np.int32)
    ob = oracle.OracleBank(n, has_env=True, has_filter=False)
    ob.wave[:] = wave
    ob.set_envelopes(attack, decay, sustain, release)
    fm = bool(rng.random() < .25)
    kw = {}
    if fm:
        kw['mod_target'] = hip.MOD_FM
        ob.mod_target = oracle.MOD_FM
        ob.mod_frequency[:] = 50. + 100. * rng.random(n)
        ob.mod_scale[:] = 2. * rng.random(n)
        ob.mod_wave[:] = rng.integers(1, 4, n)
    bank = hip.Bank(ctx, n, has_env=True, poly_voices=poly, **kw)
    bank.set(hip.F_WAVE, wave)
    for field, value in ((hip.F_ATTACK, attack), (hip.F_DECAY, decay), (hip.F_SUSTAIN, sustain),
                         (hip.F_RELEASE, release)):
        bank.set(field, value)
    if fm:
        bank.set(hip.F_MOD_FREQUENCY, ob.mod_frequency)
        bank.set(hip.F_MOD_SCALE, ob.mod_scale)
        bank.set(hip.F_MOD_WAVE, ob.mod_wave)
    rows = []
    for inst in range(n_inst):
        for _ in range(int(rng.integers(0, 3 * poly + 4))):
            blk = int(rng.integers(0, nb - 1))
            pitch = int(rng.integers(36, 96))
            f = 440. * 2 ** ((pitch - 69) / 12)
            rows.append((inst, blk, pitch, 0, .2 + .8 * rng.random(), f))
            if rng.random() < .8:
                rows.append((inst, min(nb - 1, blk + int(rng.integers(0, 6))), pitch, 0, 0., f))
    notes = np.array(rows, dtype=oracle.NOTE_DTYPE) if rows else np.zeros(0, dtype=oracle.NOTE_DTYPE)
    bank.notes(notes)
    want_out, want_voices = ob.render_synth(poly, nb, notes, want_voices=True)
    note = 'draw poly=%d instruments=%d nb=%d fm=%s notes=%d' % (poly, n_inst, nb, fm, len(notes))
    np.testing.assert_array_equal(bank.render_voices(nb, f64=True), want_voices, err_msg=note)
    bank.set_mix_divisor(poly * n_inst)
    cuts = sorted(set(int(c) for c in rng.integers(1, nb, int(rng.integers(0, 3)))))
    launches = [b - a for a, b in zip([0] + cuts, cuts + [nb])]
    mix = np.concatenate([bank.render(k) for k in launches])[:, 0, :].ravel()
    assert np.abs(mix - want_out.sum(axis=0) / n_inst).max() <= 1e-6, note
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), ob.phase, err_msg=note)
    np.testing.assert_array_equal(bank.get_state(hip.F_CURRENT_PITCH), ob.current_pitch, err_msg=note)
    np.testing.assert_array_equal(bank.get_state(hip.F_ROUND_ROBIN), np.repeat(ob.round_robin, poly),
                                  err_msg=note)
    stage, value, _ = ob.env_state()
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_STAGE), stage, err_msg=note)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_VALUE), value, err_msg=note)
    if fm:
        np.testing.assert_array_equal(bank.get_state(hip.F_MOD_PHASE), ob.mod_phase, err_msg=note)
    bank.close()
