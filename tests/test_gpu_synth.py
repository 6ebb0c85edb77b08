"""HIP path vs the reference's fixtures and vs the oracle for synthesizer banks (SURVEY 8f
rank 3): PolyphonicSynthesizer note routing on the device (round robin, note-off by pitch)
and the activity gating of Mono-/PolyphonicSynthesizer (synthesizer.py:151-219).
Everything goes through the C-ABI."""
import numpy as np
import pytest

from tests.helpers import B, CHIRPS, oracle_synth_bank, report, synth_notes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def synth_bank(hip, ctx, n_voices, template, wave, **kw):
    bank = hip.Bank(ctx, n_voices, has_env=True, **kw)
    bank.set(hip.F_WAVE, [wave] * n_voices)
    for field, value in zip((hip.F_ATTACK, hip.F_DECAY, hip.F_SUSTAIN, hip.F_RELEASE), template):
        bank.set(field, [float(value)] * n_voices)
    return bank


def hip_notes(hip, rows, instrument=0):
    rows = np.asarray(rows, dtype=float).reshape(-1, 4)
    nt = np.zeros(len(rows), dtype=hip.NOTE_DTYPE)
    nt['instrument'] = instrument
    nt['block'], nt['pitch'] = rows[:, 0].astype(int), rows[:, 1].astype(int)
    nt['velocity'], nt['frequency'] = rows[:, 2], rows[:, 3]
    return nt


@pytest.mark.parametrize('key,poly', [('poly24', 24), ('poly3', 3)])
def test_polyphonic_synthesizer_vs_reference(hip, ctx, golden, key, poly):
    g = golden('synthesizer')
    ref = g[key + '_out']
    n_blocks = len(ref)
    bank = synth_bank(hip, ctx, poly, g['template'], hip.WAVE_SAWTOOTH, poly_voices=poly)
    bank.notes(hip_notes(hip, g[key + '_notes']))
    # preview: float64 voice streams (zeros where a voice is idle); summed in voice order
    # and divided like synthesizer.py:213-219 they ARE the reference's samples
    voices = bank.render_voices(n_blocks, f64=True)
    out = np.zeros(n_blocks * B)
    for v in voices:
        out += v
    np.testing.assert_array_equal((out / poly).reshape(-1, B), ref)
    # the mix bus (float32), rendered in three calls: routing state carries over
    mix = np.concatenate([bank.render(k) for k in (5, 2, n_blocks - 7)])[:, 0, :]
    assert report('PolyphonicSynthesizer', np.abs(mix - ref).max()) <= 1e-6
    state = g[key + '_state']   # currentPhase, currentPitch, amplitude, active, frequency
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), state[:, 0])
    np.testing.assert_array_equal(bank.get_state(hip.F_CURRENT_PITCH), state[:, 1])
    np.testing.assert_array_equal(bank.get_state(hip.F_AMPLITUDE), state[:, 2])
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_STAGE) != 0, state[:, 3] != 0)
    np.testing.assert_array_equal(bank.get_state(hip.F_FREQUENCY), state[:, 4])
    n_on = int((g[key + '_notes'][:, 2] > 0).sum())
    np.testing.assert_array_equal(bank.get_state(hip.F_ROUND_ROBIN), n_on % poly)
    bank.close()


@pytest.mark.parametrize('policy', ['newest', 'oldest', 'lowest', 'highest'])
def test_monophonic_synthesizer_vs_reference(hip, ctx, golden, policy):
    """A gated one-voice bank fed with the notes the reference's NoteScheduler handed to the
    voice (klang_amd.audio.synthesizer.NoteScheduler reproduces them, test_host_blocks)."""
    g = golden('synthesizer')
    ref = g['mono_%s_out' % policy]
    bank = synth_bank(hip, ctx, 1, g['template'], hip.WAVE_SAWTOOTH, gated=True)
    for blk, pitch, velocity, frequency in g['mono_%s_handed' % policy]:
        bank.event(0, int(blk), hip.EV_NOTE_ON if velocity > 0 else hip.EV_NOTE_OFF, velocity,
                   frequency)
    got = bank.render_voices(len(ref), f64=True)[0].reshape(-1, B)
    np.testing.assert_array_equal(got, ref)
    mix = bank.render(len(ref))[:, 0, :]
    assert report('MonophonicSynthesizer', np.abs(mix - ref).max()) <= 1e-6
    assert bank.get_state(hip.F_PHASE)[0] == g['mono_%s_state' % policy][0]
    bank.close()


def random_notes(rng, dtype, n_inst, n_blocks, per_inst):
    """Random chords with note-offs for most of them, some in the same buffer."""
    rows = []
    for inst in range(n_inst):
        for _ in range(per_inst):
            blk = int(rng.integers(0, n_blocks - 2))
            pitch = int(rng.integers(36, 96))
            rows.append((inst, blk, pitch, 0, .2 + .8 * rng.random(), 440. * 2 ** ((pitch - 69) / 12)))
            if rng.random() < .8:
                rows.append((inst, min(n_blocks - 1, blk + int(rng.integers(0, 6))), pitch, 0, 0.,
                             440. * 2 ** ((pitch - 69) / 12)))
    return np.array(rows, dtype=dtype)


@pytest.mark.parametrize('poly,wave,mod', [(24, 'sawtooth', False), (5, 'square', False),
                                           (24, 'triangle', True)])
def test_many_instruments_vs_oracle(hip, ctx, oracle, poly, wave, mod):
    """100 PolyphonicSynthesizers (instrument boundaries fall inside wavefronts), random notes,
    two launches: voices bit-exact vs the oracle, all routing / phase / envelope state equal."""
    n_inst, n_blocks = 100, 24
    n = n_inst * poly
    rng = np.random.default_rng(77)
    template = (.003, .015, .5, .02)
    obank = oracle_synth_bank(oracle, n, template, wave)
    kw = {}
    if mod:
        kw['mod_target'] = hip.MOD_FM
        obank.mod_target = oracle.MOD_FM
        obank.mod_frequency[:] = 50. + 100. * rng.random(n)
        obank.mod_scale[:] = 2. * rng.random(n)
        obank.mod_wave[:] = oracle.WAVE_IDS['square']
    bank = synth_bank(hip, ctx, n, template, oracle.WAVE_IDS[wave], poly_voices=poly, **kw)
    if mod:
        bank.set(hip.F_MOD_FREQUENCY, obank.mod_frequency)
        bank.set(hip.F_MOD_SCALE, obank.mod_scale)
        bank.set(hip.F_MOD_WAVE, obank.mod_wave)
    notes = random_notes(rng, oracle.NOTE_DTYPE, n_inst, n_blocks, 40)
    bank.notes(notes)
    want_out, want_voices = obank.render_synth(poly, n_blocks, notes, want_voices=True)
    np.testing.assert_array_equal(bank.render_voices(n_blocks, f64=True), want_voices)
    # instruments mixed: sum over instruments of (sum of voices / poly), / instruments
    bank.set_mix_divisor(poly * n_inst)
    mix = np.concatenate([bank.render(9), bank.render(n_blocks - 9)])[:, 0, :].ravel()
    assert np.abs(mix - want_out.sum(axis=0) / n_inst).max() <= 1e-6
    np.testing.assert_array_equal(bank.get_state(hip.F_PHASE), obank.phase)
    np.testing.assert_array_equal(bank.get_state(hip.F_CURRENT_PITCH), obank.current_pitch)
    np.testing.assert_array_equal(bank.get_state(hip.F_ROUND_ROBIN),
                                  np.repeat(obank.round_robin, poly))
    stage, value, _ = obank.env_state()
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_STAGE), stage)
    np.testing.assert_array_equal(bank.get_state(hip.F_ENV_VALUE), value)
    if mod:
        np.testing.assert_array_equal(bank.get_state(hip.F_MOD_PHASE), obank.mod_phase)
    bank.close()


def test_gated_bank_freezes_idle_oscillators(hip, ctx):
    """An idle voice of a gated bank keeps its phase; the same voice in a plain bank runs on."""
    for gated in (False, True):
        bank = synth_bank(hip, ctx, 2, (.001, .001, .5, .001), hip.WAVE_SINE, gated=gated)
        bank.set(hip.F_FREQUENCY, [100., 100.])
        bank.event(1, 0, hip.EV_NOTE_ON, 1., 100.)
        bank.render(3)
        p = bank.get_state(hip.F_PHASE)
        assert p[1] != 0.
        assert (p[0] == 0.) == gated
        bank.close()


def test_argument_errors(hip, ctx):
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 25, has_env=True, poly_voices=24)          # not whole instruments
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 24, has_env=True, has_filter=True, poly_voices=24)
    with pytest.raises(hip.NativeError):
        hip.Bank(ctx, 24, has_env=False, gated=True)
    plain = hip.Bank(ctx, 4, has_env=True)
    with pytest.raises(hip.NativeError):
        plain.note(0, 0, 60, 1., 261.6)
    with pytest.raises(hip.NativeError):
        plain.get_state(hip.F_CURRENT_PITCH)
    plain.close()
    bank = hip.Bank(ctx, 48, has_env=True, poly_voices=24)
    for bad in ((2, 0, 60, 1., 261.6), (0, 0, 128, 1., 261.6), (0, 0, 60, 1.5, 261.6),
                (0, 0, 60, 1., -1.)):
        with pytest.raises(hip.NativeError):
            bank.note(*bad)
    bank.set_state(hip.F_ROUND_ROBIN, [5.], first=30)     # the whole second instrument
    rr = bank.get_state(hip.F_ROUND_ROBIN)
    assert (rr[:24] == 0).all() and (rr[24:] == 5).all()
    with pytest.raises(hip.NativeError):
        bank.set_state(hip.F_ROUND_ROBIN, [24.])
    bank.close()


# ---- Kick (synthesizer.py:253-283) -----------------------------------------------------------
def test_kick_batch_vs_reference(hip, ctx, golden):
    """Two kicks in one batch, three render calls; float64 streams vs the real reference
    (np.exp / np.sin vs the device's: ulp-level differences fed back through the phase)."""
    g = golden('synthesizer')
    batch = hip.KickBatch(ctx, 2)
    for field, value in ((hip.KICK_FREQUENCY, 110.), (hip.KICK_DECAY, .25),
                         (hip.KICK_INTENSITY, 3.5), (hip.KICK_PITCH_DECAY, .08)):
        batch.set(field, [value], first=1)
    batch.trigger([0, 0, 1], [0, 70, 3])
    out = np.concatenate([batch.render(k) for k in (1, 30, 69)], axis=1)
    for i, name in enumerate(('kick_default', 'kick_tom')):
        want = g[name + '_out'].ravel()
        assert report('Kick (float64 out)', np.abs(out[i] - want).max()) <= 1e-11, name
    time, phase = batch.state()
    for i, name in enumerate(('kick_default', 'kick_tom')):
        assert abs(time[i] - g[name + '_final'][0]) <= 1e-12
        assert np.abs(phase[i] - g[name + '_final'][1:]).max() <= 1e-9
    with pytest.raises(hip.NativeError):
        batch.trigger([2], [0])
    with pytest.raises(hip.NativeError):
        batch.set(hip.KICK_DECAY, [0.])
    batch.close()


def test_many_kicks_vs_oracle(hip, ctx, oracle):
    rng = np.random.default_rng(3)
    n, nb = 300, 20
    batch = hip.KickBatch(ctx, n)
    par = np.stack([30. + 200. * rng.random(n), .05 + rng.random(n), 4. * rng.random(n),
                    .02 + .5 * rng.random(n)])
    for field in range(4):
        batch.set(field, par[field])
    hits = rng.random((n, nb)) < .15
    which, block = np.nonzero(hits)
    batch.trigger(which, block)
    got = batch.render(nb)
    for i in range(0, n, 7):
        kick = oracle.OracleKick(*par[:, i])
        want = np.concatenate([kick.block(hits[i, b]) for b in range(nb)])
        assert np.abs(got[i] - want).max() <= 1e-11
    batch.close()


# ---- Chirper (oscillators.py:280-308) --------------------------------------------------------
def test_chirp_batch_vs_reference(hip, ctx, golden):
    """All ten fixtures as ONE batch, rendered in four calls; every one runs past its tEnd
    into the held region (tolerances: tests/helpers.py CHIRPS)."""
    g = golden('chirper')
    waves = {'sine': hip.WAVE_SINE, 'sawtooth': hip.WAVE_SAWTOOTH, 'square': hip.WAVE_SQUARE,
             'triangle': hip.WAVE_TRIANGLE}
    batch = hip.ChirpBatch(ctx, [c[1][0] for c in CHIRPS], [c[1][1] for c in CHIRPS],
                           [c[1][2] for c in CHIRPS], [c[1][3] for c in CHIRPS],
                           [waves[c[2]] for c in CHIRPS])
    out = np.concatenate([batch.render(k) for k in (1, 9, 30, 60)], axis=1)
    t = batch.time_vectors()
    for i, (name, _, _, tol) in enumerate(CHIRPS):
        want = g[name].ravel()
        report('Chirper %s (float64 out)' % name, np.abs(out[i] - want).max())
        if tol:
            assert np.abs(out[i] - want).max() <= tol, name
        else:
            np.testing.assert_array_equal(out[i], want)
        np.testing.assert_array_equal(t[i], g[name + '_t'])
    batch.reset(1, 1)
    np.testing.assert_array_equal(batch.render(1)[1], g['linear_saw'][0])
    batch.close()
    with pytest.raises(hip.NativeError) as e:
        hip.ChirpBatch(ctx, [100.], [1.], [-100.], 'logarithmic')
    assert 'same sign' in str(e.value)
    with pytest.raises(hip.NativeError):
        hip.ChirpBatch(ctx, [0.], [1.], [100.], 'hyperbolic')
    hip.ChirpBatch(ctx, [100.], [0.], [-100.], 'logarithmic').close()   # never sweeps: no error
    with pytest.raises(ValueError):
        hip.ChirpBatch(ctx, [10.], [1.], [100.], 'cubic')


def test_chirper_and_kick_through_the_api(hip):
    """A Chirper and a Kick into one Mixer: both blocks run on the device, the mixer takes
    their outputs as external voices; equal to the host evaluation."""
    from klang_amd.audio import Chirper, Dac, Kick, Mixer, sawtooth
    from klang_amd.klang import run_klang
    from klang_amd.messages import Note

    def build():
        kick = Kick(frequency=60., decay=.3)
        kick.input.push(Note(36))
        mixer = Mixer()
        mixer += Chirper(200., .2, 3000., wave_func=sawtooth)
        mixer += kick
        return mixer | Dac()
    dur = 30 * B / 44100. + 1e-9
    got = np.concatenate(run_klang(build(), offline=True, duration=dur))
    want = np.concatenate(run_klang(build(), offline=True, duration=dur, backend='host'))
    assert got.shape == want.shape and np.abs(want).max() > .3
    assert np.abs(got - want).max() <= 1e-6


# ---- HiHat (synthesizer.py:222-250): envelope on the device, noise from the host RNG --------------
@pytest.mark.parametrize('name,seed,kw', [('hihat_fresh', 1234, {}),
                                          ('hihat_looped', 99, dict(decay=.12, loopedNoise=True))])
def test_hihat_on_the_device_vs_reference(hip, golden, name, seed, kw):
    """The reference's HiHat with NumPy's global RNG seeded as the fixture was: the device
    block draws the same noise in the same order and multiplies it with the decay envelope
    (float32 mix output vs the reference's float64 samples)."""
    from klang_amd import fusion
    from klang_amd.audio import Dac, HiHat
    from klang_amd.messages import Note
    from klang_amd.schedule import determine_execution_order
    want = golden('synthesizer')[name + '_out']
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        hat = HiHat(**kw)
        order, banks = fusion.compile_order(determine_execution_order([hat | Dac()]))
        assert [type(b).__name__ for b in order] == ['GpuHiHat', 'Dac']
        out = []
        for b in range(len(want)):
            if b in (2, 30):
                hat.input.push(Note(42))
            if b == 35:
                hat.input.push(Note(42, 0.))
            for blk in order:
                blk.update()
            out.append(np.array(hat.output.value))
    finally:
        np.random.set_state(state)
    err = np.abs(np.array(out) - want).max()
    report('HiHat (device envelope x host noise)', err)
    assert err <= 1e-6 and np.abs(want).max() > .5
    banks[0].sync_host_blocks()
    assert hat.envelope.stage == 0          # decayed to OFF by the end of the fixture
    banks[0].close()
