"""HIP path vs WAV files written by the real reference and vs the oracle for the output side
of an offline bounce: pack_signals, fade-out, peak-normalised int16
(klanggeber.py:81-97, 205-249, wavfile.py:54-63).  Integer output: bit-exact."""
import wave

import numpy as np
import pytest

from klang_amd import patches as P
from tests.helpers import B, bounce_signals, hip_bank_from_patch, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx(hip):
    c = hip.Context(0)
    yield c
    c.close()


def external_bank(hip, ctx, sig):
    """A bank whose mix bus is exactly float32(sig): one external voice per channel, panned
    hard left / right with the linear law (gains exactly 1 and 0), divisor 1."""
    n_blocks, ch, _ = sig.shape
    bank = hip.Bank(ctx, ch, has_env=False, stereo=ch == 2, pan_mode='linear')
    bank.set(hip.F_WAVE, [hip.WAVE_EXTERNAL] * ch)
    if ch == 2:
        bank.set(hip.F_PAN, [-1., 1.])
    bank.set_mix_divisor(1.)
    bank.set_input(sig.transpose(1, 0, 2).reshape(ch, -1))
    return bank


@pytest.mark.parametrize('name,ch', [('mono', 1), ('stereo', 2), ('nofade', 2), ('allfade', 1)])
def test_bounce_vs_reference_wav(hip, ctx, golden, name, ch):
    g = golden('bounce')
    duration, fadeout = g[name + '_params']
    want = g[name + '_pcm']
    n_blocks = len(want) // B
    sig = bounce_signals(name, n_blocks, ch)
    assert sig.sum() == g[name + '_signal_sum'][0]
    bank = external_bank(hip, ctx, sig)
    frames, pcm, peak = bank.bounce(duration, fadeout)
    assert frames.shape == (n_blocks * B, ch)
    report('bounce: int16 samples (steps)', np.abs(pcm.astype(int) - want.astype(int)).max())
    np.testing.assert_array_equal(pcm, want)
    assert peak == np.abs(frames).max()
    bank.close()


@pytest.mark.parametrize('stereo', [False, True])
def test_bounce_of_a_rendered_patch_vs_oracle(hip, ctx, oracle, stereo):
    """The same patch rendered (mix bus) and bounced: frames and int16 samples equal the
    oracle's restatement applied to the bus, bit for bit; state advances alike."""
    duration, fadeout = 40.5 * B / 44100., .12
    if stereo:
        patch = P.haunting_patch(replicas=4, n_blocks=40)
    else:
        patch = P.polyphonic_saw_patch(n_voices=700, n_blocks=40, off_base=8, off_span=20)
    a = hip_bank_from_patch(hip, ctx, patch)
    b = hip_bank_from_patch(hip, ctx, patch)
    bus = a.render(40)
    want_frames, want_pcm, want_peak = oracle.bounce(bus, duration, fadeout)
    frames, pcm, peak = b.bounce(duration, fadeout)
    np.testing.assert_array_equal(frames, want_frames)
    np.testing.assert_array_equal(pcm, want_pcm)
    assert peak == want_peak and np.abs(pcm).max() == 32767
    assert np.abs(frames[-B:]).max() < .05 * peak          # faded out
    np.testing.assert_array_equal(a.get_state(hip.F_PHASE), b.get_state(hip.F_PHASE))
    np.testing.assert_array_equal(a.render(2), b.render(2))
    a.close()
    b.close()


def test_bounce_errors(hip, ctx):
    bank = hip.Bank(ctx, 4, has_env=True)      # nobody plays: silence
    with pytest.raises(hip.NativeError) as e:
        bank.bounce(.05, 0.)
    assert e.value.code == hip.E_ARG and 'Zero amplitude' in str(e.value)
    frames, pcm, peak = bank.bounce(.05, 0., want_pcm=False)
    assert pcm is None and peak == 0. and not frames.any()
    for bad in (float('inf'), 0., -1., 100 / 44100.):
        with pytest.raises(hip.NativeError):
            bank.bounce(bad, 0.)
    with pytest.raises(hip.NativeError):
        bank.bounce(.05, -1.)
    assert hip.load().klmi_bounce_cycles(1.0) == 172
    bank.close()


def test_run_klang_writes_the_wav_from_the_device(hip, tmp_path):
    """run_klang(filepath=...) with [bank, Dac]: one device pass; the WAV equals the host
    evaluation's within one int16 step (float32 mix rounding), the frames within 1e-6."""
    from klang_amd.audio import ADSR, Dac, Filter, Oscillator, StereoMixer, Voice, sawtooth
    from klang_amd.klang import run_klang
    from klang_amd.messages import Note

    def build():
        mixer = StereoMixer()
        for v in range(6):
            voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .05, .6, .1))
            fil = Filter(frequency=800. + 300. * v)
            voice | fil
            mixer += fil
            mixer.pannings[-1] = -1. + .4 * v
            voice.input.push(Note(50 + 3 * v, .5 + .08 * v))
        return mixer | Dac(nChannels=2)
    out = {}
    for backend in ('hip', 'host'):
        path = str(tmp_path / (backend + '.wav'))
        frames = run_klang(build(), backend=backend, filepath=path, offline=True, duration=.3,
                           fadeout=.1)
        with wave.open(path, 'rb') as wf:
            pcm = np.frombuffer(wf.readframes(-1), dtype='<i2').reshape(-1, 2).astype(int)
        out[backend] = (np.concatenate(frames), pcm)
    assert out['hip'][0].shape == out['host'][0].shape == (51 * B, 2)
    assert np.abs(out['hip'][0] - out['host'][0]).max() <= 1e-6
    assert np.abs(out['hip'][1] - out['host'][1]).max() <= 1
    assert np.abs(out['hip'][1]).max() == 32767
