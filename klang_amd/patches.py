"""Synthetic polyphonic patches, as plain parameter arrays.

One definition shared by bench.py, the parity tests and tools/gen_golden.py so
that the reference, the oracle and the HIP bank all see the same patch.
NumPy only; no reference code is imported here.
"""
import numpy as np

BUFFER_SIZE = 256
SAMPLING_RATE = 44100

WAVE_SINE, WAVE_SQUARE, WAVE_SAWTOOTH, WAVE_TRIANGLE, WAVE_PWM, WAVE_ONE = range(6)
EV_NOTE_OFF, EV_NOTE_ON, EV_SET_CUTOFF = 0, 1, 2

EVENT_DTYPE = np.dtype([('voice', '<u4'), ('block', '<u4'), ('kind', '<u4'), ('index', '<u4'),
                        ('velocity', '<f8'), ('frequency', '<f8')])


def make_events(voice, block, kind, velocity=0., frequency=0., index=0):
    voice = np.atleast_1d(np.asarray(voice))
    ev = np.zeros(voice.shape[0], dtype=EVENT_DTYPE)
    ev['voice'] = voice
    ev['block'] = block
    ev['kind'] = kind
    ev['index'] = index
    ev['velocity'] = velocity
    ev['frequency'] = frequency
    return ev


def sort_events(events):
    """Voice-major, then block; stable (same-block events keep their order)."""
    events = np.asarray(events, dtype=EVENT_DTYPE)
    order = np.lexsort((events['block'], events['voice']))
    return np.ascontiguousarray(events[order])


class Patch(dict):
    """Parameter bundle: per-voice arrays + an event list + rendering span."""
    __getattr__ = dict.__getitem__


def polyphonic_saw_patch(n_voices=65536, seed=1234, n_blocks=512, off_base=86, off_span=172,
                         first_voice=0, count=None):
    """Config 3 / 5: saw Oscillator -> ADSR (Voice) -> Butterworth-2 Filter ->
    Mixer.  Draw order fixed by SURVEY.md §8d: u_f,u_v,u_a,u_d,u_s,u_r,u_c,u_t.
    `first_voice`/`count` select a contiguous shard of the same global patch."""
    rng = np.random.default_rng(seed)
    u = [rng.random(n_voices) for _ in range(8)]
    u_f, u_v, u_a, u_d, u_s, u_r, u_c, u_t = u
    if count is None:
        count = n_voices - first_voice
    sl = slice(first_voice, first_voice + count)
    frequency = 27.5 * 2. ** (8. * u_f[sl])
    velocity = .25 + .75 * u_v[sl]
    off_block = (off_base + np.floor(off_span * u_t[sl])).astype(np.uint32)
    local = np.arange(count, dtype=np.uint32)
    events = np.concatenate([
        make_events(local, 0, EV_NOTE_ON, velocity, frequency),
        make_events(local, off_block, EV_NOTE_OFF, 0., frequency),
    ])
    return Patch(
        name='polyphonic_saw', n_voices=count, n_total=n_voices, n_blocks=n_blocks,
        has_env=True, has_filter=True, stereo=False,
        wave=np.full(count, WAVE_SAWTOOTH, dtype=np.int32),
        frequency=frequency, velocity=velocity,
        attack=.005 + .195 * u_a[sl], decay=.05 + .45 * u_d[sl],
        sustain=.3 + .6 * u_s[sl], release=.1 + .9 * u_r[sl],
        overshoot=np.full(count, 1e-3), loop=np.zeros(count, bool), retrigger=np.zeros(count, bool),
        cutoff=200. * 40. ** u_c[sl],
        gain=np.ones(count), pan=np.zeros(count),
        events=sort_events(events),
    )


def haunting_patch(replicas=1, n_blocks=2048):
    """Config 2: the reference's examples/haunting_envelopes.py:16-92,
    VARIATION 2 (20 sine voices under looping AR envelopes into a constant-power
    StereoMixer), optionally replicated for throughput runs."""
    n = 20
    i = np.arange(n)
    frac = np.array([(k + 2) / (k + 1) for k in range(n)])          # :61 Fraction -> float
    duration = np.array([(10 * k + 15) / (k + 2) for k in range(n)])  # :62  5 + 5/frac (exact)
    frequency = 440. / frac                                          # :64
    alpha = i / (n - 1)
    velocity = (1. - alpha) * 1. + alpha * .5                        # :27-30 blend(1., .5, a)
    pan = np.where(i % 2 == 1, -1 * (i / (n - 1)), i / (n - 1))       # :33-41
    attack = .02 * duration                                          # :81
    release = (1 - .02) * duration                                   # :82
    rep = lambda a: np.tile(a, replicas)
    count = n * replicas
    local = np.arange(count, dtype=np.uint32)
    return Patch(
        name='haunting_envelopes', n_voices=count, n_total=count, n_blocks=n_blocks,
        has_env=True, has_filter=False, stereo=True,
        wave=np.full(count, WAVE_SINE, dtype=np.int32),
        frequency=rep(frequency), velocity=rep(velocity),
        attack=rep(attack), decay=np.zeros(count), sustain=np.ones(count), release=rep(release),
        overshoot=np.full(count, 1e-3), loop=np.ones(count, bool), retrigger=np.zeros(count, bool),
        cutoff=np.full(count, 440.),
        gain=np.ones(count), pan=rep(pan),
        events=sort_events(make_events(local, 0, EV_NOTE_ON, rep(velocity), rep(frequency))),
    )


REVERB_PITCHES = (60, 62, 67, 69, 71, 76, 72, 53, 55, 57)  # examples/reverberation_demo.py:22-32


def reverb_patch(n_channels=4096, seed=4321, n_blocks=512, period=86, note_blocks=11):
    """Config 4: per channel Voice(Oscillator(sine), AR(.01, 1.)) re-gated every
    `period` blocks -> its own Reverb(decay=3., dryWet=.7) -> mono Mixer."""
    rng = np.random.default_rng(seed)
    pick = np.floor(len(REVERB_PITCHES) * rng.random(n_channels)).astype(int)
    pitch = np.asarray(REVERB_PITCHES)[pick]
    frequency = 440. * 2. ** ((pitch - 69) / 12.)
    local = np.arange(n_channels, dtype=np.uint32)
    evs = []
    for start in range(0, n_blocks, period):
        evs.append(make_events(local, start, EV_NOTE_ON, 1., frequency))
        if start + note_blocks < n_blocks:
            evs.append(make_events(local, start + note_blocks, EV_NOTE_OFF, 0., frequency))
    return Patch(
        name='reverb_channels', n_voices=n_channels, n_total=n_channels, n_blocks=n_blocks,
        has_env=True, has_filter=False, stereo=False,
        wave=np.full(n_channels, WAVE_SINE, dtype=np.int32),
        frequency=frequency, velocity=np.ones(n_channels),
        attack=np.full(n_channels, .01), decay=np.zeros(n_channels),
        sustain=np.ones(n_channels), release=np.ones(n_channels),
        overshoot=np.full(n_channels, 1e-3), loop=np.zeros(n_channels, bool),
        retrigger=np.zeros(n_channels, bool),
        cutoff=np.full(n_channels, 440.),
        gain=np.ones(n_channels), pan=np.zeros(n_channels),
        reverb=dict(decay=3., pre_delay=.03, dry_wet=.7, n_echos=10),
        events=sort_events(np.concatenate(evs)),
    )
