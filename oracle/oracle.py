"""ctypes front-end to oracle/libklang_oracle.so.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module (see klang_oracle.h).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libklang_oracle.so')

BUFFER_SIZE = 256
SAMPLING_RATE = 44100
DT = 1. / SAMPLING_RATE
GRID_POINTS = 1000

WAVE_IDS = {'sine': 0, 'square': 1, 'sawtooth': 2, 'triangle': 3, 'pwm': 4, 'one': 5}
RING_KINDS = {'forward': 0, 'backward': 1, 'echo': 2}

_dp = C.POINTER(C.c_double)


class Envelope(C.Structure):
    """Mirror of ko_envelope."""
    _fields_ = [
        ('attack', C.c_double), ('decay', C.c_double), ('sustain', C.c_double),
        ('release', C.c_double), ('dt', C.c_double), ('overshoot', C.c_double),
        ('retrigger', C.c_int), ('loop', C.c_int), ('stage', C.c_int),
        ('value', C.c_double), ('enabled', C.c_int),
        ('attackCoef', C.c_double), ('attackBase', C.c_double),
        ('decayCoef', C.c_double), ('decayBase', C.c_double),
        ('releaseCoef', C.c_double), ('releaseBase', C.c_double),
    ]


class RingFilter(C.Structure):
    _fields_ = [('length', C.c_size_t), ('position', C.c_size_t),
                ('alpha', C.c_double), ('data', _dp)]


class Reverb(C.Structure):
    _fields_ = [('n_echos', C.c_int), ('dry_wet', C.c_double),
                ('filters', C.POINTER(RingFilter))]


class Event(C.Structure):
    _fields_ = [('voice', C.c_uint32), ('block', C.c_uint32), ('kind', C.c_uint32),
                ('index', C.c_uint32), ('velocity', C.c_double), ('frequency', C.c_double)]


EVENT_DTYPE = np.dtype([('voice', '<u4'), ('block', '<u4'), ('kind', '<u4'), ('index', '<u4'),
                        ('velocity', '<f8'), ('frequency', '<f8')])


NOTE_DTYPE = np.dtype([('instrument', '<u4'), ('block', '<u4'), ('pitch', '<i4'),
                       ('reserved', '<u4'), ('velocity', '<f8'), ('frequency', '<f8')])


class Chirper(C.Structure):
    _fields_ = [('f0', C.c_double), ('t1', C.c_double), ('f1', C.c_double), ('method', C.c_int),
                ('wave', C.c_int), ('t', C.c_double * BUFFER_SIZE)]


class Kick(C.Structure):
    _fields_ = [('frequency', C.c_double), ('decay', C.c_double), ('intensity', C.c_double),
                ('pitch_decay', C.c_double), ('current_time', C.c_double),
                ('current_phase', C.c_double * BUFFER_SIZE)]


class Bank(C.Structure):
    _fields_ = [
        ('n_voices', C.c_int), ('has_env', C.c_int), ('has_filter', C.c_int), ('stereo', C.c_int),
        ('frequency', _dp), ('amplitude', _dp), ('duty', _dp),
        ('wave', C.POINTER(C.c_int32)),
        ('env', C.POINTER(Envelope)),
        ('coef_index', C.POINTER(C.c_int32)),
        ('coef_table', _dp),
        ('gain', _dp), ('pan_l', _dp), ('pan_r', _dp),
        ('phase', _dp), ('z0', _dp), ('z1', _dp),
        ('cutoff_lfo', C.c_int),
        ('lfo_frequency', _dp), ('lfo_scale', _dp), ('lfo_offset', _dp), ('lfo_phase', _dp),
        ('lfo_wave', C.POINTER(C.c_int32)),
        ('grid', _dp),
        ('mod_target', C.c_int),
        ('mod_frequency', _dp), ('mod_scale', _dp), ('mod_offset', _dp), ('mod_phase', _dp),
        ('mod_wave', C.POINTER(C.c_int32)),
        ('n_fx', C.c_int), ('fx_kind', C.c_int * 8),
        ('fx_par', _dp), ('fx_state', _dp), ('fx_ring', _dp), ('fx_ring_capacity', C.c_int),
    ]


FX_NONE, FX_FILTER, FX_GAIN, FX_TRANSFORM, FX_TANH, FX_OCTAVE, FX_TREMOLO, FX_DELAY, \
    FX_SUBSAMPLE, FX_BITCRUSH, FX_RINGMOD = range(11)
FX_PARAMS = 8


MOD_NONE, MOD_FM, MOD_FREQUENCY, MOD_DUTY = range(4)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('oracle library missing: run `make -C oracle` (%s)' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ko_np_mod.restype = C.c_double
        L.ko_np_mod.argtypes = [C.c_double, C.c_double]
        L.ko_sample_phase_const.argtypes = [C.c_double, C.c_double, _dp, _dp]
        L.ko_sample_phase_array.argtypes = [_dp, C.c_double, _dp, _dp]
        L.ko_phasor_step.restype = C.c_double
        L.ko_phasor_step.argtypes = [C.c_double, C.c_double]
        L.ko_wrap.restype = C.c_double
        L.ko_wrap.argtypes = [C.c_double]
        L.ko_wave.argtypes = [C.c_int, _dp, C.c_size_t, C.c_double, _dp]
        L.ko_env_coefficient.restype = C.c_double
        L.ko_env_coefficient.argtypes = [C.c_double, C.c_double]
        L.ko_envelope_init.argtypes = [C.POINTER(Envelope)] + [C.c_double] * 6 + [C.c_int, C.c_int]
        L.ko_envelope_recompute.argtypes = [C.POINTER(Envelope)]
        L.ko_envelope_gate.argtypes = [C.POINTER(Envelope), C.c_int]
        L.ko_envelope_sample.argtypes = [C.POINTER(Envelope), C.c_size_t, _dp]
        L.ko_butter2_lowpass.argtypes = [C.c_double, _dp, _dp]
        L.ko_filter_grid.argtypes = [_dp]
        L.ko_filter_grid_index.restype = C.c_int
        L.ko_filter_grid_index.argtypes = [_dp, C.c_int, C.c_double]
        L.ko_lfilter2.argtypes = [_dp, _dp, _dp, C.c_size_t, _dp, _dp]
        L.ko_lfilter1.argtypes = [_dp, _dp, _dp, C.c_size_t, _dp, _dp]
        L.ko_low_pass_coefficients.argtypes = [C.c_double, _dp, _dp]
        L.ko_ring_filter_init.restype = C.c_int
        L.ko_ring_filter_init.argtypes = [C.POINTER(RingFilter), C.c_size_t, C.c_double]
        L.ko_ring_filter_free.argtypes = [C.POINTER(RingFilter)]
        L.ko_ring_filter_run.argtypes = [C.POINTER(RingFilter), C.c_int, _dp, C.c_size_t, _dp]
        L.ko_next_primes.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.ko_reverb_alpha.restype = C.c_double
        L.ko_reverb_alpha.argtypes = [C.c_double, C.c_double]
        L.ko_reverb_init.restype = C.c_int
        L.ko_reverb_init.argtypes = [C.POINTER(Reverb), C.c_double, C.c_double, C.c_double, C.c_int]
        L.ko_reverb_free.argtypes = [C.POINTER(Reverb)]
        L.ko_reverb_run.argtypes = [C.POINTER(Reverb), _dp, C.c_size_t, _dp]
        L.ko_mixer.argtypes = [_dp, _dp, C.c_int, C.c_size_t, _dp]
        L.ko_panning_amplitudes.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, _dp]
        L.ko_stereo_mixer.argtypes = [_dp, _dp, _dp, C.c_int, C.c_size_t, _dp]
        L.ko_bank_render.argtypes = [C.POINTER(Bank), C.c_int, C.POINTER(Event), C.c_size_t,
                                     _dp, _dp, C.c_int]
        L.ko_synth_render.argtypes = [C.POINTER(Bank), C.c_int, C.c_int, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_int, C.c_void_p, C.c_size_t,
                                      _dp, _dp]
        L.ko_bounce.restype = C.c_float
        L.ko_bounce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                C.c_void_p, C.c_void_p]
        L.ko_kick_block.argtypes = [C.POINTER(Kick), C.c_int, _dp]
        L.ko_chirp_reset.argtypes = [C.POINTER(Chirper)]
        L.ko_chirp_block.argtypes = [C.POINTER(Chirper), _dp]
        L.ko_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- oscillators / waves ---------------------------------------------------
def np_mod(a, b):
    return lib().ko_np_mod(a, b)


def sample_phase(frequency, start=0.):
    phase = np.empty(BUFFER_SIZE)
    nxt = C.c_double()
    if np.ndim(frequency) == 0:
        lib().ko_sample_phase_const(float(frequency), float(start), _p(phase), C.byref(nxt))
    else:
        f = _f64(frequency)
        assert f.shape == (BUFFER_SIZE,)
        lib().ko_sample_phase_array(_p(f), float(start), _p(phase), C.byref(nxt))
    return phase, nxt.value


def phasor_step(frequency, current):
    return lib().ko_phasor_step(float(frequency), float(current))


def wave(name, phase, duty=.5):
    phase = _f64(phase)
    out = np.empty_like(phase)
    lib().ko_wave(WAVE_IDS[name], _p(phase), phase.size, float(duty), _p(out))
    return out


class OscillatorState:
    """oscillators.py:111-140 restated over the oracle primitives."""

    def __init__(self, frequency=440., wave_name='sine', start_phase=0., duty=.5):
        self.frequency = frequency
        self.wave_name = wave_name
        self.phase = start_phase
        self.duty = duty

    def block(self):
        phase, self.phase = sample_phase(self.frequency, self.phase)
        return wave(self.wave_name, phase, self.duty)


# ---- envelope --------------------------------------------------------------
class OracleEnvelope:
    def __init__(self, attack, decay, sustain, release, dt=DT, overshoot=1e-3,
                 retrigger=False, loop=False):
        self.c = Envelope()
        lib().ko_envelope_init(C.byref(self.c), attack, decay, sustain, release, dt, overshoot,
                               int(retrigger), int(loop))

    def gate(self, on):
        lib().ko_envelope_gate(C.byref(self.c), int(bool(on)))

    def sample(self, n):
        out = np.empty(int(n))
        lib().ko_envelope_sample(C.byref(self.c), int(n), _p(out))
        return out

    @property
    def state(self):
        return (self.c.stage, self.c.value, self.c.enabled)

    @property
    def active(self):
        return self.c.stage != 0


# ---- filter ---------------------------------------------------------------
def butter2_lowpass(wn):
    b = np.empty(3)
    a = np.empty(3)
    lib().ko_butter2_lowpass(float(wn), _p(b), _p(a))
    return b, a


def filter_grid():
    g = np.empty(GRID_POINTS)
    lib().ko_filter_grid(_p(g))
    return g


def filter_grid_index(grid, frequency):
    grid = _f64(grid)
    return lib().ko_filter_grid_index(_p(grid), grid.size, float(frequency))


def coefficient_table(grid=None):
    """[1000][5] = b0 b1 b2 a1 a2 from the closed-form design on the grid."""
    if grid is None:
        grid = filter_grid()
    table = np.empty((len(grid), 5))
    for i, f in enumerate(grid):
        b, a = butter2_lowpass(f / (SAMPLING_RATE // 2))
        table[i] = (b[0], b[1], b[2], a[1], a[2])
    return table


def lfilter2(b, a, x, z):
    b, a, x = _f64(b), _f64(a), _f64(x)
    z = _f64(z).copy()
    y = np.empty_like(x)
    lib().ko_lfilter2(_p(b), _p(a), _p(x), x.size, _p(z), _p(y))
    return y, z


def lfilter1(b, a, x, z):
    b, a, x = _f64(b), _f64(a), _f64(x)
    z = _f64(z).copy()
    y = np.empty_like(x)
    lib().ko_lfilter1(_p(b), _p(a), _p(x), x.size, _p(z), _p(y))
    return y, z


def low_pass_coefficients(frequency):
    b = np.empty(1)
    a = np.empty(2)
    lib().ko_low_pass_coefficients(float(frequency), _p(b), _p(a))
    return b, a


# ---- ring filters / reverb -------------------------------------------------
class OracleRingFilter:
    def __init__(self, kind, length, alpha=.9):
        self.kind = RING_KINDS[kind]
        self.c = RingFilter()
        if lib().ko_ring_filter_init(C.byref(self.c), int(length), float(alpha)) != 0:
            raise RuntimeError('ring filter init failed')

    def filter(self, x):
        x = _f64(x)
        y = np.empty_like(x)
        lib().ko_ring_filter_run(C.byref(self.c), self.kind, _p(x), x.size, _p(y))
        return y

    def __del__(self):
        if _lib is not None:
            _lib.ko_ring_filter_free(C.byref(self.c))


def next_primes(n, start):
    out = (C.c_int * n)()
    lib().ko_next_primes(n, int(start), out)
    return list(out)


def reverb_alpha(delay_time, decay):
    return lib().ko_reverb_alpha(float(delay_time), float(decay))


class OracleReverb:
    def __init__(self, decay=1.5, pre_delay=.03, dry_wet=.7, n_echos=10):
        self.c = Reverb()
        if lib().ko_reverb_init(C.byref(self.c), decay, pre_delay, dry_wet, n_echos) != 0:
            raise RuntimeError('reverb init failed')

    def run(self, x):
        x = _f64(x)
        out = np.empty_like(x)
        lib().ko_reverb_run(C.byref(self.c), _p(x), x.size, _p(out))
        return out

    def __del__(self):
        if _lib is not None:
            _lib.ko_reverb_free(C.byref(self.c))


# ---- mixers ----------------------------------------------------------------
def mixer(inputs, gains):
    inputs = _f64(inputs)
    n, length = inputs.shape
    gains = _f64(gains)
    out = np.empty(length)
    lib().ko_mixer(_p(inputs), _p(gains), n, length, _p(out))
    return out


PAN_MODES = {'linear': 0, 'constant_power': 1, 'mixture': 2}


def panning_amplitudes(level, mode='constant_power', pan_law=None):
    lr = np.empty(2)
    lib().ko_panning_amplitudes(float(level), PAN_MODES[mode], int(pan_law is not None),
                                float(pan_law or 0.), _p(lr))
    return lr


def stereo_mixer(inputs, gains, pannings, mode='constant_power', pan_law=None):
    inputs = _f64(inputs)
    n, length = inputs.shape
    gains = _f64(gains)
    pan = _f64([panning_amplitudes(p, mode, pan_law) for p in pannings]).reshape(n, 2)
    out = np.empty((2, length))
    lib().ko_stereo_mixer(_p(inputs), _p(gains), _p(pan), n, length, _p(out))
    return out


# ---- bank -----------------------------------------------------------------
class OracleBank:
    """A bank of independent Voice(osc, env) [-> Filter] chains into a Mixer /
    StereoMixer, the oracle twin of klang_amd's device bank."""

    def __init__(self, n_voices, has_env=True, has_filter=True, stereo=False):
        n = int(n_voices)
        self.n = n
        self.frequency = np.full(n, 440.)
        self.amplitude = np.zeros(n)
        self.duty = np.full(n, .5)
        self.wave = np.zeros(n, dtype=np.int32)
        self.env = (Envelope * n)()
        for i in range(n):
            lib().ko_envelope_init(C.byref(self.env[i]), .1, .2, .8, 1., DT, 1e-3, 0, 0)
        self.coef_index = np.zeros(n, dtype=np.int32)
        self.coef_table = np.zeros((GRID_POINTS, 5))
        self.gain = np.ones(n)
        self.pan_l = np.ones(n)
        self.pan_r = np.ones(n)
        self.phase = np.zeros(n)
        self.z0 = np.zeros(n)
        self.z1 = np.zeros(n)
        self.has_env, self.has_filter, self.stereo = bool(has_env), bool(has_filter), bool(stereo)
        # modulation inputs (off by default)
        self.cutoff_lfo = False
        self.lfo_frequency, self.lfo_scale, self.lfo_offset = np.ones(n), np.ones(n), np.zeros(n)
        self.lfo_phase = np.zeros(n)
        self.lfo_wave = np.zeros(n, dtype=np.int32)
        self.grid = filter_grid()
        self.mod_target = MOD_NONE
        self.mod_frequency, self.mod_scale, self.mod_offset = np.ones(n), np.ones(n), np.zeros(n)
        self.mod_phase = np.zeros(n)
        self.mod_wave = np.zeros(n, dtype=np.int32)
        # effect stages (none by default)
        self.fx_kind = []
        self.fx_par = np.zeros((8, FX_PARAMS, n))
        self.fx_state = np.zeros((8, 2, n))
        self.fx_ring = np.zeros((n, 1))

    def set_effects(self, stages):
        """stages: [(FX_* kind, p0, p1, ...)], each p a scalar or [n] array."""
        self.fx_kind = [int(st[0]) for st in stages]
        self.fx_par[:] = 0.
        self.fx_state[:] = 0.
        for s, st in enumerate(stages):
            for k, p in enumerate(st[1:]):
                self.fx_par[s, k, :] = p
            if st[0] == FX_DELAY:
                lengths = (self.fx_par[s, 0] * 44100).astype(int)
                if lengths.min() < 1:
                    raise ValueError('empty delay line')
                # the RingBuffer's data array: the line and one buffer of tail (ring_buffer.py:38-42)
                self.fx_ring = np.zeros((self.n, int(lengths.max()) + BUFFER_SIZE))

    def set_envelope(self, i, attack, decay, sustain, release, overshoot=1e-3, retrigger=False,
                     loop=False):
        lib().ko_envelope_init(C.byref(self.env[i]), attack, decay, sustain, release, DT,
                               overshoot, int(retrigger), int(loop))

    def set_envelopes(self, attack, decay, sustain, release, overshoot=1e-3, retrigger=False,
                      loop=False):
        a, d, s, r = (np.broadcast_to(_f64(v), (self.n,)) for v in (attack, decay, sustain, release))
        o = np.broadcast_to(_f64(overshoot), (self.n,))
        rt = np.broadcast_to(np.asarray(retrigger, dtype=bool), (self.n,))
        lp = np.broadcast_to(np.asarray(loop, dtype=bool), (self.n,))
        for i in range(self.n):
            self.set_envelope(i, a[i], d[i], s[i], r[i], o[i], rt[i], lp[i])

    def _struct(self):
        ip = C.POINTER(C.c_int32)
        self.coef_table = _f64(self.coef_table)
        for name in ('lfo_frequency', 'lfo_scale', 'lfo_offset', 'lfo_phase', 'grid',
                     'mod_frequency', 'mod_scale', 'mod_offset', 'mod_phase'):
            setattr(self, name, _f64(getattr(self, name)))
        self.lfo_wave = np.ascontiguousarray(self.lfo_wave, dtype=np.int32)
        self.mod_wave = np.ascontiguousarray(self.mod_wave, dtype=np.int32)
        return Bank(self.n, int(self.has_env), int(self.has_filter), int(self.stereo),
                    _p(self.frequency), _p(self.amplitude), _p(self.duty),
                    self.wave.ctypes.data_as(ip), self.env,
                    self.coef_index.ctypes.data_as(ip), _p(self.coef_table),
                    _p(self.gain), _p(self.pan_l), _p(self.pan_r),
                    _p(self.phase), _p(self.z0), _p(self.z1),
                    int(self.cutoff_lfo), _p(self.lfo_frequency), _p(self.lfo_scale),
                    _p(self.lfo_offset), _p(self.lfo_phase), self.lfo_wave.ctypes.data_as(ip),
                    _p(self.grid), int(self.mod_target), _p(self.mod_frequency),
                    _p(self.mod_scale), _p(self.mod_offset), _p(self.mod_phase),
                    self.mod_wave.ctypes.data_as(ip),
                    len(self.fx_kind), (C.c_int * 8)(*(self.fx_kind + [0] * (8 - len(self.fx_kind)))),
                    _p(self.fx_par), _p(self.fx_state), _p(self.fx_ring), self.fx_ring.shape[1])

    def render(self, n_blocks, events=None, want_voices=False, n_threads=1):
        """events: structured array EVENT_DTYPE (any order; sorted here)."""
        if events is None:
            events = np.zeros(0, dtype=EVENT_DTYPE)
        events = np.asarray(events, dtype=EVENT_DTYPE)
        order = np.lexsort((events['block'], events['voice']))  # stable, voice-major
        events = np.ascontiguousarray(events[order])
        ch = 2 if self.stereo else 1
        mix = np.empty((n_blocks, ch, BUFFER_SIZE))
        voices = np.empty((self.n, n_blocks * BUFFER_SIZE)) if want_voices else None
        bank = self._struct()
        lib().ko_bank_render(C.byref(bank), int(n_blocks),
                             events.ctypes.data_as(C.POINTER(Event)), len(events),
                             _p(mix), _p(voices) if want_voices else None, int(n_threads))
        return mix, voices

    def render_synth(self, poly, n_blocks, notes=None, mono=False, want_voices=False):
        """PolyphonicSynthesizer / MonophonicSynthesizer routing and activity gating over
        this bank (synthesizer.py:151-219): notes = structured array NOTE_DTYPE (any order;
        sorted here by (instrument, block), stable).  -> out [instruments][n_blocks*256],
        voices.  Routing state (round robin, currentPitch) persists on the object."""
        poly = int(poly)
        n_inst = self.n // poly
        if not hasattr(self, 'round_robin'):
            self.round_robin = np.zeros(n_inst, dtype=np.int32)
            self.current_pitch = np.zeros(self.n, dtype=np.int32)
        if notes is None:
            notes = np.zeros(0, dtype=NOTE_DTYPE)
        notes = np.asarray(notes, dtype=NOTE_DTYPE)
        order = np.lexsort((notes['block'], notes['instrument']))
        notes = np.ascontiguousarray(notes[order])
        out = np.empty((n_inst, n_blocks * BUFFER_SIZE))
        voices = np.empty((self.n, n_blocks * BUFFER_SIZE)) if want_voices else None
        bank = self._struct()
        ip = C.POINTER(C.c_int32)
        lib().ko_synth_render(C.byref(bank), poly, int(bool(mono)),
                              self.round_robin.ctypes.data_as(ip),
                              self.current_pitch.ctypes.data_as(ip), int(n_blocks),
                              notes.ctypes.data_as(C.c_void_p), len(notes), _p(out),
                              _p(voices) if want_voices else None)
        return out, voices

    def env_state(self):
        return (np.array([e.stage for e in self.env]), np.array([e.value for e in self.env]),
                np.array([e.enabled for e in self.env]))


def bounce(bus, duration, fadeout=0., want_pcm=True):
    """Output side of an offline bounce (klanggeber.py:81-97, 205-249, wavfile.py:54-63):
    bus [n_blocks][channels][256] -> (frames float32 [n_blocks*256][channels], int16 pcm or
    None, peak)."""
    bus = np.ascontiguousarray(bus, dtype=np.float32)
    n_blocks, channels, _ = bus.shape
    frames = np.empty((n_blocks * BUFFER_SIZE, channels), dtype=np.float32)
    pcm = np.zeros((n_blocks * BUFFER_SIZE, channels), dtype=np.int16) if want_pcm else None
    peak = lib().ko_bounce(bus.ctypes.data, n_blocks, channels, float(duration), float(fadeout),
                           frames.ctypes.data, pcm.ctypes.data if want_pcm else None)
    return frames, pcm, float(peak)


CHIRP_METHODS = {'linear': 0, 'quadratic': 1, 'logarithmic': 2, 'hyperbolic': 3}


class OracleChirper:
    """klang.audio.oscillators.Chirper (oscillators.py:280-308)."""

    def __init__(self, start_frequency, duration, end_frequency, method='linear', wave_name='sine'):
        self.c = Chirper(start_frequency, duration, end_frequency, CHIRP_METHODS[method],
                         WAVE_IDS[wave_name])
        lib().ko_chirp_reset(C.byref(self.c))

    def block(self):
        out = np.empty(BUFFER_SIZE)
        lib().ko_chirp_block(C.byref(self.c), _p(out))
        return out

    @property
    def t(self):
        return np.array(self.c.t[:])


class OracleKick:
    """klang.audio.synthesizer.Kick (synthesizer.py:253-283)."""

    def __init__(self, frequency=40., decay=.8, intensity=2., pitch_decay=.3):
        self.c = Kick(frequency, decay, intensity, pitch_decay, 0.)

    def block(self, trigger=False):
        out = np.empty(BUFFER_SIZE)
        lib().ko_kick_block(C.byref(self.c), int(bool(trigger)), _p(out))
        return out

    @property
    def current_phase(self):
        return np.array(self.c.current_phase[:])

    @property
    def current_time(self):
        return self.c.current_time


def max_threads():
    return lib().ko_max_threads()
