"""ctypes binding of libklang_mi.so (include/klang_mi.h).

The HIP library is the only compute path of this package.  ``load()`` raises
``KlangError`` when the shared library is missing and ``Context()`` raises when
no GPU is present: there is no CPU fallback behind these classes.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

from klang_amd.errors import KlangError

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'csrc', 'libklang_mi.so')
HEADER_PATH = os.path.join(os.path.dirname(HERE), 'include', 'klang_mi.h')

BUFFER_SIZE = 256

# klmi_field
F_FREQUENCY, F_AMPLITUDE, F_WAVE, F_DUTY, F_ATTACK, F_DECAY, F_SUSTAIN, F_RELEASE, F_OVERSHOOT, \
    F_LOOP, F_RETRIGGER, F_CUTOFF, F_GAIN, F_PAN = range(14)
F_LFO_FREQUENCY, F_LFO_WAVE, F_LFO_SCALE, F_LFO_OFFSET, F_MOD_FREQUENCY, F_MOD_WAVE, \
    F_MOD_SCALE, F_MOD_OFFSET = range(14, 22)
F_PHASE, F_ENV_VALUE, F_ENV_STAGE, F_ENV_ENABLED, F_Z0, F_Z1, F_LFO_PHASE, F_MOD_PHASE = \
    range(32, 40)
F_CURRENT_PITCH, F_ROUND_ROBIN = 40, 41
MOD_NONE, MOD_FM, MOD_FREQUENCY, MOD_DUTY = range(4)   # klmi_mod_target
# klmi_fx_kind, KLMI_F_FX / KLMI_F_FX_STATE
FX_NONE, FX_FILTER, FX_GAIN, FX_TRANSFORM, FX_TANH, FX_OCTAVE, FX_TREMOLO, FX_DELAY, \
    FX_SUBSAMPLE, FX_BITCRUSH, FX_RINGMOD = range(11)
FX_PARAMS = 8
MAX_FX = 8


def F_FX(stage, k):
    """Field id of parameter k (0..7) of effect stage `stage`."""
    return 64 + FX_PARAMS * stage + k


def F_FX_STATE(stage, k):
    """Field id of state value k (0..1) of effect stage `stage`."""
    return 128 + 2 * stage + k

WAVE_SINE, WAVE_SQUARE, WAVE_SAWTOOTH, WAVE_TRIANGLE, WAVE_PWM, WAVE_ONE, WAVE_EXTERNAL = range(7)
EV_NOTE_OFF, EV_NOTE_ON, EV_SET_CUTOFF, EV_SET_FREQUENCY = range(4)
COMB_FORWARD, COMB_BACKWARD, COMB_ECHO = range(3)
PAN_MODES = {'linear': 0, 'constant_power': 1, 'mixture': 2}

EVENT_DTYPE = np.dtype([('voice', '<u4'), ('block', '<u4'), ('kind', '<u4'), ('reserved', '<u4'),
                        ('velocity', '<f8'), ('frequency', '<f8')])
NOTE_DTYPE = np.dtype([('instrument', '<u4'), ('block', '<u4'), ('pitch', '<i4'),
                       ('reserved', '<u4'), ('velocity', '<f8'), ('frequency', '<f8')])

EXPORTS = [
    'klmi_last_error', 'klmi_version', 'klmi_ctx_create', 'klmi_ctx_destroy',
    'klmi_ctx_synchronize', 'klmi_ctx_stream_rate', 'klmi_filter_grid', 'klmi_filter_grid_index', 'klmi_butter2_lowpass',
    'klmi_envelope_coefficients', 'klmi_low_pass_coefficients', 'klmi_panning_amplitudes', 'klmi_reverb_design',
    'klmi_bank_create', 'klmi_bank_destroy', 'klmi_bank_set_params', 'klmi_bank_state_get',
    'klmi_bank_state_set', 'klmi_bank_events', 'klmi_bank_notes',
    'klmi_bank_set_mix_divisor', 'klmi_bank_set_input', 'klmi_bank_set_input_from_bank',
    'klmi_bank_render', 'klmi_bank_render_device', 'klmi_bank_render_device_noalloc',
    'klmi_bank_render_voices',
    'klmi_bank_render_voices_f64', 'klmi_bank_render_voices_device', 'klmi_bank_reverb_state_get',
    'klmi_bank_last_kernel_ms', 'klmi_bounce_cycles', 'klmi_bank_bounce',
    'klmi_bank_bounce_device', 'klmi_bank_bounce_bus', 'klmi_env_create', 'klmi_env_destroy', 'klmi_env_set',
    'klmi_env_get', 'klmi_env_gate', 'klmi_env_sample', 'klmi_env_active', 'klmi_comb_create',
    'klmi_comb_destroy', 'klmi_comb_set_alpha', 'klmi_comb_filter',
    'klmi_kick_create', 'klmi_kick_destroy', 'klmi_kick_set', 'klmi_kick_trigger',
    'klmi_kick_render', 'klmi_kick_state_get',
    'klmi_chirp_create', 'klmi_chirp_destroy', 'klmi_chirp_reset', 'klmi_chirp_render',
    'klmi_chirp_state_get',
]


class ChainDesc(C.Structure):
    _fields_ = [('has_env', C.c_int32), ('has_filter', C.c_int32), ('has_reverb', C.c_int32),
                ('stereo', C.c_int32), ('pan_mode', C.c_int32), ('has_pan_law', C.c_int32),
                ('pan_law', C.c_double), ('reverb_decay', C.c_double),
                ('reverb_pre_delay', C.c_double), ('reverb_dry_wet', C.c_double),
                ('reverb_n_echos', C.c_int32), ('cutoff_lfo', C.c_int32),
                ('mod_target', C.c_int32), ('n_fx', C.c_int32), ('fx', C.c_uint8 * 8),
                ('gated', C.c_int32), ('poly_voices', C.c_int32),
                ('reverb_rings_f64', C.c_int32), ('struct_size', C.c_int32)]


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p
_lib = None


def load():
    """Load libklang_mi.so (built by klang_amd/csrc/Makefile). Loud on failure."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KlangError('HIP library missing: %s (build it with `make -C klang_amd/csrc`); '
                         'klang_amd has no CPU fallback' % LIB_PATH)
    # PyTorch's ROCm wheel bundles its own HIP runtime.  A process that uses both must load
    # torch's first: with the library's runtime loaded first,
    # whichever side initialises second finds no device.  torch is optional plumbing
    # (sharding, stream hand-over); without it the library runs on the system runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.klmi_last_error.restype = C.c_char_p
    L.klmi_version.restype = C.c_char_p
    L.klmi_ctx_create.argtypes = [C.c_int, _vp, C.POINTER(_vp)]
    L.klmi_ctx_destroy.argtypes = [_vp]
    L.klmi_ctx_synchronize.argtypes = [_vp]
    L.klmi_ctx_stream_rate.argtypes = [_vp, C.c_int, C.c_size_t, C.c_int, _dp]
    L.klmi_filter_grid.argtypes = [_dp]
    L.klmi_filter_grid_index.argtypes = [C.c_double]
    L.klmi_butter2_lowpass.argtypes = [C.c_double, _dp, _dp]
    L.klmi_envelope_coefficients.argtypes = [C.c_double] * 6 + [_dp]
    L.klmi_low_pass_coefficients.argtypes = [C.c_double, _dp, _dp]
    L.klmi_panning_amplitudes.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, _dp]
    L.klmi_reverb_design.argtypes = [C.c_double, C.c_double, C.c_int, _ip, _dp]
    L.klmi_bank_create.argtypes = [_vp, C.c_int, C.POINTER(ChainDesc), C.POINTER(_vp)]
    L.klmi_bank_destroy.argtypes = [_vp]
    L.klmi_bank_set_params.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_bank_state_get.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_bank_state_set.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_bank_events.argtypes = [_vp, _vp, C.c_size_t]
    L.klmi_bank_notes.argtypes = [_vp, _vp, C.c_size_t]
    L.klmi_bank_set_mix_divisor.argtypes = [_vp, C.c_double]
    L.klmi_bank_set_input.argtypes = [_vp, _dp, C.c_int]
    L.klmi_bank_set_input_from_bank.argtypes = [_vp, _vp, C.c_int]
    L.klmi_bank_render.argtypes = [_vp, C.c_int, _fp]
    L.klmi_bank_render_device.argtypes = [_vp, C.c_int, _vp]
    L.klmi_bank_render_device_noalloc.argtypes = [_vp, C.c_int, _vp]
    L.klmi_bank_render_voices.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _fp]
    L.klmi_bank_render_voices_f64.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_bank_render_voices_device.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp]
    L.klmi_bank_reverb_state_get.argtypes = [_vp, C.c_int, _dp, _ip]
    L.klmi_bounce_cycles.argtypes = [C.c_double]
    L.klmi_bank_bounce.argtypes = [_vp, C.c_double, C.c_double, _vp, _vp, _fp]
    L.klmi_bank_bounce_device.argtypes = [_vp, C.c_double, C.c_double, _vp, _vp, _vp]
    L.klmi_bank_bounce_bus.argtypes = [_vp, _vp, C.c_double, C.c_double, _vp, _vp, _fp]
    L.klmi_bank_last_kernel_ms.argtypes = [_vp, _fp, C.POINTER(C.c_int)]
    L.klmi_env_create.argtypes = [_vp, C.c_int] + [C.c_double] * 6 + [C.c_int, C.c_int,
                                                                       C.POINTER(_vp)]
    L.klmi_env_destroy.argtypes = [_vp]
    L.klmi_env_set.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_env_get.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_env_gate.argtypes = [_vp, C.c_int, C.c_int, C.c_int]
    L.klmi_env_sample.argtypes = [_vp, C.c_long, _dp]
    L.klmi_env_active.argtypes = [_vp, _ip]
    L.klmi_comb_create.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(_vp)]
    L.klmi_comb_destroy.argtypes = [_vp]
    L.klmi_comb_set_alpha.argtypes = [_vp, C.c_int, C.c_int, _dp]
    L.klmi_comb_filter.argtypes = [_vp, _dp, C.c_long, _dp]
    L.klmi_kick_create.argtypes = [_vp, C.c_int, C.POINTER(_vp)]
    L.klmi_kick_destroy.argtypes = [_vp]
    L.klmi_kick_set.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    L.klmi_kick_trigger.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.klmi_kick_render.argtypes = [_vp, C.c_int, _dp]
    L.klmi_kick_state_get.argtypes = [_vp, _dp, _dp]
    L.klmi_chirp_create.argtypes = [_vp, C.c_int, _dp, _dp, _dp, _ip, _ip, C.POINTER(_vp)]
    L.klmi_chirp_destroy.argtypes = [_vp]
    L.klmi_chirp_reset.argtypes = [_vp, C.c_int, C.c_int]
    L.klmi_chirp_render.argtypes = [_vp, C.c_int, _dp]
    L.klmi_chirp_state_get.argtypes = [_vp, _dp]
    _lib = L
    return L


class NativeError(KlangError):
    """A klmi_* call failed. ``code`` is the klmi_status."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


E_ARG, E_DEVICE, E_MEMORY, E_STATE = -1, -2, -3, -4


def _check(rc):
    if rc != 0:
        raise NativeError(rc, load().klmi_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


# ---- host-side design helpers ------------------------------------------------------
def filter_grid():
    g = np.empty(1000)
    _check(load().klmi_filter_grid(_p(g)))
    return g


def filter_grid_index(frequency):
    return load().klmi_filter_grid_index(float(frequency))


def butter2_lowpass(wn):
    b, a = np.empty(3), np.empty(3)
    _check(load().klmi_butter2_lowpass(float(wn), _p(b), _p(a)))
    return b, a


def low_pass_coefficients(frequency):
    """Single-pole low-pass (b, a) as lists, like effects.py:33-55."""
    b, a = np.empty(1), np.empty(2)
    _check(load().klmi_low_pass_coefficients(float(frequency), _p(b), _p(a)))
    return [float(b[0])], [float(a[0]), float(a[1])]


def envelope_coefficients(attack, decay, sustain, release, dt, overshoot=1e-3):
    out = np.empty(6)
    _check(load().klmi_envelope_coefficients(attack, decay, sustain, release, dt, overshoot,
                                             _p(out)))
    return out


def panning_amplitudes(level, mode='constant_power', pan_law=None):
    if mode not in PAN_MODES:
        raise ValueError('Invalid panning mode %r!' % mode)
    lr = np.empty(2)
    rc = load().klmi_panning_amplitudes(float(level), PAN_MODES[mode], int(pan_law is not None),
                                        float(pan_law or 0.), _p(lr))
    if rc == E_ARG:
        raise ValueError(load().klmi_last_error().decode())
    _check(rc)
    return lr


def reverb_design(decay=1.5, pre_delay=.03, n_echos=10):
    taps = np.empty(n_echos, dtype=np.int32)
    alphas = np.empty(n_echos)
    _check(load().klmi_reverb_design(decay, pre_delay, n_echos, taps.ctypes.data_as(_ip),
                                     _p(alphas)))
    return taps, alphas


# ---- objects --------------------------------------------------------------------------
# Device objects must be released before the HIP runtime's own exit handlers run:
# close whatever is still alive when the interpreter shuts down (children first).
_live_objects = weakref.WeakSet()
_live_contexts = weakref.WeakSet()


def _shutdown():
    for group in (_live_objects, _live_contexts):
        for obj in list(group):
            try:
                obj.close()
            except Exception:
                pass


atexit.register(_shutdown)


class Context:
    """One GPU. ``stream`` is a raw hipStream_t (int) or None."""

    def __init__(self, device=0, stream=None):
        self._h = _vp()
        _check(load().klmi_ctx_create(int(device), _vp(stream) if stream else None,
                                      C.byref(self._h)))
        self.device = device
        self.stream = int(stream) if stream else 0      # raw hipStream_t; 0 = the null stream
        self._children = weakref.WeakSet()
        _live_contexts.add(self)

    def synchronize(self):
        _check(load().klmi_ctx_synchronize(self._h))

    def stream_rate(self, kind, n_bytes=1 << 30, reps=10):
        """GB/s of a plain streaming kernel on this GPU: kind 'copy', 'update' (in-place
        read-modify-write), 'fill' or 'sum' (klmi_ctx_stream_rate)."""
        out = C.c_double()
        _check(load().klmi_ctx_stream_rate(self._h, ('copy', 'update', 'fill', 'sum').index(kind),
                                           int(n_bytes), int(reps), C.byref(out)))
        return out.value

    def close(self):
        """Destroys the context after closing every object created on it."""
        if self._h:
            for child in list(self._children):
                child.close()
            _check(load().klmi_ctx_destroy(self._h))
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Process-wide context on LOCAL_RANK's GPU (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get('LOCAL_RANK', '0')))
    return _default_ctx


class Bank:
    """Device-resident bank of independent voices (klmi_bank_*)."""

    def __init__(self, ctx, n_voices, has_env=True, has_filter=False, has_reverb=False,
                 stereo=False, pan_mode='constant_power', pan_law=None, reverb=None,
                 cutoff_lfo=False, mod_target=MOD_NONE, effects=(), gated=False, poly_voices=0):
        reverb = reverb or {}
        effects = [int(k) for k in effects]
        if len(effects) > MAX_FX:
            raise ValueError('at most %d effect stages per chain' % MAX_FX)
        self.desc = ChainDesc(int(has_env), int(has_filter), int(has_reverb), int(stereo),
                              PAN_MODES[pan_mode], int(pan_law is not None), float(pan_law or 0.),
                              float(reverb.get('decay', 1.5)), float(reverb.get('pre_delay', .03)),
                              float(reverb.get('dry_wet', .7)), int(reverb.get('n_echos', 10)),
                              int(cutoff_lfo), int(mod_target), len(effects),
                              (C.c_uint8 * 8)(*(effects + [0] * (MAX_FX - len(effects)))),
                              int(gated), int(poly_voices),
                              int(reverb.get('rings', 'f32') == 'f64'), C.sizeof(ChainDesc))
        self.effects = effects
        self.ctx = ctx
        self.n = int(n_voices)
        self.channels = 2 if stereo else 1
        self._h = _vp()
        _check(load().klmi_bank_create(ctx._h, self.n, C.byref(self.desc), C.byref(self._h)))
        _live_objects.add(self)
        ctx._children.add(self)

    def close(self):
        if self._h:
            load().klmi_bank_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, field, values, first=0):
        """Per-voice values for voices [first, first + len(values)); a true scalar (0-d) is
        one value for every voice from `first` on.  A sequence of one value sets one voice."""
        if np.ndim(values) == 0:
            values = np.full(self.n - first, float(values))
        values = _f64(values).ravel()
        _check(load().klmi_bank_set_params(self._h, field, first, values.size, _p(values)))

    def get_state(self, field, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty(count)
        _check(load().klmi_bank_state_get(self._h, field, first, count, _p(out)))
        return out

    def set_state(self, field, values, first=0):
        values = _f64(values).ravel()
        _check(load().klmi_bank_state_set(self._h, field, first, values.size, _p(values)))

    def load_patch(self, patch):
        """Parameters and events of a klang_amd.patches.Patch."""
        self.set(F_WAVE, patch.wave)
        self.set(F_FREQUENCY, patch.frequency)
        self.set(F_ATTACK, patch.attack)
        self.set(F_DECAY, patch.decay)
        self.set(F_SUSTAIN, patch.sustain)
        self.set(F_RELEASE, patch.release)
        self.set(F_OVERSHOOT, patch.overshoot)
        self.set(F_LOOP, patch.loop)
        self.set(F_RETRIGGER, patch.retrigger)
        self.set(F_CUTOFF, patch.cutoff)
        self.set(F_GAIN, patch.gain)
        self.set(F_PAN, patch.pan)
        self.events(patch.events)

    def events(self, events):
        """events: structured array with fields voice, block, kind, velocity, frequency."""
        ev = np.zeros(len(events), dtype=EVENT_DTYPE)
        for name in ('voice', 'block', 'kind', 'velocity', 'frequency'):
            ev[name] = events[name]
        _check(load().klmi_bank_events(self._h, ev.ctypes.data_as(_vp), len(ev)))

    def event(self, voice, block, kind, velocity=0., frequency=0.):
        ev = np.zeros(1, dtype=EVENT_DTYPE)
        ev[0] = (voice, block, kind, 0, velocity, frequency)
        _check(load().klmi_bank_events(self._h, ev.ctypes.data_as(_vp), 1))

    def notes(self, notes):
        """Notes for the instruments of a polyphonic bank: structured array with fields
        instrument, block, pitch, velocity, frequency (klmi_bank_notes)."""
        nt = np.zeros(len(notes), dtype=NOTE_DTYPE)
        for name in ('instrument', 'block', 'pitch', 'velocity', 'frequency'):
            nt[name] = notes[name]
        _check(load().klmi_bank_notes(self._h, nt.ctypes.data_as(_vp), len(nt)))

    def note(self, instrument, block, pitch, velocity, frequency):
        nt = np.zeros(1, dtype=NOTE_DTYPE)
        nt[0] = (instrument, block, pitch, 0, velocity, frequency)
        _check(load().klmi_bank_notes(self._h, nt.ctypes.data_as(_vp), 1))

    def set_mix_divisor(self, divisor):
        _check(load().klmi_bank_set_mix_divisor(self._h, float(divisor)))

    def set_input(self, x):
        x = _f64(x).reshape(self.n, -1)
        assert x.shape[1] % BUFFER_SIZE == 0
        _check(load().klmi_bank_set_input(self._h, _p(x), x.shape[1] // BUFFER_SIZE))

    def set_input_from(self, source, n_blocks):
        """The first n_blocks buffers of `source`'s last render() as this bank's external
        input, device to device (klmi_bank_set_input_from_bank)."""
        _check(load().klmi_bank_set_input_from_bank(self._h, source._h, int(n_blocks)))

    def render(self, n_blocks):
        """-> float32 [n_blocks][channels][256]; advances state, consumes events."""
        out = np.empty((n_blocks, self.channels, BUFFER_SIZE), dtype=np.float32)
        _check(load().klmi_bank_render(self._h, int(n_blocks), out.ctypes.data_as(_fp)))
        return out

    def render_device(self, n_blocks, dev_ptr):
        _check(load().klmi_bank_render_device(self._h, int(n_blocks), _vp(dev_ptr)))

    def render_device_noalloc(self, n_blocks, dev_ptr):
        """render_device through the library's static scratch array: allocates nothing."""
        _check(load().klmi_bank_render_device_noalloc(self._h, int(n_blocks), _vp(dev_ptr)))

    def render_voices(self, n_blocks, first=0, count=None, f64=False):
        """Preview of per-voice streams [count][n_blocks*256]; state untouched."""
        count = self.n - first if count is None else count
        if f64:
            out = np.empty((count, n_blocks * BUFFER_SIZE))
            _check(load().klmi_bank_render_voices_f64(self._h, first, count, int(n_blocks),
                                                      _p(out)))
        else:
            out = np.empty((count, n_blocks * BUFFER_SIZE), dtype=np.float32)
            _check(load().klmi_bank_render_voices(self._h, first, count, int(n_blocks),
                                                  out.ctypes.data_as(_fp)))
        return out

    def render_voices_device(self, n_blocks, dev_ptr, first=0, count=None):
        count = self.n - first if count is None else count
        _check(load().klmi_bank_render_voices_device(self._h, first, count, int(n_blocks),
                                                     _vp(dev_ptr)))

    def bounce(self, duration, fadeout=0., want_pcm=True):
        """Offline bounce of the whole piece on the device (klmi_bank_bounce):
        -> (frames float32 [cycles*256][channels], int16 pcm or None, peak).  Raises
        NativeError (E_ARG) for a silent piece when pcm is wanted, like write_wave."""
        cycles = load().klmi_bounce_cycles(float(duration)) if np.isfinite(duration) else 0
        shape = (max(cycles, 0) * BUFFER_SIZE, self.channels)
        frames = np.empty(shape, dtype=np.float32)
        pcm = np.empty(shape, dtype=np.int16) if want_pcm else None
        peak = C.c_float(0.)
        _check(load().klmi_bank_bounce(self._h, float(duration), float(fadeout),
                                       frames.ctypes.data_as(_vp),
                                       pcm.ctypes.data_as(_vp) if want_pcm else None,
                                       C.byref(peak)))
        return frames, pcm, peak.value

    def bounce_bus(self, bus_ptr, duration, fadeout=0., want_pcm=True):
        """The output side of bounce() for a device bus [cycles][channels][256] float32 that was
        rendered and summed elsewhere (klmi_bank_bounce_bus) -> (frames, pcm or None, peak)."""
        cycles = load().klmi_bounce_cycles(float(duration)) if np.isfinite(duration) else 0
        shape = (max(cycles, 0) * BUFFER_SIZE, self.channels)
        frames = np.empty(shape, dtype=np.float32)
        pcm = np.empty(shape, dtype=np.int16) if want_pcm else None
        peak = C.c_float(0.)
        _check(load().klmi_bank_bounce_bus(self._h, _vp(bus_ptr), float(duration), float(fadeout),
                                           frames.ctypes.data_as(_vp),
                                           pcm.ctypes.data_as(_vp) if want_pcm else None,
                                           C.byref(peak)))
        return frames, pcm, peak.value

    def bounce_device(self, duration, fadeout, frames_ptr, pcm_ptr, peak_ptr):
        _check(load().klmi_bank_bounce_device(self._h, float(duration), float(fadeout),
                                              frames_ptr, pcm_ptr, peak_ptr))

    def reverb_state(self, voice):
        taps, _ = reverb_design(self.desc.reverb_decay, self.desc.reverb_pre_delay,
                                self.desc.reverb_n_echos)
        rings = np.empty(int(taps.sum()))
        pos = np.empty(len(taps), dtype=np.int32)
        _check(load().klmi_bank_reverb_state_get(self._h, voice, _p(rings),
                                                 pos.ctypes.data_as(_ip)))
        return np.split(rings, np.cumsum(taps)[:-1]), pos

    def last_kernel_ms(self):
        ms = C.c_float()
        launches = C.c_int()
        _check(load().klmi_bank_last_kernel_ms(self._h, C.byref(ms), C.byref(launches)))
        return ms.value, launches.value


class EnvelopeBatch:
    """`count` copies of klang.audio._envelope.Envelope on the GPU (klmi_env_*)."""

    def __init__(self, ctx, count, attack, decay, sustain, release, dt, overshoot=1e-3,
                 retrigger=False, loop=False):
        self.ctx = ctx
        self.count = int(count)
        self._h = _vp()
        _check(load().klmi_env_create(ctx._h, self.count, attack, decay, sustain, release, dt,
                                      overshoot, int(bool(retrigger)), int(bool(loop)),
                                      C.byref(self._h)))
        _live_objects.add(self)
        ctx._children.add(self)

    def close(self):
        if self._h:
            load().klmi_env_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, field, values, first=0):
        values = _f64(values).ravel()
        _check(load().klmi_env_set(self._h, field, first, values.size, _p(values)))

    def get(self, field, first=0, count=None):
        count = self.count - first if count is None else count
        out = np.empty(count)
        _check(load().klmi_env_get(self._h, field, first, count, _p(out)))
        return out

    def gate(self, on, first=0, count=None):
        count = self.count - first if count is None else count
        _check(load().klmi_env_gate(self._h, first, count, int(bool(on))))

    def sample(self, n):
        out = np.empty((self.count, max(int(n), 0)))
        _check(load().klmi_env_sample(self._h, int(n), _p(out)))
        return out

    def active(self):
        out = np.empty(self.count, dtype=np.int32)
        _check(load().klmi_env_active(self._h, out.ctypes.data_as(_ip)))
        return out.astype(bool)


class CombBatch:
    """`count` ring-buffer filters of one kind and length (klmi_comb_*)."""

    def __init__(self, ctx, kind, count, length, alpha=.9):
        self.ctx = ctx
        self.count = int(count)
        self.length = int(length)
        self._h = _vp()
        _check(load().klmi_comb_create(ctx._h, int(kind), self.count, self.length, float(alpha),
                                       C.byref(self._h)))
        _live_objects.add(self)
        ctx._children.add(self)

    def close(self):
        if self._h:
            load().klmi_comb_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_alpha(self, values, first=0):
        values = _f64(values).ravel()
        _check(load().klmi_comb_set_alpha(self._h, first, values.size, _p(values)))

    def filter(self, x):
        x = _f64(x).reshape(self.count, -1)
        y = np.empty_like(x)
        _check(load().klmi_comb_filter(self._h, _p(x), x.shape[1], _p(y)))
        return y


KICK_FREQUENCY, KICK_DECAY, KICK_INTENSITY, KICK_PITCH_DECAY = range(4)


class KickBatch:
    """count x klang.audio.synthesizer.Kick on the device (klmi_kick_*)."""

    def __init__(self, ctx, count):
        self.ctx, self.count = ctx, int(count)
        self._h = _vp()
        _check(load().klmi_kick_create(ctx._h, self.count, C.byref(self._h)))
        _live_objects.add(self)
        ctx._children.add(self)

    def close(self):
        if self._h:
            load().klmi_kick_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, field, values, first=0):
        values = _f64(values).ravel()
        _check(load().klmi_kick_set(self._h, field, first, values.size, _p(values)))

    def trigger(self, which, block):
        which = np.ascontiguousarray(np.atleast_1d(which), dtype=np.uint32)
        block = np.ascontiguousarray(np.atleast_1d(block), dtype=np.uint32)
        _check(load().klmi_kick_trigger(self._h, which.ctypes.data_as(_vp),
                                        block.ctypes.data_as(_vp), which.size))

    def render(self, n_blocks):
        out = np.empty((self.count, int(n_blocks) * BUFFER_SIZE))
        _check(load().klmi_kick_render(self._h, int(n_blocks), _p(out)))
        return out

    def state(self):
        time, phase = np.empty(self.count), np.empty((self.count, BUFFER_SIZE))
        _check(load().klmi_kick_state_get(self._h, _p(time), _p(phase)))
        return time, phase


CHIRP_METHODS = {'linear': 0, 'lin': 0, 'li': 0, 'quadratic': 1, 'quad': 1, 'q': 1,
                 'logarithmic': 2, 'log': 2, 'lo': 2, 'hyperbolic': 3, 'hyp': 3}


class ChirpBatch:
    """count x klang.audio.oscillators.Chirper on the device (klmi_chirp_*).  Arguments are
    arrays of `count` values; methods by scipy's names; waves as klmi_wave ids."""

    def __init__(self, ctx, start_frequency, duration, end_frequency, method='linear',
                 wave=WAVE_SINE):
        f0 = _f64(np.atleast_1d(start_frequency))
        self.count = f0.size
        t1 = _f64(np.broadcast_to(duration, f0.shape))
        f1 = _f64(np.broadcast_to(end_frequency, f0.shape))
        methods = np.broadcast_to(np.atleast_1d(method), f0.shape)
        for m in methods:
            if m not in CHIRP_METHODS:   # scipy's message
                raise ValueError("method must be 'linear', 'quadratic', 'logarithmic', or "
                                 "'hyperbolic', but a value of %r was given." % (m,))
        mid = np.ascontiguousarray([CHIRP_METHODS[m] for m in methods], dtype=np.int32)
        wid = np.ascontiguousarray(np.broadcast_to(wave, f0.shape), dtype=np.int32)
        self.ctx = ctx
        self._h = _vp()
        _check(load().klmi_chirp_create(ctx._h, self.count, _p(f0), _p(t1), _p(f1),
                                        mid.ctypes.data_as(_ip), wid.ctypes.data_as(_ip),
                                        C.byref(self._h)))
        _live_objects.add(self)
        ctx._children.add(self)

    def close(self):
        if self._h:
            load().klmi_chirp_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, first=0, count=None):
        _check(load().klmi_chirp_reset(self._h, first, self.count - first if count is None else count))

    def render(self, n_blocks):
        out = np.empty((self.count, int(n_blocks) * BUFFER_SIZE))
        _check(load().klmi_chirp_render(self._h, int(n_blocks), _p(out)))
        return out

    def time_vectors(self):
        t = np.empty((self.count, BUFFER_SIZE))
        _check(load().klmi_chirp_state_get(self._h, _p(t)))
        return t
