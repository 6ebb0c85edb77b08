"""Time base, wave functions and oscillator blocks (host side).

Mirrors klang/audio/helpers.py:9-42, klang/audio/waves.py:22-53 and
klang/audio/oscillators.py:41-267.  The ``update()`` methods here are the
explicit ``backend='host'`` evaluation of an un-fused block (NumPy, like the
reference); under the default backend the graph compiler (klang_amd/fusion.py)
hands oscillator -> envelope -> filter -> mixer chains to the HIP library and
these methods are not called for fused blocks.
"""
import functools

import numpy as np

from klang_amd.block import Block
from klang_amd.config import BUFFER_SIZE, SAMPLING_RATE
from klang_amd.constants import PI, SCALAR, TAU
from klang_amd.tempo import compute_rate
from klang_amd.wiring import Input

DT = 1. / SAMPLING_RATE
INTERVAL = DT * BUFFER_SIZE
NYQUIST_FREQUENCY = SAMPLING_RATE // 2


@functools.lru_cache()
def get_silence(shape, dtype=float):
    arr = np.zeros(shape, dtype)
    arr.setflags(write=False)
    return arr


@functools.lru_cache()
def get_time(length, dt=DT):
    arr = dt * np.arange(length)
    arr.setflags(write=False)
    return arr


MONO_SILENCE = get_silence(BUFFER_SIZE)
STEREO_SILENCE = get_silence((2, BUFFER_SIZE))
T = get_time(BUFFER_SIZE, DT)


def wrap(phase):
    """Phase to [-pi, pi) (klang/math.py:9-11)."""
    return (phase + PI) % TAU - PI


def sine(phase):
    return np.sin(phase)


def square(phase):
    return 2. * (wrap(phase) >= 0) - 1.


def sawtooth(phase):
    return 2. * np.mod(phase, TAU) / TAU - 1.


def triangle(phase):
    return 1. - np.abs((4 * wrap(phase) / TAU) % 4 - 2)


def random(phase):
    return 2. * np.random.random(size=np.shape(phase)) - 1.


WAVE_FUNCTIONS = {'sine': sine, 'square': square, 'sawtooth': sawtooth, 'triangle': triangle,
                  'random': random}

# wave function -> klmi_wave id for the device path ('random' has no device twin)
WAVE_IDS = {sine: 0, square: 1, sawtooth: 2, triangle: 3}


def sample_wave(frequency, startPhase=0., wave_func=None, shape=BUFFER_SIZE):
    """`shape` samples of a wave and the next start phase (waves.py:56-61; the Kick's
    oscillator).  Unlike sample_phase the phase is not reduced before the wave function, and
    with an array of frequencies the next start phase is an array too."""
    t = get_time(shape)
    phase = TAU * frequency * t + startPhase
    return (wave_func or sine)(phase), wrap(phase[-1] + TAU * frequency * DT)


def sample_phase(frequency, startPhase=0.):
    """BUFFER_SIZE phase samples and the next start phase (oscillators.py:41-67);
    scalar frequency or an array of BUFFER_SIZE frequencies."""
    if np.ndim(frequency) == 0:
        phase = TAU * frequency * get_time(BUFFER_SIZE + 1, DT) + startPhase
    else:
        phase = np.empty(BUFFER_SIZE + 1)
        phase[0] = startPhase
        phase[1:] = TAU * DT * np.cumsum(frequency) + startPhase
    phase = np.mod(phase, TAU)
    return phase[:-1], phase[-1]


def linear_mapping(xRange, yRange):
    """Coefficients a, b of y = a x + b through two points (klang/math.py:83-98)."""
    (x0, x1), (y0, y1) = xRange, yRange
    return np.linalg.solve([[x0, 1.], [x1, 1.]], [y0, y1])


class Phasor(Block):
    """One phase value in [0, TAU) per buffer (oscillators.py:70-108)."""

    def __init__(self, frequency=1., startPhase=0.):
        super().__init__(nInputs=1, nOutputs=1)
        self.frequency, = self.inputs
        self.frequency.set_value(frequency)
        self.currentPhase = startPhase

    def _advance_scalar(self):
        phase = self.currentPhase
        freq = compute_rate(self.frequency.value)
        self.currentPhase = (TAU * freq * INTERVAL + self.currentPhase) % TAU
        return phase

    def sample(self):
        return self._advance_scalar()

    def update(self):
        self.output.set_value(self.sample())

    def _ctor_kwargs(self):
        return dict(frequency=self.frequency.value, startPhase=self.currentPhase)

    def __deepcopy__(self, memo):
        return type(self)(**self._ctor_kwargs())

    def __str__(self):
        return '%s(%.1f Hz)' % (type(self).__name__, self.frequency.value)


class Oscillator(Phasor):
    """BUFFER_SIZE audio samples per buffer (oscillators.py:111-140)."""

    def __init__(self, frequency=440., wave_func=sine, startPhase=0.):
        super().__init__(frequency, startPhase)
        self.wave_func = wave_func

    def sample(self):
        freq = compute_rate(self.frequency.value)
        phase, self.currentPhase = sample_phase(freq, self.currentPhase)
        return self.wave_func(phase)

    def _ctor_kwargs(self):
        return dict(super()._ctor_kwargs(), wave_func=self.wave_func)


class Lfo(Phasor):
    """Low frequency oscillator with an output range, scalar or per-sample
    (oscillators.py:143-190)."""

    def __init__(self, frequency=1., wave_func=sine, shape=1, outputRange=(0, 1), startPhase=0.):
        assert shape in {SCALAR, BUFFER_SIZE}
        super().__init__(frequency, startPhase)
        self.wave_func = wave_func
        self.shape = shape
        self.outputRange = outputRange
        self.scale, self.offset = linear_mapping((-1, 1), outputRange)

    def sample(self):
        if self.shape == SCALAR:
            phase = self._advance_scalar()
        else:
            freq = compute_rate(self.frequency.value)
            phase, self.currentPhase = sample_phase(freq, self.currentPhase)
        return self.scale * self.wave_func(phase) + self.offset

    def _ctor_kwargs(self):
        return dict(super()._ctor_kwargs(), wave_func=self.wave_func, shape=self.shape,
                    outputRange=self.outputRange)


class FmOscillator(Oscillator):
    """Frequency modulation oscillator: wave(phase + intensity * modulator), the
    modulator being an inner Oscillator of the same wave function
    (oscillators.py:193-234)."""

    def __init__(self, frequency=440., intensity=1., modFrequency=10., wave_func=sine,
                 startPhase=0.):
        super().__init__(frequency, wave_func, startPhase)
        self.intensity = intensity
        self.modulator = Oscillator(modFrequency, wave_func=wave_func)

    def sample(self):
        freq = compute_rate(self.frequency.value)
        phase, self.currentPhase = sample_phase(freq, self.currentPhase)
        return self.wave_func(phase + self.intensity * self.modulator.output.value)

    def update(self):
        self.modulator.update()
        self.output.set_value(self.sample())

    def _ctor_kwargs(self):
        return dict(frequency=self.frequency.value, intensity=self.intensity,
                    modFrequency=self.modulator.frequency.value, wave_func=self.wave_func,
                    startPhase=self.currentPhase)


class PwmOscillator(Phasor):
    """Pulse wave with a duty-cycle input (oscillators.py:237-267)."""

    def __init__(self, frequency=440., dutyCycle=.5, startPhase=0.):
        super().__init__(frequency, startPhase)
        self.dutyCycle = Input(owner=self)
        self.dutyCycle.set_value(dutyCycle)
        self.inputs.append(self.dutyCycle)

    def sample(self):
        freq = compute_rate(self.frequency.value)
        phase, self.currentPhase = sample_phase(freq, self.currentPhase)
        return 2. * (phase < TAU * self.dutyCycle.value) - 1.

    def _ctor_kwargs(self):
        return dict(super()._ctor_kwargs(), dutyCycle=self.dutyCycle.value)


def chirp_phase(t, freqStart, tEnd, freqEnd, method='linear'):
    """Chirp phase with the constant-frequency extension for t >= tEnd
    (klang/audio/oscillators.py:18-38): a buffer that starts at or after tEnd (or a sweep with
    equal ends) is TAU * freqEnd * t with no offset, a buffer that straddles tEnd continues
    the sweep's phase from its first sample at or after tEnd."""
    t = np.asarray(t, dtype=float)
    if tEnd <= t[0] or freqStart == freqEnd:
        return TAU * freqEnd * t
    phase = sweep_phase(t, freqStart, tEnd, freqEnd, method)
    if t[-1] < tEnd:
        return phase
    mid = (t >= tEnd).argmax()
    offset = phase[mid] - TAU * freqEnd * t[mid]
    phase[mid:] = TAU * freqEnd * t[mid:] + offset
    return phase


def sweep_phase(t, f0, t1, f1, method='linear'):
    """Phase of a frequency sweep from f0 (at t = 0) to f1 (at t = t1): scipy.signal's
    `_chirp_phase` with vertex_zero=True, which the reference imports
    (klang/audio/oscillators.py:3, 27), restated so that no private SciPy name is needed."""
    t = np.asarray(t)
    f0, t1, f1 = float(f0), float(t1), float(f1)
    if method in ('linear', 'lin', 'li'):
        beta = (f1 - f0) / t1
        return 2 * np.pi * (f0 * t + 0.5 * beta * t * t)
    if method in ('quadratic', 'quad', 'q'):
        beta = (f1 - f0) / (t1 ** 2)
        return 2 * np.pi * (f0 * t + beta * t ** 3 / 3)
    if method in ('logarithmic', 'log', 'lo'):
        if f0 * f1 <= 0.0:
            raise ValueError('For a logarithmic chirp, f0 and f1 must be nonzero and have the '
                             'same sign.')
        if f0 == f1:
            return 2 * np.pi * f0 * t
        beta = t1 / np.log(f1 / f0)
        return 2 * np.pi * beta * f0 * (np.power(f1 / f0, t / t1) - 1.0)
    if method in ('hyperbolic', 'hyp'):
        if f0 == 0 or f1 == 0:
            raise ValueError('For a hyperbolic chirp, f0 and f1 must be nonzero.')
        if f0 == f1:
            return 2 * np.pi * f0 * t
        sing = -f1 * t1 / (f0 - f1)
        return 2 * np.pi * (-sing * f0) * np.log(np.abs(1 - t / sing))
    raise ValueError("method must be 'linear', 'quadratic', 'logarithmic', or 'hyperbolic', but "
                     'a value of %r was given.' % (method,))


class Chirper(Block):
    """Chirp oscillator: frequency ascends or descends from startFrequency to endFrequency
    over `duration` and keeps going (oscillators.py:280-308)."""

    def __init__(self, startFrequency, duration, endFrequency, method='linear', wave_func=sine):
        super().__init__(nOutputs=1)
        self.startFrequency, self.duration, self.endFrequency = startFrequency, duration, endFrequency
        self.method, self.wave_func = method, wave_func
        self.t = get_time(BUFFER_SIZE).copy()

    def reset(self):
        self.t = get_time(BUFFER_SIZE).copy()

    def update(self):
        phase = chirp_phase(self.t, self.startFrequency, self.duration, self.endFrequency,
                            self.method)
        self.output.set_value(self.wave_func(phase))
        self.t += DT * BUFFER_SIZE
