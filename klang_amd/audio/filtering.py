"""Ring-buffer filters, the Butterworth ``Filter`` block and ``Reverb`` (host side).

Mirrors klang/audio/filters.py:34-108 + klang/audio/_filters.c:231-306 (comb and
echo filters), klang/audio/effects.py:239-385 (Filter) and :564-615 (Reverb).
The ``update()`` / ``filter()`` methods here are the ``backend='host'``
evaluation; fused chains use the HIP kernels (biquad in bank_kernels.h, combs in
reverb_kernels.h / klmi_comb_*).  Filter coefficients always come from the
library's own design helpers (klmi_filter_grid / klmi_butter2_lowpass), so host
and device evaluate the same table.
"""
import functools
import math

import numpy as np

from klang_amd import native
from klang_amd.audio.signals import NYQUIST_FREQUENCY
from klang_amd.block import Block
from klang_amd.config import KAMMERTON, SAMPLING_RATE
from klang_amd.constants import MONO, PI, STEREO

__all__ = ['ForwardCombFilter', 'BackwardCombFilter', 'EchoFilter', 'FilterCoefficients',
           'Filter', 'Reverb', 'blend', 'clip', 'find_next_primes']

DEFAULT_ALPHA = .9
MAX_RING_LENGTH = 60 * SAMPLING_RATE   # _filters.c MAX_RING_BUFFER_CAPACITY


def clip(value, lower, upper):
    """klang/math.py:24-31."""
    return min(max(value, lower), upper)


def blend(a, b, x):
    """Dry/wet blend (klang/math.py:68-75)."""
    return (1. - x) * a + x * b


def find_next_primes(nPrimes, start=1):
    """The next `nPrimes` primes >= start (klang/primes.py:133-141)."""
    def is_prime(v):
        if v < 2:
            return False
        return all(v % d for d in range(2, math.isqrt(v) + 1))

    primes, v = [], int(start)
    while len(primes) < nPrimes:
        if is_prime(v):
            primes.append(v)
        v += 1
    return primes


class _RingFilter:
    """Shared part of the three ring-buffer filters: ``length`` zeroed samples,
    a read/write position and a gain ``alpha`` (_filters.c:98-186)."""

    def __init__(self, length, alpha=DEFAULT_ALPHA):
        length = int(length)
        if length <= 0 or length > MAX_RING_LENGTH:
            raise RuntimeError('Could not initialize RingBuffer!')   # _filters.c:132-136
        self.alpha = float(alpha)
        self._ring = np.zeros(length)
        self._pos = 0

    @property
    def length(self):
        return self._ring.size

    def _step(self, x, old):
        """-> (y, value stored back) for vectors of simultaneous samples."""
        raise NotImplementedError

    def filter(self, x):
        x = np.asarray(x, dtype=float)
        if x.ndim != 1:
            raise ValueError('samples have to be ndim 1!')           # _filters.c:202-205
        y = np.empty_like(x)
        L, done = self._ring.size, 0
        while done < x.size:
            # up to `L` consecutive samples (and up to the wrap point) only read
            # ring entries written before this chunk
            n = min(x.size - done, L - self._pos)
            seg = slice(self._pos, self._pos + n)
            y[done:done + n], self._ring[seg] = self._step(x[done:done + n],
                                                           self._ring[seg].copy())
            self._pos = (self._pos + n) % L
            done += n
        return y


class ForwardCombFilter(_RingFilter):
    """y[i] = x[i] + alpha * x[i-K] (_filters.c:231-254)."""

    def _step(self, x, old):
        return x + self.alpha * old, x


class BackwardCombFilter(_RingFilter):
    """y[i] = x[i] + alpha * y[i-K] (_filters.c:257-280)."""

    def _step(self, x, old):
        y = x + self.alpha * old
        return y, y


class EchoFilter(_RingFilter):
    """y[i] = ring; ring = alpha * y[i] + x[i] (_filters.c:283-306)."""

    def _step(self, x, old):
        return old, self.alpha * old + x


@functools.lru_cache(maxsize=None)
def _library_table():
    grid = native.filter_grid()
    table = [native.butter2_lowpass(f / NYQUIST_FREQUENCY) for f in grid]
    return grid, table


class FilterCoefficients:
    """The 1000-point log-spaced cutoff grid and its Butterworth-2 low-pass
    designs (effects.py:239-274), computed by the library (closed-form bilinear
    design; agrees with scipy.signal.butter to <= 4e-15 relative)."""
    F_MIN = 20.
    F_MAX = 20000.

    def __init__(self):
        self.frequencies, self.coefficients = _library_table()

    def index(self, frequency):
        return native.filter_grid_index(float(frequency))

    def get_coefficients(self, frequency):
        return self.coefficients[self.index(frequency)]


class _Filter:
    """Chunk-wise direct-form-II-transposed filtering with carried state
    (effects.py:283-320; scipy.signal.lfilter there)."""

    def __init__(self):
        self.coefficients = FilterCoefficients()
        self.set_frequency(.5 * NYQUIST_FREQUENCY)
        self.reset()

    def set_frequency(self, frequency):
        self.currentCoeffs = self.coefficients.get_coefficients(frequency)

    def reset(self):
        self.state = np.zeros(2)

    def filter(self, signal):
        (b0, b1, b2), (_, a1, a2) = self.currentCoeffs
        z0, z1 = self.state
        out = np.empty(len(signal))
        for i, x in enumerate(signal):
            y = z0 + b0 * x
            z0 = z1 + b1 * x - a1 * y
            z1 = b2 * x - a2 * y
            out[i] = y
        self.state = np.array([z0, z1])
        return out


class Filter(Block):
    """Butterworth-2 low-pass with a cutoff input quantised to the grid
    (effects.py:342-385).  Only the reference's default design
    (scipy.signal.butter, N=2, 'lowpass') is offered."""
    MAX_CHANNELS = STEREO

    def __init__(self, frequency=KAMMERTON):
        super().__init__(nInputs=2, nOutputs=1)
        _, self.frequency = self.inputs
        self.frequency.set_value(frequency)
        self.filters = [_Filter() for _ in range(self.MAX_CHANNELS)]
        self._seen = None

    def update_frequency(self):
        freq = float(self.frequency.value)
        for fil in self.filters:
            fil.set_frequency(freq)

    def update(self):
        value = self.frequency.get_value()
        if self._seen is None or self._seen != value:   # Observer, effects.py:323-339
            self._seen = value
            self.update_frequency()
        signal = self.input.get_value()
        if np.ndim(signal) == MONO:
            out = self.filters[0].filter(signal)
        else:
            out = np.array([f.filter(ch) for f, ch in zip(self.filters, signal)])
        self.output.set_value(out)


class Reverb(Block):
    """Parallel prime-length feedback combs, mean, dry/wet (effects.py:564-615)."""

    def __init__(self, decay=1.5, preDelay=.03, dryWet=.7, nEchos=10,
                 echoType=BackwardCombFilter):
        assert nEchos > 0
        super().__init__(nInputs=1, nOutputs=1)
        self.decay, self.preDelay, self.nEchos = decay, preDelay, nEchos
        self.dryWet = clip(dryWet, 0., 1.)
        self.echoType = echoType
        self.filters = [echoType(k, self.compute_alpha(k / SAMPLING_RATE, decay))
                        for k in find_next_primes(nEchos, int(preDelay * SAMPLING_RATE))]

    @staticmethod
    def compute_alpha(delayTime, decay):
        if decay <= 0:
            return 0.
        return math.exp(-PI * delayTime / decay)

    def update(self):
        x = self.input.value
        y = self.filters[0].filter(x)
        for fil in self.filters[1:]:
            y += fil.filter(x)
        y /= len(self.filters)
        self.output.set_value(blend(x, y, self.dryWet))
