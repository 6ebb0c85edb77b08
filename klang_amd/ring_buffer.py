"""Circular sample store with shadow copies (the reference's klang/ring_buffer.py:10-122;
only the host ``Delay`` block uses it).

`length` live samples sit at the front of a store of `capacity + bufferSize` samples.
Whatever is written into the first `bufferSize` live positions is written again `length`
positions further back, and whatever runs past the live end is written again at the front.
A window of up to `bufferSize` samples that starts at the read / write position is therefore
always contiguous, and ``peek()`` can hand it out as a view.  The device's delay lines use
the same idea with mirror slots (klang_amd/csrc/mod_kernels.h, reverb_kernels.h).

Every write -- one sample or a block -- goes through ``_store``.
"""
import numpy as np

from klang_amd.config import BUFFER_SIZE
from klang_amd.constants import MONO

__all__ = ['RingBuffer']


class RingBuffer:
    def __init__(self, length, capacity=None, bufferSize=BUFFER_SIZE, nChannels=MONO, dtype=float):
        self._length = int(length)
        self.capacity = int(capacity) if capacity is not None else self._length
        self.bufferSize = bufferSize
        shape = (self.capacity + bufferSize,) + ((nChannels,) if nChannels > MONO else ())
        self.data = np.zeros(shape, dtype)
        self.pos = 0

    def _get_length(self):
        return self._length

    def _set_length(self, value):
        """A new length keeps the contents and folds the position into it."""
        if value > self.capacity:
            raise ValueError(f'New length {value} larger than max capacity {self.capacity}!')
        self._length, self.pos = value, self.pos % value

    length = property(_get_length, _set_length)

    def peek(self, nValues=None):
        """The next `nValues` (default: one buffer) samples that will be overwritten: a view."""
        span = self.bufferSize if nValues is None else nValues
        return self.data[self.pos:self.pos + span]

    def _store(self, values):
        """`values` at the current position, plus their shadow copies; advances the position."""
        live, window = self._length, self.bufferSize
        count = len(values)
        first, last = self.pos, self.pos + count
        self.data[first:last] = values
        if first < window:                       # head of the live part -> behind its end
            lo, hi = first + live, min(live + window, last + live)
            self.data[lo:hi] = values[:hi - lo]
        if last > live:                          # ran past the live end -> the front
            lo, hi = max(0, first - live), last - live
            self.data[lo:hi] = values[count - (hi - lo):]
        self.pos = last % live

    def append(self, value):
        """One sample."""
        self._store(np.asarray(value, dtype=self.data.dtype)[np.newaxis])

    def extend(self, values):
        if len(values) > self.bufferSize:
            raise ValueError('To many samples!')
        self._store(np.asarray(values))

    def peek_extend(self, values):
        """A delay line step: what `values` is about to overwrite (a copy), then the write."""
        delayed = np.array(self.peek(nValues=len(values)))
        self.extend(values)
        return delayed

    def __repr__(self):
        shown = f'length={self.length}'
        if self.capacity > self.length:
            shown += f', capacity={self.capacity}'
        return f'{type(self).__qualname__}({shown})'
