"""Circular sample store with a shadow copy (klang/ring_buffer.py:10-122).

`length` live samples sit at the front of a store of `capacity + bufferSize`
samples; whatever is written into the first `bufferSize` live positions is
written again `length` positions further back, and whatever runs past the live
end is written again at the front.  A window of up to `bufferSize` samples that
starts at the read/write position is therefore always contiguous, and
``peek()`` can hand it out without copying.  The device's delay lines use the
same idea with 8 mirror slots (klang_amd/csrc/mod_kernels.h).
"""
import numpy as np

from klang_amd.config import BUFFER_SIZE
from klang_amd.constants import MONO

__all__ = ['RingBuffer']


class RingBuffer:
    def __init__(self, length, capacity=None, bufferSize=BUFFER_SIZE, nChannels=MONO, dtype=float):
        self._length = int(length)
        self.capacity = self._length if capacity is None else int(capacity)
        self.bufferSize = bufferSize
        rows = self.capacity + bufferSize
        self.data = np.zeros((rows, nChannels) if nChannels > MONO else rows, dtype)
        self.pos = 0

    @property
    def length(self):
        return self._length

    @length.setter
    def length(self, value):
        """A new length keeps the contents and folds the position into it."""
        if value > self.capacity:
            raise ValueError(f'New length {value} larger than max capacity {self.capacity}!')
        self._length = value
        self.pos %= value

    def peek(self, nValues=None):
        """The next `nValues` (default: one buffer) samples that will be overwritten: a view."""
        count = self.bufferSize if nValues is None else nValues
        return self.data[self.pos:self.pos + count]

    def _write(self, values):
        """`values` at the current position, plus their shadow copies."""
        count = len(values)
        first, last = self.pos, self.pos + count
        self.data[first:last] = values
        if first < self.bufferSize:              # head of the live part -> behind its end
            lo = first + self._length
            hi = min(self._length + self.bufferSize, last + self._length)
            self.data[lo:hi] = values[:hi - lo]
        if last > self._length:                  # ran past the live end -> the front
            lo = max(0, first - self._length)
            hi = last - self._length
            self.data[lo:hi] = values[count - (hi - lo):]
        self.pos = last % self._length

    def append(self, value):
        """One sample.  (The reference mirrors to the front only for pos > length, which a
        position kept below length never reaches; same here.)"""
        self.data[self.pos] = value
        if self.pos < self.bufferSize:
            self.data[self.pos + self._length] = value
        self.pos = (self.pos + 1) % self._length

    def extend(self, values):
        if len(values) > self.bufferSize:
            raise ValueError('To many samples!')
        self._write(np.asarray(values))

    def peek_extend(self, values):
        """The samples about to be overwritten (a copy), then the write: a delay line."""
        old = self.peek(nValues=len(values)).copy()
        self.extend(values)
        return old

    def __repr__(self):
        infos = ['length=%d' % self.length]
        if self.length < self.capacity:
            infos.append('capacity=%d' % self.capacity)
        return '%s(%s)' % (type(self).__qualname__, ', '.join(infos))
