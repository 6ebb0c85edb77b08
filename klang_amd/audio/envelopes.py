"""Envelope generator blocks (klang/audio/envelopes.py:18-114) over a host
restatement of the reference's C envelope type (klang/audio/_envelope.c).

``Envelope`` below has the ext type's interface (constructor arguments,
``gate``, ``sample``, validated attributes, ``active``).  Its per-sample Python
loop only serves ``backend='host'`` / un-fused graphs; fused chains run the
same state machine on the GPU (klang_amd/csrc/dsp_device.h), batches of
envelopes through klmi_env_* (klang_amd.native.EnvelopeBatch).
"""
import math

import numpy as np

from klang_amd.audio.signals import DT
from klang_amd.block import Block
from klang_amd.config import BUFFER_SIZE
from klang_amd.wiring import MessageInput

__all__ = ['Envelope', 'ADSR', 'AR', 'D', 'R', 'DEFAULT_OVERSHOOT']

DEFAULT_OVERSHOOT = 1e-3
OFF, ATTACKING, DECAYING, SUSTAINING, RELEASING = range(5)


def exponential_coefficient(rate, overshoot):
    """_envelope.c:112-119; `rate` in samples."""
    if rate <= 0.:
        return 0.
    return math.exp(-math.log((1.0 + overshoot) / overshoot) / rate)


class Envelope:
    """One-pole ADSR state machine (_envelope.c:125-663)."""

    def __init__(self, attack, decay, sustain, release, dt, overshoot=DEFAULT_OVERSHOOT,
                 retrigger=False, loop=False):
        # like _envelope.c:518-535 the constructor stores without validating
        self._attack, self._decay = float(attack), float(decay)
        self._sustain, self._release = float(sustain), float(release)
        self.dt = float(dt)
        self._overshoot = min(max(float(overshoot), 1e-9), 1e9)
        self.retrigger = bool(retrigger)
        self.loop = bool(loop)
        self.stage = OFF
        self.value = 0.
        self.enabled = False
        self._recompute()

    def _recompute(self):
        """_envelope.c:187-198."""
        o = self._overshoot
        self.attackCoef = exponential_coefficient(self._attack / self.dt, o)
        self.attackBase = (1. + o) * (1. - self.attackCoef)
        self.decayCoef = exponential_coefficient(self._decay / self.dt, o)
        self.decayBase = (self._sustain - o) * (1. - self.decayCoef)
        self.releaseCoef = exponential_coefficient(self._release / self.dt, o)
        self.releaseBase = (0. - o) * (1. - self.releaseCoef)

    def _timed(name):  # noqa: N805  (property factory)
        def getter(self):
            return getattr(self, '_' + name)

        def setter(self, value):
            value = float(value)
            if value < 0:
                raise ValueError('%s must be positive!' % name)   # _envelope.c:229-232
            setattr(self, '_' + name, value)
            self._recompute()
        return property(getter, setter)

    attack = _timed('attack')
    decay = _timed('decay')
    release = _timed('release')
    del _timed

    @property
    def sustain(self):
        return self._sustain

    @sustain.setter
    def sustain(self, value):
        value = float(value)
        if not 0. <= value <= 1.:
            raise ValueError('sustain not within bounds!')         # _envelope.c:299-302
        self._sustain = value
        self._recompute()

    @property
    def overshoot(self):
        return self._overshoot

    @overshoot.setter
    def overshoot(self, value):
        value = float(value)
        if value < 0:
            raise ValueError('overshoot must be positive!')
        self._overshoot = min(max(value, 1e-9), 1e9)               # _envelope.c:374
        self._recompute()

    @property
    def active(self):
        return self.stage != OFF

    def gate(self, trigger):
        """_envelope.c:549-567."""
        if trigger:
            self.enabled = True
            if self.retrigger or self.stage in (OFF, RELEASING):
                self.stage = ATTACKING
        else:
            self.enabled = False
            if self.stage in (ATTACKING, DECAYING, SUSTAINING):
                self.stage = RELEASING

    def sample(self, bufferSize):
        """_envelope.c:579-663."""
        n = int(bufferSize)
        if n < 0:
            raise ValueError('Could not cast *bufferSize to size_t!')
        out = np.empty(n)
        stage, value = self.stage, self.value
        sustain = self._sustain
        for i in range(n):
            if stage == ATTACKING:
                value = self.attackBase + value * self.attackCoef
                if value >= 1.:
                    value, stage = 1., DECAYING
            elif stage == DECAYING:
                value = self.decayBase + value * self.decayCoef
                if value <= sustain:
                    value, stage = sustain, (RELEASING if self.loop else SUSTAINING)
            elif stage == RELEASING:
                value = self.releaseBase + value * self.releaseCoef
                if value <= 0.:
                    value = 0.
                    stage = ATTACKING if (self.loop and self.enabled) else OFF
            elif stage == SUSTAINING:
                value = sustain
            else:
                value = 0.
            out[i] = value
        self.stage, self.value = stage, value
        return out


class EnvelopeBase(Block, Envelope):
    """Envelope as a block with a trigger message input (envelopes.py:18-59)."""

    def __init__(self, attack, decay, sustain, release, overshoot=DEFAULT_OVERSHOOT,
                 retrigger=False, loop=False):
        Block.__init__(self, nOutputs=1)
        self.inputs = self.trigger, = [MessageInput(owner=self)]
        Envelope.__init__(self, attack, decay, sustain, release, DT, overshoot, retrigger, loop)

    @property
    def current_level(self):
        return self.output.get_value()[-1]

    def update(self):
        for note in self.input.receive():
            self.gate(note.on)
        self.output.set_value(self.sample(BUFFER_SIZE))

    def _ctor_args(self):
        return (self.attack, self.decay, self.sustain, self.release)

    def __deepcopy__(self, memo):
        return type(self)(*self._ctor_args(), overshoot=self.overshoot, retrigger=self.retrigger,
                          loop=self.loop)

    def __str__(self):
        bits = ['%s=%s' % (k, getattr(self, k))
                for k in ('attack', 'decay', 'sustain', 'release', 'retrigger', 'loop')
                if getattr(self, k)]
        return '%s(%s)' % (type(self).__name__, ', '.join(bits))


class ADSR(EnvelopeBase):
    def __init__(self, attack=.1, decay=.2, sustain=.8, release=1., *args, **kwargs):
        super().__init__(attack, decay, sustain, release, *args, **kwargs)


class AR(EnvelopeBase):
    """Attack-release; sustain fixed at 1 (envelopes.py:70-83)."""

    def __init__(self, attack=.1, release=1., *args, **kwargs):
        super().__init__(attack, 0., 1., release, *args, **kwargs)

    def _ctor_args(self):
        return (self.attack, self.release)


class D(EnvelopeBase):
    """Decay only (envelopes.py:86-101)."""

    def __init__(self, decay=1., *args, **kwargs):
        super().__init__(0., decay, 0., decay, *args, **kwargs)

    def _ctor_args(self):
        return (self.decay,)


class R(EnvelopeBase):
    """Release only (envelopes.py:104-114)."""

    def __init__(self, release=1., *args, **kwargs):
        super().__init__(0., 0., 1., release, *args, **kwargs)

    def _ctor_args(self):
        return (self.release,)
