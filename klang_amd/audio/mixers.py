"""Mono and stereo mixers (the reference's klang/audio/mixer.py:15-104 and
klang/audio/panning.py:35-81), host side.

A mixer is a list of channels; each contributes ``weight * signal`` to the bus, and a bus
with more than one channel is divided by the channel count (mixer.py:46-47).  The two
mixers differ only in the weight: the channel's gain, or gain * (left, right) amplitudes of
its pan position.  The pan law itself is computed by the library's host helper
(klmi_panning_amplitudes), the same one the device banks are parameterised with.
"""
from klang_amd import native
from klang_amd.audio.signals import MONO_SILENCE, STEREO_SILENCE
from klang_amd.block import Block, fetch_output
from klang_amd.wiring import Input

__all__ = ['Mixer', 'StereoMixer', 'panning_amplitudes', 'LEFT', 'CENTER', 'RIGHT']

DEFAULT_GAIN = 1.
LEFT, CENTER, RIGHT = -1., 0., 1.


def panning_amplitudes(panLevel, mode='constant_power', panLaw=None):
    """(2, 1) array of left / right amplitudes; ValueError for an unknown mode or a positive
    pan law (panning.py:49-52)."""
    return native.panning_amplitudes(panLevel, mode, panLaw).reshape(2, 1)


def pan_signal(signal, panLevel, mode='constant_power', panLaw=None):
    """A mono signal -> stereo, or per-channel amplitudes on a stereo one (panning.py:84-99)."""
    return panning_amplitudes(panLevel, mode, panLaw) * signal


def _per_channel(values, count, default):
    values = count * [default] if values is None else values
    assert len(values) == count
    return values


class Mixer(Block):
    """Gain-weighted mean of its inputs (a plain weighted sum for a single input)."""

    silence = MONO_SILENCE

    def __init__(self, nInputs=0, gains=None):
        super().__init__(nInputs=nInputs, nOutputs=1)
        self.gains = _per_channel(gains, nInputs, DEFAULT_GAIN)

    def add_new_channel(self, gain=DEFAULT_GAIN):
        self.gains.append(gain)
        self.inputs.append(Input(owner=self))

    def weight(self, channel):
        """What channel number `channel` is multiplied with."""
        return self.gains[channel]

    def update(self):
        bus = self.silence.copy()
        for number, port in enumerate(self.inputs):
            bus += self.weight(number) * port.get_value()
        self.output.set_value(bus / len(self.inputs) if len(self.inputs) > 1 else bus)

    def attach(self, source):
        """`source` (a block or an output) on a new channel; the mixer itself is returned, so
        that ``mixer += a``, ``mixer + a`` and ``a + mixer`` all extend this mixer
        (mixer.py:51-61)."""
        self.add_new_channel()
        fetch_output(source).connect(self.inputs[-1])
        return self

    __iadd__ = __add__ = __radd__ = attach


class StereoMixer(Mixer):
    """Every channel is placed in the stereo field by its pan level (mixer.py:64-104)."""

    silence = STEREO_SILENCE

    def __init__(self, nInputs=0, gains=None, pannings=None, mode='constant_power',
                 panLaw=None):
        super().__init__(nInputs, gains=gains)
        self.pannings = _per_channel(pannings, nInputs, CENTER)
        self.mode, self.panLaw = mode, panLaw

    def add_new_channel(self, gain=DEFAULT_GAIN, panning=CENTER):
        super().add_new_channel(gain)
        self.pannings.append(panning)

    def weight(self, channel):
        return self.gains[channel] * panning_amplitudes(self.pannings[channel], self.mode,
                                                        self.panLaw)
