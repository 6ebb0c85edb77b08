"""Element-wise and delay effects (host side): ``Gain``, ``Transformer``,
``TanhDistortion``, ``OctaveDistortion``, ``Tremolo``, ``Delay``, ``StereoDelay`` (with
``AudioSplitter`` / ``AudioCombiner``), ``Subsampler``, ``Bitcrusher``, ``RingModulator``.

Mirrors klang/audio/effects.py:33-236, 388-449, 452-561, 618-662 and
klang/ring_buffer.py:10-113.
The ``update()`` methods are the ``backend='host'`` evaluation; inside a fused
voice chain the same stages run on the device (mod_kernels.h).
"""
import functools
import math

import numpy as np

from klang_amd import native
from klang_amd.audio.filtering import blend
from klang_amd.audio.signals import Oscillator, PwmOscillator, square
from klang_amd.block import Block
from klang_amd.composite import Composite
from klang_amd.config import BUFFER_SIZE, SAMPLING_RATE
from klang_amd.ring_buffer import RingBuffer
from klang_amd.constants import MONO, STEREO
from klang_amd.tempo import compute_duration
from klang_amd.wiring import Input, Relay

__all__ = ['Gain', 'Transformer', 'TanhDistortion', 'OctaveDistortion', 'Tremolo', 'Delay',
           'DelayLine', 'AudioSplitter', 'AudioCombiner', 'StereoDelay', 'Subsampler', 'Bitcrusher', 'RingModulator', 'low_pass_coefficients']


def _value_ports(owner, *spec):
    """inputs = [signal input] + one value port per (kind, initial value); -> the value ports."""
    ports = [kind(owner=owner) for kind, _ in spec]
    for port, (_, initial) in zip(ports, spec):
        port.set_value(initial)
    owner.inputs = [Input(owner=owner)] + ports
    return ports


@functools.lru_cache()
def low_pass_coefficients(frequency):
    """Single-pole low-pass (b, a) (effects.py:33-55), from the library's helper."""
    return native.low_pass_coefficients(float(frequency))


class Gain(Block):
    """gain * samples, the gain being a value input (effects.py:58-70)."""

    def __init__(self, gain=1.):
        super().__init__(nInputs=2, nOutputs=1)
        _, self.gain = self.inputs
        self.gain.set_value(gain)

    def update(self):
        self.output.set_value(self.gain.value * self.input.value)


class Transformer(Block):
    """scale * samples + offset (effects.py:522-561)."""

    def __init__(self, scale, offset):
        super().__init__(nInputs=1, nOutputs=1)
        self.scale, self.offset = scale, offset

    @classmethod
    def from_limits(cls, lower, upper):
        """Input range [0, 1] -> [lower, upper]."""
        assert lower < upper
        return cls(scale=upper - lower, offset=lower)

    @classmethod
    def from_limits_2(cls, lower, upper):
        """Input range [-1, 1] -> [lower, upper]."""
        assert lower < upper
        width = upper - lower
        return cls(scale=.5 * width, offset=lower + .5 * width)

    @classmethod
    def from_ranges(cls, xRange=(0., 1.), yRange=(0., 1.)):
        from klang_amd.audio.signals import linear_mapping
        return cls(*linear_mapping(xRange, yRange))

    def update(self):
        self.output.set_value(self.scale * self.input.value + self.offset)


class TanhDistortion(Block):
    """tanh(drive * samples) (effects.py:469-487)."""

    def __init__(self, drive=1.):
        super().__init__(nInputs=1, nOutputs=1)
        self.drive = drive

    def update(self):
        self.output.set_value(np.tanh(self.drive * self.input.get_value()))


class OctaveDistortion(Block):
    """2 * samples**2 - 1 (effects.py:452-466)."""

    def __init__(self):
        super().__init__(nInputs=1, nOutputs=1)

    def update(self):
        samples = self.input.get_value()
        self.output.set_value(2. * samples ** 2 - 1.)


class Tremolo(Composite):
    """Amplitude modulation by a low-passed pulse Lfo (effects.py:73-133):
    env = (1 - depth) + depth * (.5 + .5 * lowpass(pwm)), out = env * x."""

    def __init__(self, rate=3., depth=.8, smoothness=1., dutyCycle=.5):
        super().__init__(nOutputs=1)
        # rate and duty cycle are relays: they pass on to the pulse Lfo inside
        self.rate, self.depth, self.smoothness, self.dutyCycle = _value_ports(
            self, (Relay, rate), (Input, depth), (Input, smoothness), (Relay, dutyCycle))
        self.lfo = PwmOscillator(rate)
        for relay, target in ((self.rate, self.lfo.frequency), (self.dutyCycle, self.lfo.dutyCycle)):
            relay.connect(target)
        self.zi = np.zeros(1)      # state of the smoothing filter

    @staticmethod
    def smoothing_coefficients(smoothness, dutyCycle):
        decay = max(1e-6, smoothness * min(dutyCycle, 1. - dutyCycle))
        return low_pass_coefficients(1. / decay)

    def update(self):
        (b0,), (_, a1) = self.smoothing_coefficients(self.smoothness.value, self.dutyCycle.value)
        self.lfo.update()
        pulse = self.lfo.output.value
        # scipy.signal.lfilter([b0], [1, a1], pulse, zi): direct form II transposed, order 1
        smooth = np.empty(BUFFER_SIZE)
        z = float(self.zi[0])
        for i, p in enumerate(pulse):
            smooth[i] = y = z + b0 * p
            z = 0. * p - a1 * y
        self.zi = np.array([z])
        depth = self.depth.value
        self.output.set_value(((1. - depth) + depth * (.5 + .5 * smooth)) * self.input.value)


DelayLine = RingBuffer   # the Delay's line is the reference's RingBuffer (effects.py:158)


class Delay(Block):
    """Mono feedback delay (effects.py:136-174).  The dry/wet blend reads the
    ring window AFTER the new samples were written into it (the window is a
    view), so the wet part is new + feedback * delayed, as in the reference."""
    MAX_TIME = 2.

    def __init__(self, time=1., feedback=.1, drywet=.5):
        if time > self.MAX_TIME:
            raise ValueError('Delay time %.3f sec to long (max %.3f)!' % (time, self.MAX_TIME))
        super().__init__(nInputs=1, nOutputs=1)
        self.feedback, self.drywet = feedback, drywet
        self.time = compute_duration(time)
        self.ring = DelayLine(int(self.time * SAMPLING_RATE), int(self.MAX_TIME * SAMPLING_RATE))

    def update(self):
        dry = self.input.get_value()
        if np.ndim(dry) != MONO:
            dry = dry.sum(axis=0) / dry.shape[0]
        window = self.ring.peek()       # a view of the line: it shows what extend() writes next
        self.ring.extend(dry + self.feedback * window)
        self.output.set_value(blend(dry, window, self.drywet))


class AudioSplitter(Block):
    """One multi-channel input -> one mono output per channel; a mono input goes to every
    output as it is (effects.py:177-192)."""

    def __init__(self, nOutputs):
        super().__init__(nInputs=1, nOutputs=nOutputs)

    def update(self):
        signal = self.input.value
        rows = [signal] * len(self.outputs) if np.ndim(signal) == MONO else signal
        for port, row in zip(self.outputs, rows):
            port.set_value(row)


class AudioCombiner(Block):
    """Mono inputs -> the rows of one multi-channel output, written in place
    (effects.py:195-207)."""

    def __init__(self, nInputs):
        super().__init__(nInputs, nOutputs=1)
        self.output.set_value(np.zeros((nInputs, BUFFER_SIZE)))

    def update(self):
        for input_, row in zip(self.inputs, self.output.value):
            row[:] = input_.value


class StereoDelay(Composite):
    """Two Delays, one per channel, between a splitter and a combiner (effects.py:210-236).
    Under backend='hip' the graph compiler opens the composite up and each Delay becomes a
    device chain fed from its splitter output."""

    def __init__(self, leftTime=1., rightTime=1., leftFeedback=.1, rightFeedback=.1, drywet=.5):
        super().__init__()
        self.inputs, self.outputs = [Relay(owner=self)], [Relay(owner=self)]
        self.leftDelay = Delay(leftTime, leftFeedback, drywet)
        self.rightDelay = Delay(rightTime, rightFeedback, drywet)
        split, join = AudioSplitter(nOutputs=STEREO), AudioCombiner(nInputs=STEREO)
        self.input.connect(split.input)
        for tap, delay, slot in zip(split.outputs, (self.leftDelay, self.rightDelay), join.inputs):
            tap | delay | slot
        join.output.connect(self.output)
        self.update_internal_exec_order()


class Subsampler(Block):
    """Repeat every `factor`-th sample `factor` times (effects.py:388-417)."""

    VALID_FACTORS = set(2 ** i for i in range(1, int(math.log2(BUFFER_SIZE))))

    def __init__(self, factor):
        assert factor in self.VALID_FACTORS
        super().__init__(nInputs=1, nOutputs=1)
        self.factor = factor

    @staticmethod
    def sub_sample(array, skip):
        return np.repeat(array[..., ::skip], skip, axis=-1)

    def update(self):
        self.output.set_value(self.sub_sample(self.input.get_value(), self.factor))


class Bitcrusher(Block):
    """Drop the `nBits` lowest bits of the 16-bit sample (effects.py:420-449 with
    wavfile.py:14-30): float -> int16 by truncation, arithmetic shifts, / 32768."""

    def __init__(self, nBits=16):
        assert 0 <= nBits < 16
        super().__init__(nInputs=1, nOutputs=1)
        self.nBits = nBits

    @staticmethod
    def bit_crush(samples, nBits):
        ints = (np.iinfo(np.int16).max * np.asarray(samples)).astype(np.int16)
        return ((ints >> nBits) << nBits) / abs(np.iinfo(np.int16).min)

    def update(self):
        self.output.set_value(self.bit_crush(self.input.get_value(), self.nBits))


class RingModulator(Composite):
    """Amplitude modulation by a sine whose frequency an Lfo sweeps
    (effects.py:618-662): out = blend(x, sin-modulator * x, dryWet)."""

    def __init__(self, frequency=500., rate=2., amount=20, wave_func=square, dryWet=1.):
        super().__init__()
        self.frequency, self.rate, self.amount, self.dryWet = _value_ports(
            self, (Input, frequency), (Relay, rate), (Input, amount), (Input, dryWet))
        self.outputs = [Relay(owner=self)]
        self.lfo, self.modulator = Oscillator(wave_func=wave_func), Oscillator()
        self.rate.connect(self.lfo.frequency)

    def update(self):
        self.lfo.update()       # sweeps the sine modulator's frequency, sample by sample
        sweep = self.frequency.value + self.amount.value * self.lfo.output.value
        self.modulator.frequency.set_value(sweep)
        self.modulator.update()
        dry = self.input.value
        self.output.set_value(blend(dry, self.modulator.output.value * dry, self.dryWet.value))
