"""Audio in/out blocks and the offline audio engine
(klang/audio/klanggeber.py:81-300, klang/audio/wavfile.py:54-63).

Only the offline path exists here: the PortAudio real-time callback of the
reference needs a sound card and is outside the DSP hot path.
"""
import logging
import wave

import numpy as np

from klang_amd.audio.signals import INTERVAL, get_silence, get_time
from klang_amd.block import Block
from klang_amd.clock import ClockMixin
from klang_amd.config import BUFFER_SIZE, SAMPLING_RATE
from klang_amd.constants import INF, MONO
from klang_amd.errors import KlangError

__all__ = ['Adc', 'Dac', 'ChannelMismatch', 'run_audio_engine', 'write_wave', 'write_pcm',
           'pack_signals']

D_TYPE = np.float32


class ChannelMismatch(KlangError):
    """Received audio does not match the Dac's channel count."""


def pack_signals(signals):
    """Interleave mono signals to (BUFFER_SIZE, nChannels) float32."""
    out = np.empty((BUFFER_SIZE, len(signals)), dtype=D_TYPE)
    for col, samples in enumerate(signals):
        out[:, col] = samples
    return out


class Adc(Block):
    def __init__(self, nChannels):
        super().__init__(nOutputs=nChannels)
        self.nChannels = nChannels
        for output in self.outputs:
            output.set_value(get_silence(BUFFER_SIZE))

    def inject_samples(self, samples):
        for signal, output in zip(samples.T, self.outputs):
            output.set_value(signal)


class Dac(Block):
    def __init__(self, nChannels=MONO):
        super().__init__(nInputs=nChannels)
        self.nChannels = nChannels
        for input_ in self.inputs:
            input_.set_value(get_silence(BUFFER_SIZE))

    def iterate_channels(self):
        """Mono values count as one channel, 2-D values as one per row, until
        nChannels are collected (klanggeber.py:141-162)."""
        taken = 0
        for input_ in self.inputs:
            val = input_.get_value()
            rows = [val] if np.ndim(val) == MONO else (list(val) if np.ndim(val) > MONO else [])
            for row in rows:
                yield row
                taken += 1
                if taken == self.nChannels:
                    return

    def collect_samples(self):
        signals = list(self.iterate_channels())
        if len(signals) != self.nChannels:
            raise ChannelMismatch('Got %d channels. Need %d!' % (len(signals), self.nChannels))
        return pack_signals(signals)


def look_for_audio_blocks(blocks):
    """First Adc / Dac of the top-level order, or empty stand-ins."""
    adc = next((b for b in blocks if isinstance(b, Adc)), None) or Adc(nChannels=0)
    dac = next((b for b in blocks if isinstance(b, Dac)), None) or Dac(nChannels=0)
    return adc, dac


def write_wave(audio, filepath, samplingRate=SAMPLING_RATE, dtype=np.int16):
    """Peak-normalised PCM WAV (wavfile.py:54-63).  ValueError on silence, as
    klang/math.py:14-21 normalize_values."""
    audio = np.asarray(audio)
    peak = np.abs(audio).max()
    if peak == 0:
        raise ValueError('Zero amplitude! Can not normalize!')
    samples = (audio / peak * np.iinfo(dtype).max).astype(dtype)
    write_pcm(samples, filepath, samplingRate)


def write_pcm(samples, filepath, samplingRate=SAMPLING_RATE):
    """(frames, channels) int16 -> PCM WAV file (the container part of wavfile.py:54-63;
    a device bounce delivers the normalised int16 samples itself)."""
    samples = np.asarray(samples)
    frames, channels = samples.shape
    assert frames > channels
    with wave.open(filepath, 'wb') as out:
        out.setnchannels(channels)
        out.setsampwidth(samples.dtype.itemsize)
        out.setframerate(samplingRate)
        out.writeframes(samples.astype('<i2').tobytes())


def run_audio_engine(adc, dac, execute_all_blocks, duration=INF, fadeout=0., filepath='',
                     offline=False, capture=None):
    """Offline bounce: nCycles = int(duration * SR // BUFFER_SIZE) buffers
    (klanggeber.py:255-263), float32 packing, optional fade-out, WAV dump.
    `capture`: optional list that receives the packed float32 buffers."""
    logger = logging.getLogger('KlangGeber')
    if not offline:
        raise KlangError('real-time playback (PortAudio) is not part of klang_amd; '
                         'use offline=True with a finite duration')
    if duration == INF:
        raise KlangError('Offline mode and duration set to INF. You will never get anything.')
    fade_start = duration - fadeout
    t_block = get_time(BUFFER_SIZE)
    frames = capture if capture is not None else []
    play_time = 0.
    n_cycles = int(duration * SAMPLING_RATE // BUFFER_SIZE)
    for _ in range(n_cycles):
        ClockMixin.set_current_time(play_time)
        execute_all_blocks()
        out = dac.collect_samples()
        if play_time >= fade_start:
            env = np.interp(play_time + t_block, [fade_start, duration], [1., 0.], left=1., right=0.)
            out *= env.reshape(-1, 1)
        frames.append(out)
        play_time += INTERVAL
    if filepath:
        logger.info('Writing audio dump to WAV file %r', filepath)
        write_wave(np.concatenate(frames), filepath)
    return frames
