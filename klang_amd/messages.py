"""Messages passed between blocks: the Note a Voice consumes
(klang/messages.py:8-78)."""
import collections
import json

from klang_amd.tunings import EQUAL_TEMPERAMENT


class Note(collections.namedtuple('Note', 'pitch velocity frequency')):
    """pitch (0..127), velocity (0..1; 0 is note-off), frequency in Hz."""

    def __new__(cls, pitch, velocity=1.0, frequency=None, temperament=EQUAL_TEMPERAMENT):
        if not 0 <= pitch < 128:
            raise ValueError('Invalid pitch %s!' % pitch)
        if not 0 <= velocity <= 1.:
            raise ValueError('Invalid velocity %s!' % velocity)
        if frequency is None:
            frequency = temperament.pitch_2_frequency(pitch)
        if frequency < 0:
            raise ValueError('Invalid frequency %s!' % frequency)
        return super().__new__(cls, pitch, velocity, frequency)

    @property
    def on(self):
        return self.velocity > 0.

    @property
    def off(self):
        return self.velocity == 0.

    def silence(self):
        return self._replace(velocity=0.)

    def to_dict(self):
        dct = dict(zip(self._fields, self))
        dct['type'] = type(self).__name__
        return dct

    @classmethod
    def from_dict(cls, dct):
        dct = dict(dct)
        if dct.pop('type') != cls.__name__:
            raise ValueError('Can not construct Note from dict %s' % dct)
        return cls(**dct)

    def to_json(self):
        return json.dumps(self.to_dict())

    @classmethod
    def from_json(cls, string):
        return cls.from_dict(json.loads(string))

    def __str__(self):
        return 'Note(pitch=%d, velocity=%.1f, frequency=%.1f Hz)' % self
