"""Messages passed between blocks: the Note a Voice consumes (the reference's
klang/messages.py:8-78): an immutable (pitch, velocity, frequency) triple; velocity 0 is a
note-off; the frequency defaults to the pitch's equal-temperament frequency."""
import collections
import json

from klang_amd.tunings import EQUAL_TEMPERAMENT

_Triple = collections.namedtuple('Note', ('pitch', 'velocity', 'frequency'))


def _valid(what, value, ok):
    if not ok:
        raise ValueError('Invalid %s %s!' % (what, value))


class Note(_Triple):
    """pitch (0..127), velocity (0..1; 0 is note-off), frequency in Hz."""
    __slots__ = ()
    TAG = 'Note'   # the 'type' entry of the dict / JSON form

    def __new__(cls, pitch, velocity=1.0, frequency=None, temperament=EQUAL_TEMPERAMENT):
        _valid('pitch', pitch, 0 <= pitch < 128)
        _valid('velocity', velocity, 0 <= velocity <= 1.)
        hertz = temperament.pitch_2_frequency(pitch) if frequency is None else frequency
        _valid('frequency', hertz, not hertz < 0)
        return _Triple.__new__(cls, pitch, velocity, hertz)

    on = property(lambda self: self.velocity > 0.)
    off = property(lambda self: self.velocity == 0.)

    def silence(self):
        """The note-off that ends this note."""
        return self._replace(velocity=0.)

    # dict / JSON forms (network messages of the reference)
    def to_dict(self):
        return dict(self._asdict(), type=self.TAG)

    @classmethod
    def from_dict(cls, dct):
        fields = {key: val for key, val in dct.items() if key != 'type'}
        if dct.get('type') != cls.TAG:
            raise ValueError('Can not construct Note from dict %s' % fields)
        return cls(**fields)

    def to_json(self):
        return json.dumps(self.to_dict())

    @classmethod
    def from_json(cls, string):
        return cls.from_dict(json.loads(string))

    def __str__(self):
        return 'Note(pitch={:d}, velocity={:.1f}, frequency={:.1f} Hz)'.format(*self)
