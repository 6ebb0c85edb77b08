"""Tempo-aware rates and durations (klang/music/tempo.py:19-108), the metre helpers they rest
on (klang/music/metre.py:23-136) and the note values (klang/music/note_values.py:40-51): a
Fraction is a note value relative to TEMPO / METRE, anything else passes through."""
from fractions import Fraction

from klang_amd.config import METRE, TEMPO
from klang_amd.constants import TAU

# note_values.py:40-90: LARGE_NOTE (8 whole notes) ... TWO_HUNDRED_FIFTY_SIXTH_NOTE, each also
# DOTTED_ (x 3/2), DOUBLE_DOTTED_ (x 7/4) and TRIPLE_DOTTED_ (x 15/8)
_NAMES = ('LARGE', 'LONG', 'DOUBLE_WHOLE', 'WHOLE', 'HALF', 'QUARTER', 'EIGHT', 'SIXTEENTH',
          'THIRTY_SECOND', 'SIXTY_FOURTH', 'HUNDRED_TWENTY', 'TWO_HUNDRED_FIFTY_SIXTH')
NOTE_VALUES = {}
for _k, _name in enumerate(_NAMES):
    _base = Fraction(8, 1) / 2 ** _k
    for _prefix, _factor in (('', Fraction(1)), ('DOTTED_', Fraction(3, 2)),
                             ('DOUBLE_DOTTED_', Fraction(7, 4)), ('TRIPLE_DOTTED_', Fraction(15, 8))):
        NOTE_VALUES['%s%s_NOTE' % (_prefix, _name)] = _base * _factor
globals().update(NOTE_VALUES)


def dot_note(note, n=1):
    """A note value with n dots: note * (2 - 1 / 2**n) (note_values.py:93-101)."""
    return note * (2 - Fraction(1, 2 ** n))


def create_metre(numerator, denominator=None):
    """A time signature: a Fraction that is NOT reduced, so 6/8 keeps its 8 (metre.py:23-37)."""
    return Fraction(numerator, denominator, _normalize=False)


FOUR_FOUR_METRE = create_metre(4, 4)
THREE_FOUR_METRE = create_metre(3, 4)
SIX_EIGHT_METRE = create_metre(6, 8)


SEVEN_EIGHT_METRE = create_metre(7, 8)


def is_irrational(metre):
    """The denominator is no power of two (metre.py:55-61)."""
    return metre.denominator & (metre.denominator - 1) != 0


def is_simple(metre):
    """Two, three or four beats (metre.py:87-88)."""
    return metre.numerator in (2, 3, 4)


def number_of_beats(metre, beatValue=None):
    """metre.py:130-136."""
    if beatValue is None:
        beatValue = default_beat_value(metre)
    return metre / beatValue


def is_compound(metre):
    """6/8, 9/8, 12/8 ... (metre.py:80-84)."""
    return metre.numerator >= 6 and metre.numerator % 3 == 0


def is_complex(metre):
    """Numerator divisible by neither 2 nor 3: 5/4, 7/8 ... (metre.py:70-77)."""
    return metre.numerator % 2 != 0 and metre.numerator % 3 != 0


def default_beat_value(metre):
    """Best-guess beat of a metre: three denominators for a compound metre, two for a complex
    one, else one (metre.py:96-127)."""
    if is_compound(metre):
        return create_metre(3, metre.denominator)
    if is_complex(metre):
        return create_metre(2, metre.denominator)
    return create_metre(1, metre.denominator)


def tempo_2_frequency(tempo):
    """Beats per minute -> beats per second (tempo.py:19-29)."""
    return tempo / 60.


def tempo_2_period(tempo):
    """Beats per minute -> seconds per beat (tempo.py:32-41)."""
    return 1. / tempo_2_frequency(tempo)


def bar_period(tempo, metre=METRE, beatValue=None):
    """Seconds per bar (tempo.py:44-61)."""
    if beatValue is None:
        beatValue = default_beat_value(metre)
    return metre / tempo_2_frequency(tempo) / beatValue


def angular_velocity(tempo, metre=METRE, beatValue=None):
    """Bar phase per second (tempo.py:64-67)."""
    return TAU / bar_period(tempo, metre, beatValue)


def note_duration(note, tempo=TEMPO, metre=METRE, beatValue=None):
    """Seconds a note value lasts (tempo.py:70-90)."""
    if beatValue is None:
        beatValue = default_beat_value(metre)
    beat_period = 1. / tempo_2_frequency(tempo)
    return beat_period / beatValue * note


def compute_duration(duration, *args, **kwargs):
    if isinstance(duration, Fraction):
        return note_duration(duration, *args, **kwargs)
    return duration


def compute_rate(frequency, *args, **kwargs):
    if isinstance(frequency, Fraction):
        return 1. / note_duration(frequency, *args, **kwargs)
    return frequency
