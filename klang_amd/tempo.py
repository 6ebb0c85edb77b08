"""Tempo-aware rates and durations (klang/music/tempo.py:19-108): a Fraction is a
note value relative to TEMPO / METRE, anything else passes through."""
from fractions import Fraction

from klang_amd.config import METRE, TEMPO


def default_beat_value(metre):
    """klang/music/metre.py: the beat value of a metre is 1 / denominator."""
    return Fraction(1, metre.denominator)


def note_duration(note, tempo=TEMPO, metre=METRE, beatValue=None):
    if beatValue is None:
        beatValue = default_beat_value(metre)
    beat_period = 1. / (tempo / 60.)
    return beat_period / beatValue * note


def compute_duration(duration, *args, **kwargs):
    if isinstance(duration, Fraction):
        return note_duration(duration, *args, **kwargs)
    return duration


def compute_rate(frequency, *args, **kwargs):
    if isinstance(frequency, Fraction):
        return 1. / note_duration(frequency, *args, **kwargs)
    return frequency
