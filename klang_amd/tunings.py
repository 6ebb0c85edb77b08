"""Tuning temperaments (klang/music/tunings.py:33-69): twelve cent values, one per pitch
class, and a reference pitch.  A Note's default frequency comes from EQUAL_TEMPERAMENT, in
the reference's arithmetic (ratio table x base frequency x power of two), which is not
bit-identical to 440 * 2 ** ((pitch - 69) / 12)."""
import numpy as np

from klang_amd.config import KAMMERTON

DODE = 12                # pitch classes per octave (constants.py:14)
REF_OCTAVE = 5           # octave of the Kammerton in pitch numbers (constants.py:17)
CENT_PER_OCTAVE = 1200   # constants.py:35
KAMMERTON_OFFSET = 9     # pitch class of the Kammerton (tunings.py:26)

__all__ = ['Temperament', 'EQUAL_TEMPERAMENT', 'ratio_2_cent', 'cent_2_ratio']


def ratio_2_cent(ratio):
    return CENT_PER_OCTAVE * np.log2(ratio)


def cent_2_ratio(cent):
    return 2 ** (cent / CENT_PER_OCTAVE)


class Temperament:
    def __init__(self, cents, kammerton=KAMMERTON, name=''):
        assert len(cents) == DODE
        self.kammerton = kammerton
        self.name = name
        self.ratios = cent_2_ratio(np.asarray(cents))
        self.baseFrequency = kammerton / self.ratios[KAMMERTON_OFFSET]

    def pitch_2_frequency(self, pitch):
        octave, note = np.divmod(pitch, DODE)
        return self.baseFrequency * self.ratios[note] * (2. ** (octave - REF_OCTAVE))


EQUAL_TEMPERAMENT = Temperament(np.arange(DODE) * 100, kammerton=KAMMERTON, name='Equal')
