"""Global parameters the DSP path is compiled against (klang/config.py:5-18)."""
from fractions import Fraction

BUFFER_SIZE = 256
KAMMERTON = 440.
SAMPLING_RATE = 44100
TEMPO = 120.
METRE = Fraction(4, 4, _normalize=False)
