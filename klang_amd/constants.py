"""Constants (klang/constants.py:5-39)."""
import math

PI = math.pi
TAU = 2. * math.pi
INF = float('inf')
MONO = 1
STEREO = 2
SCALAR = 1
