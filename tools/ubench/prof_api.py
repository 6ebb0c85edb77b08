import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from klang_amd.audio import ADSR, Dac, Filter, Mixer, Oscillator, Voice, sawtooth
from klang_amd.klang import run_klang
from klang_amd.messages import Note
n = 32768
rng = np.random.default_rng(0)
mixer = Mixer()
for v in range(n):
    voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .1, .5, .2))
    fil = Filter(frequency=500. + 3000. * rng.random())
    voice | fil
    mixer += fil
    voice.input.push(Note(40 + v % 60, .5))
dac = mixer | Dac()
pr = cProfile.Profile()
pr.enable()
frames = run_klang(dac, offline=True, duration=2.0, filepath='/tmp/api.wav')
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
