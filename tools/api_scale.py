#!/usr/bin/env python3
"""The whole drop-in path at scale: N Voice | Filter chains built with the Block API into
one Mixer | Dac, run_klang(offline, filepath=...).  Prints the time to build the Python
graph and the time of run_klang (scheduler, graph compiler, upload, device bounce, WAV).

    python tools/api_scale.py
"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from klang_amd.audio import ADSR, Dac, Filter, Mixer, Oscillator, Voice, sawtooth
from klang_amd.klang import run_klang
from klang_amd.messages import Note
for n in (1024, 8192, 65536):
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    mixer = Mixer()
    for v in range(n):
        voice = Voice(Oscillator(wave_func=sawtooth), ADSR(.01, .1, .5, .2))
        fil = Filter(frequency=500. + 3000. * rng.random())
        voice | fil
        mixer += fil
        voice.input.push(Note(40 + v % 60, .5))
    dac = mixer | Dac()
    t1 = time.perf_counter()
    frames = run_klang(dac, offline=True, duration=2.0, filepath='/tmp/api_%d.wav' % n)
    t2 = time.perf_counter()
    print(n, 'build %.2f s, run_klang(2 s of audio, WAV) %.2f s, peak %.3f' % (t1 - t0, t2 - t1, np.abs(np.concatenate(frames)).max()), flush=True)
