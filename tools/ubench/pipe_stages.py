#!/usr/bin/env python3
"""Which stage bounds the pipelined voice kernel on a tiny bank?  20 voices x 2,048 buffers
with parts of the chain switched off.  Usage: python tools/ubench/pipe_stages.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from klang_amd import native, patches  # noqa: E402

import torch  # noqa: E402,F401
torch.cuda.init()
ctx = native.Context(0)
nb = 2048
for label, kw in (('sine, looping AR, stereo (config 2 native)', {}),
                  ('sine, no envelope, stereo', dict(has_env=False)),
                  ('constant 1, looping AR, stereo', dict(wave=native.WAVE_ONE)),
                  ('sine, looping AR, mono', dict(stereo=False)),
                  ('saw, ADSR, biquad, mono', dict(stereo=False, has_filter=True, wave=native.WAVE_SAWTOOTH))):
    p = patches.haunting_patch(1, nb)
    has_env = kw.get('has_env', True)
    bank = native.Bank(ctx, p.n_voices, has_env=has_env, has_filter=kw.get('has_filter', False),
                       stereo=kw.get('stereo', True))
    bank.load_patch(p)
    if 'wave' in kw:
        bank.set(native.F_WAVE, np.full(p.n_voices, kw['wave']))
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        bank.render(nb)
        ts.append(time.perf_counter() - t0)
    ms, launches = bank.last_kernel_ms()
    print('%-45s wall %.2f ms  kernel %.2f ms  (%.0f ns per 32-sample step)' %
          (label, min(ts) * 1e3, ms, ms * 1e6 / (nb * 8)))
    bank.close()
ctx.close()
