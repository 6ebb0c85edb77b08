"""Fixed cost per launch of the fused voice kernel: config-3 patch, 65,536 voices, rendered
n blocks per call for several n; kernel time from the library's HIP events.
Usage (GPU box): PYTHONPATH=. python tools/ubench/launch_fixed_cost.py"""
import numpy as np
import torch

from klang_amd import native, patches

torch.cuda.init()
ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
N = 65536
for nb in (16, 64, 128, 256):
    patch = patches.polyphonic_saw_patch(n_voices=N, n_blocks=nb, off_base=2, off_span=max(1, nb // 2))
    bank = native.Bank(ctx, N, has_env=True, has_filter=True)
    ev = patch.events
    p = patches.Patch(patch)
    p['events'] = ev[:0]
    bank.load_patch(p)
    reps = 6
    allev = []
    for r in range(reps):
        e = ev.copy()
        e['block'] += r * nb
        allev.append(e)
    bank.events(np.concatenate(allev))
    ms = []
    for r in range(reps):
        bank.render(nb)
        k, launches = bank.last_kernel_ms()
        ms.append(k)
    print('blocks/launch %4d: kernel %.3f ms (min %.3f) -> %.3e voice-samples/s, launches %d'
          % (nb, float(np.median(ms[1:])), min(ms[1:]), N * nb * 256 / (np.median(ms[1:]) * 1e-3), launches),
          flush=True)
    bank.close()
