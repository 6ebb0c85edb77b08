"""Experiment: materialised mode (klmi_bank_render_voices_device) against the row pitch of the output."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from klang_amd import native, patches

ctx = native.Context()
B = 256
for n, nbs in ((65536, (256, 255, 250, 192, 128)), (131072, (128, 125, 96))):
    for nb in nbs:
        patch = patches.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=10, off_span=20)
        bank = native.Bank(ctx, n, has_env=True, has_filter=True)
        bank.load_patch(patch)
        buf = torch.empty((n, nb * B), dtype=torch.float32, device='cuda')
        ms = []
        for _ in range(4):
            bank.render_voices_device(nb, buf.data_ptr())
            ctx.synchronize()
            ms.append(bank.last_kernel_ms()[0])
        m = sorted(ms[1:])[1]
        print('%d voices x %d blocks: pitch %d B  kernel %.3f ms  %.0f GB/s' % (n, nb, nb * B * 4, m, n * nb * B * 4.594 / (m * 1e-3) / 1e9), flush=True)
        del buf
        bank.close()
