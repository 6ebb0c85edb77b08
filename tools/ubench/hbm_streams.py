#!/usr/bin/env python3
"""Achievable HBM rates on this box for the three stream shapes the kernels here produce:
write-only (materialised voice streams), copy (read + write, the reverb rings) and read-only.
torch ops on 1 GiB buffers, median of 10, GB/s."""
import time

import torch

n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device='cuda')
b = torch.empty(n, dtype=torch.float32, device='cuda')


def rate(fn, bytes_moved):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return bytes_moved / ts[len(ts) // 2] / 1e9


print('write-only (fill)   %7.0f GB/s' % rate(lambda: a.fill_(1.5), 4 * n))
print('copy (read + write) %7.0f GB/s' % rate(lambda: b.copy_(a), 8 * n))
print('read-only (sum)     %7.0f GB/s' % rate(lambda: a.sum(), 4 * n))
