import torch, numpy as np
torch.cuda.init()
from klang_amd import native, patches
from klang_amd.sharding import ShardedMixer
ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
n = 65536
patch = patches.polyphonic_saw_patch(n_voices=n, n_blocks=512, first_voice=0, count=n)
bank = native.Bank(ctx, n, has_env=True, has_filter=True)
ev = patch.events
p2 = patches.Patch(patch); p2['events'] = ev[:0]
bank.load_patch(p2)
bank.events(ev)
mixer = ShardedMixer(bank, n, 512, 1, 0)
host_bus = torch.empty((512, 1, 256), dtype=torch.float32).pin_memory()
bus = mixer.render()
pass
host_bus.copy_(bus, non_blocking=True)
print(bank.last_kernel_ms())
torch.cuda.synchronize()
print('host peak', float(host_bus.abs().max()))
bank.close()
ctx.close()
