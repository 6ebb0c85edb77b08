#!/usr/bin/env python3
"""bench.py -- voice-samples/sec of the fused osc -> ADSR -> biquad -> Mixer path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N == 1: BASELINE.json configs[2], the synthetic 65,536-voice polyphonic patch
(saw Oscillator -> ADSR -> Butterworth-2 lowpass -> Mixer, 44.1 kHz, 256-sample
blocks), 512 blocks per step.  N > 1 (launched by torch.distributed.run, one
rank per GPU): every rank renders its own contiguous 65,536-voice shard of the
N*65,536-voice patch (weak scaling, configs[4] at N == 16 would be the 1M-voice
patch), the un-normalised per-GPU buses are summed by ONE RCCL reduce per step
and the root divides by the global voice count (mixer.py:46-47).

A step = one klmi_bank_render_device call over 512 blocks (state resident,
events pre-uploaded) + the reduce (N > 1) + the D2H copy of the 512 KB bus on
the root.  Every step renders the same patch re-triggered (note-on at the
step's block 0, note-off at 86..257, all envelopes back to OFF before 512).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOICES_PER_GPU = int(os.environ.get('KLANG_BENCH_VOICES', '65536'))  # override: experiments only
BLOCKS_PER_STEP = 512
B = 256
# params 80 B + 2 x state 36 B per voice per 256-sample block, + 4 B/N bus
ALGO_BYTES_PER_VOICE_SAMPLE = 152. / 256.
HBM_PEAK_GBS = 8000.  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
# VALU issue roof: 256 CUs x 4 SIMDs, one 64-lane float64 instruction per 4 cycles at 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.          # wave-instructions/s (= 39.3e12 lane-ops/s / 64)
VALU_INSTR_PER_VOICE_SAMPLE = 24.2              # SQ_INSTS_VALU per wave-sample (profiles/), fallback


def cpu_baseline(n_voices, log):
    """The oracle (C restatement, kind 'port') on a bounded sample of the same
    workload: same patch generator, all 65,536 voices, fewer blocks."""
    from oracle import oracle
    from klang_amd import native, patches
    from tests.helpers import library_table, oracle_bank_from_patch
    table, grid = library_table(native)
    threads = max(1, min(oracle.max_threads(), int(os.environ.get('OMP_NUM_THREADS', '16')), 16))
    out = {}
    for label, nthreads, blocks in (('single', 1, 8), ('all', threads, 64)):   # ~20 core-seconds
        patch = patches.polyphonic_saw_patch(n_voices=n_voices, n_blocks=blocks, off_base=2,
                                             off_span=max(1, blocks // 2))
        bank = oracle_bank_from_patch(oracle, patch, table, grid)
        t0 = time.perf_counter()
        bank.render(blocks, patch.events, n_threads=nthreads)
        dt = time.perf_counter() - t0
        out[label] = (n_voices * blocks * B / dt, nthreads, blocks, dt)
        log('cpu_baseline %s: %d thread(s), %d blocks, %.2f s' % (label, nthreads, blocks, dt))
    v, c, blocks, dt = out['all']
    return {
        'value': v, 'unit': 'voice-samples/s', 'cores': c, 'kind': 'port',
        'sample': '%d voices x %d blocks of the same patch (oracle/klang_oracle.c, float64, '
                  'OpenMP over voices, %.1f s)' % (n_voices, blocks, dt),
        'single_thread_value': out['single'][0],
        'host_cpus_visible': os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch N>1 with: python -m torch.distributed.run --nproc-per-node N ...')

    def log(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    from klang_amd import native, patches
    from klang_amd.sharding import ShardedMixer

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    # KLANG_BENCH_REHEARSE=1: a rehearsal of the N>1 path on a ONE-GPU box -- every rank on
    # device 0, the bus reduce through gloo on host copies.  It checks the sharded patch, the
    # event upload, the barriers and the timing protocol; its numbers mean nothing.
    rehearse = world > 1 and os.environ.get('KLANG_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if rehearse:
        from klang_amd import sharding
        dist.init_process_group('gloo')

        def reduce_via_host(bus, n_total, world_size, dst=0, group=None):
            host = bus.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            bus.copy_(host)
            if n_total > 1:
                bus /= float(n_total)
            return bus
        sharding.reduce_bus = reduce_via_host
    elif world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    n_total = VOICES_PER_GPU * world
    steps_total = args.warmup + args.steps
    stream = torch.cuda.current_stream()
    ctx = native.Context(local_rank, stream.cuda_stream)
    patch = patches.polyphonic_saw_patch(n_voices=n_total, n_blocks=BLOCKS_PER_STEP,
                                         first_voice=rank * VOICES_PER_GPU, count=VOICES_PER_GPU)
    bank = native.Bank(ctx, patch.n_voices, has_env=True, has_filter=True)
    events = patch.events
    patch_no_events = patches.Patch(patch)
    patch_no_events['events'] = events[:0]
    bank.load_patch(patch_no_events)
    # pre-upload the events of every step (re-triggered patch), block numbers absolute
    all_events = []
    for s in range(steps_total):
        ev = events.copy()
        ev['block'] += s * BLOCKS_PER_STEP
        all_events.append(ev)
    bank.events(np.concatenate(all_events))

    mixer = ShardedMixer(bank, n_total, BLOCKS_PER_STEP, world, rank)
    host_bus = torch.empty((BLOCKS_PER_STEP, 1, B), dtype=torch.float32).pin_memory()

    def step():
        bus = mixer.render()            # voice kernel(s) + bus reduce (+ RCCL reduce)
        if rank == 0:
            host_bus.copy_(bus, non_blocking=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms, launches = [], 0
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP-event durations of this step's voice-kernel launches (same stream);
        # reading them waits for the step, which the next step depends on anyway
        ms, launches = bank.last_kernel_ms()
        kernel_ms.append(ms)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        peak = float(host_bus.abs().max())
        assert np.isfinite(peak) and peak > 1e-4, 'mix bus is silent or not finite (peak %r)' % peak
        units = n_total * BLOCKS_PER_STEP * B * args.steps
        value = units / elapsed
        per_launch_units = VOICES_PER_GPU * BLOCKS_PER_STEP * B / max(launches, 1)
        avg_launch_s = (sum(kernel_ms) / len(kernel_ms)) / max(launches, 1) * 1e-3
        achieved = ALGO_BYTES_PER_VOICE_SAMPLE * per_launch_units / avg_launch_s / 1e9
        traffic, valu_instr = None, VALU_INSTR_PER_VOICE_SAMPLE
        tfile = os.path.join(ROOT, 'profiles', 'traffic_voice_kernel.json')
        if os.path.exists(tfile):
            with open(tfile) as fh:
                prof = json.load(fh)
            # PMC results of tools/profile_bench.sh for this same workload (per launch)
            if prof.get('voice_samples_per_launch') == per_launch_units:
                traffic = prof.get('hbm_bytes_per_launch')
                valu_instr = prof.get('valu_instructions_per_voice_sample', valu_instr)
        result = {
            'metric': 'voice-samples/sec (osc+ADSR+biquad)', 'value': value,
            'unit': 'voice-samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': 'synthetic %d-voice polyphonic patch (saw Oscillator -> ADSR -> '
                            'Butterworth-2 lowpass -> Mixer), 44.1 kHz, 256-sample blocks, '
                            '%d blocks/step, fused mix-down' % (n_total, BLOCKS_PER_STEP),
                'voices_per_gpu': VOICES_PER_GPU, 'blocks_per_step': BLOCKS_PER_STEP,
                'parallelism': 'voice shards x%d, one RCCL reduce of the mix bus per step' % world
                               if world > 1 else 'single GPU',
            },
            'realtime_voices': value / 44100.,
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'kernel': 'voice_kernel<env,biquad,mix_mono>', 'launches_per_step': launches,
                'avg_launch_ms': avg_launch_s * 1e3,
                'algorithmic_bytes_per_voice_sample': ALGO_BYTES_PER_VOICE_SAMPLE,
                'note': 'fused path is float64-VALU-issue bound, not HBM bound (SURVEY 8d): state '
                        'stays in registers for the whole launch, so measured traffic is below the '
                        'per-block algorithmic bytes.  valu_frac = wave-instructions/s (voice-samples/s '
                        '/ 64 x measured VALU instructions per wave-sample) / (1024 SIMDs x 0.6e9 '
                        'issue slots/s); the measured float64 issue ceiling is ~0.85 of that '
                        '(profiles/r01_ubench_f64_issue.txt)',
                'valu_instructions_per_voice_sample': valu_instr,
                'valu_frac': (per_launch_units / avg_launch_s) / 64. * valu_instr / VALU_ISSUE_PEAK,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(VOICES_PER_GPU, log)
        print(json.dumps(result), flush=True)

    fence()
    bank.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
