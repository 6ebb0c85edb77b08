#!/usr/bin/env python3
"""bench.py -- voice-samples/sec of the fused osc -> ADSR -> biquad -> Mixer path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N == 1: BASELINE.json configs[2], the synthetic 65,536-voice polyphonic patch
(saw Oscillator -> ADSR -> Butterworth-2 lowpass -> Mixer, 44.1 kHz, 256-sample
blocks), 512 blocks per step.  N > 1 (launched by torch.distributed.run, one
rank per GPU): BASELINE.json configs[4]'s sharding -- every rank renders its own
contiguous 131,072-voice shard of the N*131,072-voice patch (weak scaling; N == 8
IS configs[4], the 1,048,576-voice patch), the un-normalised per-GPU buses are
summed by ONE RCCL reduce per step and the root divides by the global voice
count (mixer.py:46-47).

At N == 1 the JSON line also carries two sub-records measured in the same run
(`extra`): config 3 materialised (one float32 stream per voice written to HBM)
and BASELINE configs[3] (4,096 reverb channels, comb delay lines in HBM), the
two HBM-bound shapes of the path, each against the 8 TB/s spec and against the
rate a plain hand-written streaming kernel reaches on this GPU in this run.

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

VOICES_SINGLE = 65536     # configs[2]
VOICES_SHARD = 131072     # configs[4]: 1,048,576 voices over 8 GPUs
BLOCKS_PER_STEP = 512
B = 256
# params 80 B + 2 x state 36 B per voice per 256-sample block, + 4 B/N bus
ALGO_BYTES_PER_VOICE_SAMPLE = 152. / 256.
HBM_PEAK_GBS = 8000.  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
# VALU issue roof: 256 CUs x 4 SIMDs, one 64-lane float64 instruction per 4 cycles at 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.          # wave-instructions/s (= 39.3e12 lane-ops/s / 64)
VALU_INSTR_PER_VOICE_SAMPLE = 24.2              # SQ_INSTS_VALU per wave-sample (profiles/), fallback


def stats(xs):
    xs = sorted(xs)
    return {'median': xs[len(xs) // 2] if len(xs) % 2 else .5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2]),
            'min': xs[0], 'max': xs[-1]}


def extra_records(ctx, native, patches, torch, log):
    """The two HBM-bound shapes of the path, measured in this run (N == 1 only)."""
    out = {}
    rates = {k: ctx.stream_rate(k, 1 << 30, 10) for k in ('copy', 'update', 'fill')}
    out['stream_probe_GBs'] = dict(rates, note='plain hand-written kernels of this library, one float4 per '
                                   'lane, 1 GiB buffers, bytes read + written (klmi_ctx_stream_rate)')
    log('stream probe GB/s: %s' % ', '.join('%s %.0f' % kv for kv in rates.items()))

    def timed(bank, n_blocks, fn, runs=5):
        wall, kern = [], []
        for _ in range(runs + 1):
            ctx.synchronize()
            t0 = time.perf_counter()
            fn()
            ctx.synchronize()
            wall.append(time.perf_counter() - t0)
            kern.append(bank.last_kernel_ms()[0])
        return wall[1:], kern[1:], bank.last_kernel_ms()[1]

    def no_events(patch):
        p = patches.Patch(patch)
        p['events'] = patch.events[:0]
        return p

    def queue(bank, ev, n_blocks, runs):
        allev = []
        for r in range(runs):
            e = ev.copy()
            e['block'] += r * n_blocks
            allev.append(e)
        bank.events(np.concatenate(allev))

    # -- config 3 materialised: A3m = 4.594 B per voice-sample.  At 65,536 voices (one
    # wave per SIMD) the mode is bound by the same float64 arithmetic as the mixed-down headline;
    # the 131,072-voice bank (two waves per SIMD, the per-GPU shard of configs[4]) shows what the
    # store path reaches when the arithmetic leaves it room
    for n, nb in ((65536, 256), (VOICES_SHARD, 128)):
        patch = patches.polyphonic_saw_patch(n_voices=n, n_blocks=nb, off_base=10, off_span=20)
        bank = native.Bank(ctx, n, has_env=True, has_filter=True)
        bank.load_patch(patch)
        buf = torch.empty((n, nb * B), dtype=torch.float32, device='cuda')
        wall, kern, launches = timed(bank, nb, lambda: bank.render_voices_device(nb, buf.data_ptr()))
        units, ms = n * nb * B, stats(kern)['median']
        gbs = units * 4.594 / (ms * 1e-3) / 1e9
        out['materialised' if n == 65536 else 'materialised_%d' % n] = {
            'workload': 'config 3 materialised: %d voices x %d blocks, one float32 stream per '
                        'voice written to HBM (klmi_bank_render_voices_device)' % (n, nb),
            'value': units / (ms * 1e-3), 'unit': 'voice-samples/s', 'kernel_ms': stats(kern),
            'wall_ms': stats([w * 1e3 for w in wall]),
            'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': gbs / HBM_PEAK_GBS, 'frac_of_probe_fill': gbs / rates['fill'],
                         'algorithmic_bytes_per_voice_sample': 4.594,
                         'kernel': 'voice_kernel<env,biquad,voices_f32>'}}
        del buf
        bank.close()
        log('materialised, %d voices: %.3g voice-samples/s, %.0f GB/s' % (n, units / (ms * 1e-3), gbs))

    # -- BASELINE configs[3]: reverb channels, delay lines in HBM.  A4 = 10 combs x
    # (read + write) x ring width; the dry voice is made inside the reverb wave, so no dry bus
    # moves through HBM (dry_bus_bytes_per_channel_sample: 0; the two-kernel path moved 16)
    for n, nb, rings in ((4096, 512, 'f32'), (16384, 128, 'f32'), (4096, 512, 'f64')):
        per_sample = 80. if rings == 'f32' else 160.
        patch = patches.reverb_patch(n_channels=n, n_blocks=nb)
        patch['reverb'] = dict(patch['reverb'], rings=rings)
        bank = native.Bank(ctx, n, has_env=True, has_reverb=True, reverb=patch['reverb'])
        assert bank.desc.reverb_rings_f64 == (rings == 'f64')
        bank.load_patch(no_events(patch))
        queue(bank, patch.events, nb, 6)
        bus = torch.empty((nb, 1, B), dtype=torch.float32, device='cuda')
        wall, kern, launches = timed(bank, nb, lambda: bank.render_device(nb, bus.data_ptr()))
        units = n * nb * B
        ms, wall_ms = stats(kern)['median'], stats([w * 1e3 for w in wall])
        gbs = units * per_sample / (ms * 1e-3) / 1e9
        whole = units * per_sample / (wall_ms['median'] * 1e-3) / 1e9
        key = 'config4_rings_' + rings + ('' if n == 4096 else '_%d' % n)
        out[key] = {
            'workload': 'configs[3]: %d channels Voice(sine, AR(.01,1.)) -> Reverb(decay 3, dryWet .7) '
                        '-> Mixer, %d blocks, %s ring storage (%.2f GB of delay lines)'
                        % (n, nb, rings, n * 13896 * (4 if rings == 'f32' else 8) / 1e9),
            'value': units / (wall_ms['median'] * 1e-3), 'unit': 'channel-samples/s',
            'reverb_kernel_ms': stats(kern), 'reverb_launches': launches, 'wall_ms': wall_ms,
            'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': gbs / HBM_PEAK_GBS, 'whole_call_frac': whole / HBM_PEAK_GBS,
                         'frac_of_probe_update': gbs / rates['update'],
                         'algorithmic_bytes_per_channel_sample': per_sample,
                         'dry_bus_bytes_per_channel_sample': 0.,
                         'kernel': 'reverb_voice_kernel'}}
        bank.close()
        log('config4 %d ch %s rings: %.3g channel-samples/s, kernel %.0f GB/s' %
            (n, rings, units / (wall_ms['median'] * 1e-3), gbs))
    return out


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


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, script=None, env=None, timeout=None, out=sys.stdout, err=sys.stderr):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes
    (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the same
    variables torch.distributed.run would set), wait for all of them, relay rank 0's stdout
    (the ONE JSON line) and return a non-zero code if any rank failed.  The parent never
    touches the GPU or imports torch: nothing that has initialised HIP is re-executed.
    A rank that dies takes the others down (they would wait in the rendezvous forever)."""
    import subprocess
    script = script or os.path.abspath(__file__)
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                MASTER_PORT=str(free_port()))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, '-u', script] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None if err is sys.stderr else subprocess.PIPE,
                                      text=True))
    deadline = None if timeout is None else time.monotonic() + timeout
    codes = [None] * n
    rank0_out = None
    try:
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None and p.poll() is not None:
                    codes[r] = p.returncode
                    if r == 0:
                        rank0_out = p.stdout.read()
            if any(c not in (None, 0) for c in codes):
                break
            if deadline is not None and time.monotonic() > deadline:
                break
            time.sleep(.05)
    finally:
        for r, p in enumerate(procs):           # exact PIDs we started, nothing by pattern
            if p.poll() is None:
                p.terminate()
        for r, p in enumerate(procs):
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
            if codes[r] is None:
                codes[r] = p.returncode if p.returncode is not None else -9
        if rank0_out is None and procs[0].stdout is not None:
            try:
                rank0_out = procs[0].stdout.read()
            except (OSError, ValueError):
                rank0_out = ''
    if err is not sys.stderr:
        for r, p in enumerate(procs):
            if p.stderr is not None:
                err.write(p.stderr.read())
    failed = [(r, c) for r, c in enumerate(codes) if c != 0]
    if failed:
        print('[bench] ranks failed (rank, exit code): %s' % failed, file=sys.stderr, flush=True)
        return 1
    # libraries may chat on stdout (gloo does); the record is the line that is a JSON object
    lines = []
    for ln in (rank0_out or '').splitlines():
        if ln.lstrip().startswith('{'):
            try:
                json.loads(ln)
                lines.append(ln.strip())
            except ValueError:
                pass
        elif ln.strip():
            print('[rank 0 stdout] ' + ln, file=sys.stderr, flush=True)
    if len(lines) != 1:
        print('[bench] rank 0 printed %d JSON lines, expected ONE' % len(lines),
              file=sys.stderr, flush=True)
        return 1
    out.write(lines[0] + '\n')
    out.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the sub-records (N == 1)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        # no launcher: this process becomes the launcher (before any torch / GPU call)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))

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
        dist.init_process_group('gloo')     # sharding.reduce_bus hops through the host then
    elif world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    # N == 1 is configs[2]; N > 1 shards like configs[4] (override: experiments only)
    VOICES_PER_GPU = int(os.environ.get('KLANG_BENCH_VOICES',
                                        VOICES_SINGLE if world == 1 else VOICES_SHARD))
    n_total = VOICES_PER_GPU * world
    stream = torch.cuda.current_stream()
    ctx = native.Context(local_rank, stream.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n_total, voices_per_gpu, warmup, steps):
        """W untimed + exactly K timed steps of the sharded patch; returns the max-over-ranks
        wall time of the K steps and the per-step HIP-event kernel times of this rank."""
        steps_total = warmup + steps
        patch = patches.polyphonic_saw_patch(n_voices=n_total, n_blocks=BLOCKS_PER_STEP,
                                             first_voice=rank * voices_per_gpu, count=voices_per_gpu)
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

        kernel_ms, step_ms, launches = [], [], 0
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            step()
            # HIP-event durations of this step's voice-kernel launches (same stream);
            # reading them waits for the step, which the next step depends on anyway
            ms, launches = bank.last_kernel_ms()
            kernel_ms.append(ms)
            step_ms.append((time.perf_counter() - ts) * 1e3)
        fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            peak = float(host_bus.abs().max())
            assert np.isfinite(peak) and peak > 1e-4, \
                'mix bus is silent or not finite (peak %r)' % peak
        fence()
        bank.close()
        return elapsed, kernel_ms, step_ms, launches

    elapsed, kernel_ms, step_ms, launches = run_steps(n_total, VOICES_PER_GPU, args.warmup,
                                                      args.steps)

    if rank == 0:
        units = n_total * BLOCKS_PER_STEP * B * args.steps
        value = units / elapsed
        per_launch_units = VOICES_PER_GPU * BLOCKS_PER_STEP * B / max(launches, 1)
        avg_launch_s = (sum(kernel_ms) / len(kernel_ms)) / max(launches, 1) * 1e-3
        median_launch_s = stats(kernel_ms)['median'] / max(launches, 1) * 1e-3
        achieved = ALGO_BYTES_PER_VOICE_SAMPLE * per_launch_units / avg_launch_s / 1e9
        # PMC counters cannot be read from inside this process: traffic and the instruction
        # count are STATIC figures from the committed rocprofv3 passes over this same command
        # (tools/profile_bench.sh), used only when they were taken at this launch size
        traffic, valu_instr, counters_from = None, VALU_INSTR_PER_VOICE_SAMPLE, None
        tfile = os.path.join(ROOT, 'profiles', 'traffic_voice_kernel.json')
        if os.path.exists(tfile):
            with open(tfile) as fh:
                prof = json.load(fh)
            if prof.get('voice_samples_per_launch') == per_launch_units:
                traffic = prof.get('hbm_bytes_per_launch')
                valu_instr = prof.get('valu_instructions_per_voice_sample', valu_instr)
                counters_from = 'profiles/traffic_voice_kernel.json (static: rocprofv3 --pmc ' \
                                'passes over this command, %s)' % prof.get('round', 'earlier round')
        result = {
            'metric': 'voice-samples/sec (osc+ADSR+biquad)', 'value': value,
            'unit': 'voice-samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'ms_per_step_spread': stats(step_ms),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': 'synthetic %s-voice polyphonic patch (saw Oscillator -> ADSR -> '
                            'Butterworth-2 lowpass -> Mixer), 44.1 kHz, 256-sample blocks, '
                            '%d blocks/step, fused mix-down%s'
                            % (format(n_total, ','), BLOCKS_PER_STEP,
                               '' if world == 1 else ', sharded %s voices per GPU across %d GPUs%s'
                               % (format(VOICES_PER_GPU, ','), world,
                                  ' (BASELINE configs[4])' if n_total == 1048576 else '')),
                'voices_per_gpu': VOICES_PER_GPU, 'blocks_per_step': BLOCKS_PER_STEP,
                'parallelism': 'voice shards x%d, one RCCL reduce of the mix bus per step' % world
                               if world > 1 else 'single GPU',
            },
            'per_gpu_value': value / world,
            'realtime_voices': value / 44100.,
            'roofline': {
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'kernel': 'voice_kernel<env,biquad,mix_mono>', 'launches_per_step': launches,
                'avg_launch_ms': avg_launch_s * 1e3, 'median_launch_ms': median_launch_s * 1e3,
                'launch_ms_spread': stats([k / max(launches, 1) for k in kernel_ms]),
                'traffic_source': counters_from,
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
        if world == 1 and not args.no_extra:
            result['extra'] = extra_records(ctx, native, patches, torch, log)
            if VOICES_PER_GPU != VOICES_SHARD:
                # the N = 1 point on the per-GPU workload of N > 1 (131,072 voices per GPU): the
                # same step, timed the same way, so that the N > 1 lines have a like-for-like base
                e2, k2, s2, l2 = run_steps(VOICES_SHARD, VOICES_SHARD, args.warmup, args.steps)
                v2 = VOICES_SHARD * BLOCKS_PER_STEP * B * args.steps / e2
                result['extra']['shard_131072'] = {
                    'workload': 'the same step on the per-GPU shard size of --gpus N > 1: 131,072 '
                                'voices on this one GPU (two waves per SIMD instead of one)',
                    'value': v2, 'unit': 'voice-samples/s', 'ms_per_step': e2 / args.steps * 1e3,
                    'ms_per_step_spread': stats(s2), 'steps': args.steps, 'warmup': args.warmup,
                    'launch_ms_spread': stats([k / max(l2, 1) for k in k2]),
                    'launches_per_step': l2,
                    'note': 'weak-scaling efficiency at N GPUs like for like = per_gpu_value(N) / '
                            'this value; the headline N = 1 line is BASELINE configs[2] (65,536 '
                            'voices, one wave per SIMD)'}
                log('shard_131072: %.4g voice-samples/s' % v2)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(VOICES_PER_GPU, log)
        print(json.dumps(result), flush=True)

    fence()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
