#!/usr/bin/env python3
"""Secondary measurements (one JSON line each) for the BASELINE configs that are
not bench.py's headline: config 2 (haunting patch, replicated), config 3 in
materialised mode (per-voice float32 streams written to HBM) and config 4
(reverb channels, delay lines in HBM).  Usage:

    python tools/bench_configs.py [2] [3m] [3fm] [3vib] [3pwm] [3fx] [3delay] [synth] [4] [4big]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from klang_amd import native, patches  # noqa: E402

B = 256
HBM_PEAK = 8000.


def timed(fn, reps=3):
    fn()
    best = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best.append(time.perf_counter() - t0)
    return min(best), float(np.median(best))


def bank_for(ctx, patch):
    bank = native.Bank(ctx, patch.n_voices, has_env=patch.has_env, has_filter=patch.has_filter,
                       stereo=patch.stereo, has_reverb='reverb' in patch,
                       reverb=patch.get('reverb'))
    ev = patch.events
    p = patches.Patch(patch)
    p['events'] = ev[:0]
    bank.load_patch(p)
    return bank, ev


def queue(bank, ev, n_blocks, runs):
    allev = []
    for r in range(runs):
        e = ev.copy()
        e['block'] += r * n_blocks
        allev.append(e)
    bank.events(np.concatenate(allev))


def config2(ctx, replicas=1024):
    n_blocks = 2048
    patch = patches.haunting_patch(replicas=replicas, n_blocks=n_blocks)
    bank, ev = bank_for(ctx, patch)
    bank.events(ev)
    best, med = timed(lambda: bank.render(n_blocks))
    ms, launches = bank.last_kernel_ms()
    units = patch.n_voices * n_blocks * B
    return dict(config='2: haunting_envelopes x%d (%d looping AR sine voices -> StereoMixer), '
                       '%d blocks' % (replicas, patch.n_voices, n_blocks),
                value=units / med, unit='voice-samples/s', wall_ms=med * 1e3, kernel_ms=ms,
                launches=launches)


def config3_materialised(ctx):
    import torch
    # 256 blocks per launch as in the fused bench (a 17 GB float32 buffer); override: experiments
    n, n_blocks = 65536, int(os.environ.get('KLANG_BENCH_BLOCKS', '256'))
    patch = patches.polyphonic_saw_patch(n_voices=n, n_blocks=n_blocks, off_base=min(10, n_blocks // 2),
                                         off_span=max(1, min(20, n_blocks // 3)))
    bank, ev = bank_for(ctx, patch)
    bank.events(ev)
    out = torch.empty((n, n_blocks * B), dtype=torch.float32, device='cuda')

    def run():
        bank.render_voices_device(n_blocks, out.data_ptr())
        ctx.synchronize()
    best, med = timed(run)
    ms, launches = bank.last_kernel_ms()
    units = n * n_blocks * B
    algo = units * 4.594  # A3m = 4.594 B per voice-sample
    return dict(config='3 materialised: 65,536 voices x %d blocks, one float32 stream per voice '
                       'written to HBM' % n_blocks,
                value=units / (ms * 1e-3), unit='voice-samples/s', wall_ms=med * 1e3,
                kernel_ms=ms,
                roofline=dict(bound='hbm', achieved=algo / (ms * 1e-3) / 1e9, peak=HBM_PEAK,
                              unit='GB/s', frac=algo / (ms * 1e-3) / 1e9 / HBM_PEAK,
                              algorithmic_bytes_per_voice_sample=4.594))


def config3_general(ctx, label, **bank_kw):
    """Config 3's patch through the general kernel (mod_kernels.h): with an FmOscillator
    per voice, or with effect stages around the Filter."""
    n, n_blocks = int(os.environ.get('KLANG_CONFIG3_VOICES', 65536)), 128
    patch = patches.polyphonic_saw_patch(n_voices=n, n_blocks=n_blocks, off_base=20, off_span=40)
    bank = native.Bank(ctx, n, has_env=True, has_filter=True, **bank_kw)
    ev = patch.events
    p = patches.Patch(patch)
    p['events'] = ev[:0]
    bank.load_patch(p)
    rng = np.random.default_rng(9)
    target = bank_kw.get('mod_target')
    if target == native.MOD_FM:
        bank.set(native.F_MOD_FREQUENCY, 20. + 200. * rng.random(n))
        bank.set(native.F_MOD_SCALE, 2. * rng.random(n))
        bank.set(native.F_MOD_WAVE, np.full(n, native.WAVE_SAWTOOTH))
    elif target == native.MOD_FREQUENCY:      # vibrato: +-2 % around the voice's pitch, 3-8 Hz
        bank.set(native.F_MOD_FREQUENCY, 3. + 5. * rng.random(n))
        bank.set(native.F_MOD_SCALE, .02 * np.asarray(patch.frequency))
        bank.set(native.F_MOD_OFFSET, np.asarray(patch.frequency, dtype=float))
        bank.set(native.F_MOD_WAVE, np.full(n, native.WAVE_SINE))
    elif target == native.MOD_DUTY:           # pulse width swept between .1 and .9
        bank.set(native.F_WAVE, np.full(n, native.WAVE_PWM))
        bank.set(native.F_MOD_FREQUENCY, .5 + 4. * rng.random(n))
        bank.set(native.F_MOD_SCALE, np.full(n, .4))
        bank.set(native.F_MOD_OFFSET, np.full(n, .5))
        bank.set(native.F_MOD_WAVE, np.full(n, native.WAVE_TRIANGLE))
    for s, kind in enumerate(bank_kw.get('effects', ())):
        if kind == native.FX_TREMOLO:
            bank.set(native.F_FX(s, 0), 1. + 8. * rng.random(n))
        elif kind == native.FX_DELAY:
            mode = os.environ.get('KLANG_BENCH_DELAY', 'random')
            if mode == 'equal':       # one delay time for the whole bank (the usual patch)
                bank.set(native.F_FX(s, 0), np.full(n, .0113))
            elif mode == 'few':       # a handful of times, as a patch with note-value delays has
                bank.set(native.F_FX(s, 0), np.repeat([.0075, .009, .0113, .015], n // 4 + 1)[:n])
            else:                     # worst case: every voice its own time
                bank.set(native.F_FX(s, 0), .006 + .01 * rng.random(n))
            label += ' [%s times]' % mode
        elif kind == native.FX_GAIN:
            bank.set(native.F_FX(s, 0), rng.random(n))
        elif kind == native.FX_RINGMOD:     # frequency, rate, amount, dryWet, Lfo wave (square)
            bank.set(native.F_FX(s, 0), 200. + 600. * rng.random(n))
            bank.set(native.F_FX(s, 1), .5 + 4. * rng.random(n))
            bank.set(native.F_FX(s, 2), 30. * rng.random(n))
            bank.set(native.F_FX(s, 3), rng.random(n))
            bank.set(native.F_FX(s, 4), np.full(n, float(native.WAVE_SQUARE)))
        elif kind == native.FX_BITCRUSH:
            bank.set(native.F_FX(s, 0), rng.integers(4, 12, n).astype(float))
        elif kind == native.FX_SUBSAMPLE:
            bank.set(native.F_FX(s, 0), (2 ** rng.integers(1, 4, n)).astype(float))
    queue(bank, ev, n_blocks, 5)
    best, med = timed(lambda: bank.render(n_blocks))
    ms, launches = bank.last_kernel_ms()
    units = n * n_blocks * B
    kernel = ('general kernel (KLMI_FM_KERNEL / KLMI_FX_KERNEL = 0)'
              if '0' in (os.environ.get('KLMI_FM_KERNEL'), os.environ.get('KLMI_FX_KERNEL'))
              else 'modulated form of the fast kernel' if bank_kw.get('mod_target') else 'compiled stage program')
    return dict(config='3 with %s, %s: %s voices x %d blocks' % (label, kernel, format(n, ','), n_blocks),
                value=units / med, unit='voice-samples/s', wall_ms=med * 1e3, kernel_ms=ms,
                launches=launches)


def config_synth(ctx, fm=False):
    """2,730 PolyphonicSynthesizers x 24 voices (65,520 voices, saw -> ADSR, gated) under one
    Mixer; ~6 notes per instrument sounding, note routing on the device (klmi_bank_notes)."""
    poly, n_inst, n_blocks = 24, 2730, 128
    n = poly * n_inst
    bank = native.Bank(ctx, n, has_env=True, poly_voices=poly, mod_target=native.MOD_FM if fm else native.MOD_NONE)
    bank.set(native.F_WAVE, np.full(n, native.WAVE_SAWTOOTH))
    if fm:      # FmOscillator voices: carrier and modulator share the wave function
        bank.set(native.F_MOD_WAVE, np.full(n, native.WAVE_SAWTOOTH))
        bank.set(native.F_MOD_FREQUENCY, np.full(n, 110.))
        bank.set(native.F_MOD_SCALE, np.full(n, 2.5))
    bank.set_mix_divisor(poly * n_inst)
    rng = np.random.default_rng(5)
    per_inst, runs = 40, 5

    def notes(run):
        m = n_inst * per_inst
        nt = np.zeros(2 * m, dtype=native.NOTE_DTYPE)
        inst = np.repeat(np.arange(n_inst), per_inst)
        on = rng.integers(0, n_blocks - 8, m)
        pitch = rng.integers(36, 96, m)
        nt['instrument'] = np.r_[inst, inst]
        nt['block'] = np.r_[on, on + rng.integers(1, 24, m).clip(max=n_blocks - 1 - on)] + run * n_blocks
        nt['pitch'] = np.r_[pitch, pitch]
        nt['velocity'] = np.r_[.2 + .8 * rng.random(m), np.zeros(m)]
        nt['frequency'] = np.tile(440. * 2. ** ((pitch - 69) / 12.), 2)
        return nt
    bank.notes(np.concatenate([notes(r) for r in range(runs)]))
    best, med = timed(lambda: bank.render(n_blocks))
    ms, launches = bank.last_kernel_ms()
    units = n * n_blocks * B
    return dict(config='synthesizers: %d PolyphonicSynthesizers x %d voices (%s -> ADSR, gated) -> '
                       'Mixer, %d notes per synthesizer and render, device-side routing, %d blocks'
                       % (n_inst, poly, 'FmOscillator(saw)' if fm else 'saw', per_inst, n_blocks),
                knobs={k: v for k, v in os.environ.items() if k.startswith('KLMI_')},
                value=units / med, unit='voice-samples/s', wall_ms=med * 1e3, kernel_ms=ms,
                launches=launches)


def config4(ctx, n_channels=4096, n_blocks=512, rings='f64'):
    patch = patches.reverb_patch(n_channels=n_channels, n_blocks=n_blocks)
    patch['reverb'] = dict(patch['reverb'], rings=rings)
    if os.environ.get('KLANG_BENCH_NOENV') == '1':      # experiment: plain oscillators
        patch['has_env'] = False
    bank, ev = bank_for(ctx, patch)
    queue(bank, ev, n_blocks, 5)
    best, med = timed(lambda: bank.render(n_blocks))
    ms, launches = bank.last_kernel_ms()
    units = n_channels * n_blocks * B
    width = 4 if rings == 'f32' else 8
    # 10 combs x (read + write) of a ring slot.  The dry signal is made inside the
    # reverb wave (KLMI_REVERB_FUSED=0: the two-kernel path, + 8 B written + 8 B read of dry bus)
    per_sample = 10 * 2 * width
    fused = os.environ.get('KLMI_REVERB_FUSED', '1') != '0'
    algo = units * per_sample
    return dict(config='4: %d channels Voice(sine, AR(.01,1.)) -> Reverb(decay 3, dryWet .7) -> '
                       'Mixer, %d blocks, %s rings (%.2f GB of delay lines)'
                       % (n_channels, n_blocks, rings, n_channels * 13896 * width / 1e9),
                value=units / med, unit='channel-samples/s', wall_ms=med * 1e3,
                reverb_kernel_ms=ms, reverb_launches=launches,
                dry_bus_bytes_per_channel_sample=0 if fused else 16,
                knobs={k: v for k, v in os.environ.items() if k.startswith(('KLMI_', 'KLANG_BENCH'))},
                roofline=dict(bound='hbm', achieved=algo / (ms * 1e-3) / 1e9, peak=HBM_PEAK,
                              unit='GB/s', frac=algo / (ms * 1e-3) / 1e9 / HBM_PEAK,
                              kernel='reverb_voice_kernel' if fused else 'reverb_kernel',
                              algorithmic_bytes_per_channel_sample=per_sample,
                              whole_call_frac=algo / med / 1e9 / HBM_PEAK))


def main():
    which = sys.argv[1:] or ['2', '3m', '4']
    # torch bundles its own HIP runtime: it has to initialise before libklang_mi.so
    # pulls in the system one ("PyTorch in the same process")
    import torch
    torch.cuda.init()
    ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
    for w in which:
        if w == '2':
            r = config2(ctx)
        elif w == '2native':
            r = config2(ctx, 1)
        elif w == '3m':
            r = config3_materialised(ctx)
        elif w == '4':
            r = config4(ctx)
        elif w == '3fm':
            r = config3_general(ctx, 'FmOscillator (saw modulator)', mod_target=native.MOD_FM)
        elif w == '3vib':
            r = config3_general(ctx, 'Lfo(256, sine) -> Oscillator.frequency', mod_target=native.MOD_FREQUENCY)
        elif w == '3pwm':
            r = config3_general(ctx, 'Lfo(256, triangle) -> PwmOscillator.dutyCycle',
                                mod_target=native.MOD_DUTY)
        elif w == '3fx':
            r = config3_general(ctx, 'Filter -> Tremolo -> Gain',
                                effects=[native.FX_FILTER, native.FX_TREMOLO, native.FX_GAIN])
        elif w == '3delay':
            r = config3_general(ctx, 'Filter -> Delay(6-16 ms) -> Gain',
                                effects=[native.FX_FILTER, native.FX_DELAY, native.FX_GAIN])
        elif w == '3lofi':
            r = config3_general(ctx, 'Filter -> Gain -> Bitcrusher -> Subsampler',
                                effects=[native.FX_FILTER, native.FX_GAIN, native.FX_BITCRUSH,
                                         native.FX_SUBSAMPLE])
        elif w == '3ring':
            r = config3_general(ctx, 'Filter -> RingModulator -> Gain',
                                effects=[native.FX_FILTER, native.FX_RINGMOD, native.FX_GAIN])
        elif w == 'synth':
            r = config_synth(ctx)
        elif w == 'synthfm':
            r = config_synth(ctx, fm=True)
        elif w == '4big':
            r = config4(ctx, 16384, 128)
        elif w == '4f32':
            r = config4(ctx, rings='f32')
        elif w == '4bigf32':
            r = config4(ctx, 16384, 128, rings='f32')
        elif w == 'probe':
            r = dict(config='stream probe (klmi_ctx_stream_rate, 1 GiB buffers)',
                     **{k: ctx.stream_rate(k) for k in ('copy', 'update', 'fill', 'sum')})
        else:
            raise SystemExit('unknown config %r' % w)
        print(json.dumps(r), flush=True)
    ctx.close()


if __name__ == '__main__':
    main()
