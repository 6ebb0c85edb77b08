// mod_kernels.h -- the general voice-chain kernel for gfx950: chains whose oscillator has an
// audio-rate modulation input and/or effect stages between the voice and
// the mixer (rank 2: Gain, Transformer, TanhDistortion, OctaveDistortion, Tremolo, Delay in
// patch order, the Filter among them; klang/audio/effects.py:58-174, 452-561).  Modulation:
//   MT_FM         FmOscillator (oscillators.py:193-234):
//                  wave(phase + intensity * Oscillator(modFrequency, mod wave))
//   MT_FREQUENCY  Lfo(shape=256) | Oscillator.frequency: the array branch of sample_phase
//                  (oscillators.py:61-64): phase[0] = start, phase[n] = (TAU*DT) *
//                  cumsum(f)[n-1] + start, np.mod; np.cumsum is a sequential float64 sum
//   MT_DUTY       Lfo(shape=256) | PwmOscillator.dutyCycle (oscillators.py:256-260)
// Same mapping and data layout as voice_kernel (one lane = one voice, state in registers
// for the whole launch, the wave's LDS tile for the mix-down); one lane's time axis is
// serial anyway, so the running frequency sum needs no cross-lane scan.  Every sample
// runs the exact per-sample code (no speculation): this kernel is about parity with the
// reference's modulated paths first; envelope and filter are run-time options so that
// only the output mode is a template parameter.
#pragma once
#include "bank_kernels.h"

namespace klmi {

template <int MODE>
__global__ __launch_bounds__(kGroup) void voice_mod_kernel(BankView bk, int has_env, int has_filter,
                                                           int first, int count, int block_base,
                                                           uint32_t clock_base, int n_blocks,
                                                           int span_total,
                                                           float *__restrict__ partial,
                                                           void *__restrict__ out)
{
    using TT = TileTraits<MODE>;
    using T = typename TT::T;
    constexpr int CH = TT::chunk;
    constexpr int NCHUNK = kBlock / CH;
    constexpr int PLANES = TT::planes;
    constexpr bool MIX = (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);

    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    T *tile = reinterpret_cast<T *>(smem_raw) + (size_t)wave * (PLANES ? PLANES : 1) * CH * TT::stride;

    const int local = blockIdx.x * kGroup + threadIdx.x;
    const bool live = local < count;
    const int v = first + (live ? local : 0);
    const size_t n = (size_t)bk.n;

    // ---- parameters and state --------------------------------------------------------
    double freq = bk.frequency_rw[v];
    double amp = bk.amplitude_rw[v];
    const int wave_id = bk.wave[v];
    const double duty_edge = kTau * bk.duty[v];
    double phase = bk.phase[v];

    const int target = bk.mod_target;  // uniform over the launch
    double mod_freq = 0., mod_scale = 0., mod_offset = 0., mod_phase = 0.;
    int mod_wave = 0;
    if (target != MT_NONE) {
        mod_freq = bk.mod_par[v], mod_scale = bk.mod_par[n + v], mod_offset = bk.mod_par[2 * n + v];
        mod_wave = bk.mod_wave[v];
        mod_phase = bk.mod_phase[v];
    }

    EnvParams ep{};
    EnvState es{};
    if (has_env) {
        ep.aCoef = bk.env_coef[0 * n + v];
        ep.aBase = bk.env_coef[1 * n + v];
        ep.dCoef = bk.env_coef[2 * n + v];
        ep.dBase = bk.env_coef[3 * n + v];
        ep.rCoef = bk.env_coef[4 * n + v];
        ep.rBase = bk.env_coef[5 * n + v];
        ep.sustain = bk.env_coef[6 * n + v];
        const uint32_t fl = bk.env_flags[v];
        ep.loop = fl & 1u;
        ep.retrigger = fl & 2u;
        const uint32_t st = bk.env_state[v];
        es.stage = (int)(st & 7u);
        es.enabled = (st >> 8) & 1u;
        es.value = bk.env_value[v];
        es.select(ep);
    }

    Biquad bq{};
    int coef_row = 0;
    if (has_filter) {
        coef_row = bk.coef_index_rw[v];
        const double *c = bk.coef_table + 5 * (size_t)coef_row;
        bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
        bq.z0 = bk.z0[v];
        bq.z1 = bk.z1[v];
    }
    const bool cutoff_lfo = has_filter && bk.lfo_par != nullptr;
    double lfo_ph = cutoff_lfo ? bk.lfo_phase[v] : 0.;

    // effect stages: at most one Tremolo and one Delay per chain, so their state is scalar
    int trem_stage = -1, delay_stage = -1;
    for (int s = 0; s < bk.n_fx; ++s) {
        if (bk.fx_kind[s] == FX_TREMOLO)
            trem_stage = s;
        if (bk.fx_kind[s] == FX_DELAY)
            delay_stage = s;
    }
    double trem_phase = 0., trem_z = 0.;
    if (trem_stage >= 0) {
        trem_phase = bk.fx_state[(size_t)(2 * trem_stage) * n + v];
        trem_z = bk.fx_state[(size_t)(2 * trem_stage + 1) * n + v];
    }
    int delay_pos = 0, delay_len = 1;
    if (delay_stage >= 0) {
        delay_pos = (int)bk.fx_state[(size_t)(2 * delay_stage) * n + v];
        delay_len = (int)bk.fx_par[(size_t)(delay_stage * kFxSlots) * n + v];
    }

    double gl = 0., gr = 0.;
    if (MIX) {
        gl = live ? bk.gain_l[v] : 0.;
        if (MODE == OUT_MIX_STEREO)
            gr = live ? bk.gain_r[v] : 0.;
    }

    uint32_t ev_i = bk.ev_offset[v] + bk.ev_cursor[v];
    const uint32_t ev_end = bk.ev_offset[v + 1];
    uint32_t next_ev = (live && ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;

    for (int b = 0; b < n_blocks; ++b) {
        const uint32_t abs_block = clock_base + (uint32_t)b;
        while (next_ev <= abs_block) {  // voices.py:55-64
            const uint32_t kind = bk.ev_kind[ev_i];
            if (kind == 1u) {
                if (has_env)
                    es.gate(ep, true);
                amp = bk.ev_velocity[ev_i];
                freq = bk.ev_frequency[ev_i];
            } else if (kind == 0u) {
                if (has_env)
                    es.gate(ep, false);
            } else if (kind == 2u) {
                if (has_filter) {
                    coef_row = bk.ev_index[ev_i];
                    const double *c = bk.coef_table + 5 * (size_t)coef_row;
                    bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
                }
            } else {
                freq = bk.ev_frequency[ev_i];
            }
            ++ev_i;
            next_ev = (ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
        }
        if (cutoff_lfo)
            lfo_cutoff_step(bk, v, lfo_ph, coef_row, bq);

        const double omega = kTau * freq;          // (TAU * frequency) * t
        const double start = phase;
        const double mod_omega = kTau * mod_freq;
        const double mod_start = mod_phase;
        double running = 0.;                        // np.cumsum(frequency) within this block
        const double trem_start = trem_phase;

        for (int c = 0; c < NCHUNK; ++c) {
            const int s0 = b * kBlock + c * CH;
            for (int i = 0; i < CH; ++i) {
                const double t = c_time[c * CH + i];
                // the modulator's sample: Oscillator.sample / Lfo.sample with shape 256
                double m = 0.;
                if (target != MT_NONE) {
                    m = wave_eval_any(mod_wave, np_mod_tau(mod_omega * t + mod_start), kPi);
                    if (target != MT_FM)
                        m = mod_scale * m + mod_offset;
                }
                double x;
                if (wave_id == W_ONE) {
                    x = 1.;
                } else if (wave_id == W_EXTERNAL) {
                    x = bk.ext_input[(size_t)(block_base * kBlock + s0 + i) * n + v];
                } else if (target == MT_NONE) {
                    x = wave_eval_any(wave_id, np_mod_tau(omega * t + start), duty_edge);
                } else if (target == MT_FM) {
                    const double p = np_mod_tau(omega * t + start);
                    x = wave_eval_any(wave_id, p + mod_scale * m, duty_edge);
                } else if (target == MT_FREQUENCY) {
                    const double p = (c == 0 && i == 0) ? np_mod_tau(start)
                                                        : np_mod_tau((kTau * kDt) * running + start);
                    running = (c == 0 && i == 0) ? m : running + m;
                    x = wave_eval_any(wave_id, p, duty_edge);
                } else {  // MT_DUTY
                    const double p = np_mod_tau(omega * t + start);
                    x = (p < kTau * m) ? 1. : -1.;
                }
                if (has_env)
                    x = amp * x * es.step(ep);
                if (bk.n_fx == 0) {
                    if (has_filter)
                        x = bq.step(x);
                } else {
                    for (int s = 0; s < bk.n_fx; ++s) {  // wave-uniform program
                        const double *par = bk.fx_par + (size_t)(s * kFxSlots) * n + v;
                        switch (bk.fx_kind[s]) {
                        case FX_FILTER:
                            x = bq.step(x);
                            break;
                        case FX_GAIN:       // effects.py:67-70
                            x = par[0] * x;
                            break;
                        case FX_TRANSFORM:  // effects.py:558-561
                            x = par[0] * x + par[n];
                            break;
                        case FX_TANH:       // effects.py:480-483
                            x = tanh(par[0] * x);
                            break;
                        case FX_OCTAVE:     // effects.py:460-462
                            x = 2. * (x * x) - 1.;
                            break;
                        case FX_TREMOLO: {  // effects.py:117-133
                            const double rate = par[0], depth = par[n], duty = par[2 * n];
                            const double b0 = par[3 * n], a1 = par[4 * n];
                            const double p = np_mod_tau(kTau * rate * t + trem_start);
                            const double pwm = (p < kTau * duty) ? 1. : -1.;
                            const double y = trem_z + b0 * pwm;   // lfilter([b0], [1, a1])
                            trem_z = 0. * pwm - a1 * y;
                            x = ((1. - depth) + depth * (.5 + .5 * y)) * x;
                            break;
                        }
                        case FX_DELAY: {    // effects.py:166-174 (the blend reads the window
                                            // after the write: wet = new + feedback * delayed)
                            int idx = delay_pos + c * CH + i;
                            idx = idx >= delay_len ? idx - delay_len : idx;  // length >= one buffer
                            double *slot = bk.fx_ring + (size_t)idx * n + v;
                            const double w = x + par[n] * *slot;
                            if (live)
                                *slot = w;
                            x = (1. - par[2 * n]) * x + par[2 * n] * w;
                            break;
                        }
                        default:
                            break;
                        }
                    }
                }
                emit_sample<MODE>(tile, i, lane, x, gl, gr, live, out, s0, count, local);
            }
            flush_chunk<MODE>(tile, lane, wave, b, c, n_blocks, count, span_total, s0, partial, out);
        }
        // next start phases: sample 256 of this block
        if (trem_stage >= 0)
            trem_phase = np_mod_tau(kTau * bk.fx_par[(size_t)(trem_stage * kFxSlots) * n + v] *
                                        c_time[kBlock] + trem_start);
        if (delay_stage >= 0)
            delay_pos = (delay_pos + kBlock) % delay_len;
        if (target != MT_NONE)
            mod_phase = np_mod_tau(mod_omega * c_time[kBlock] + mod_start);
        if (wave_id != W_ONE && wave_id != W_EXTERNAL) {
            if (target == MT_FREQUENCY)
                phase = np_mod_tau((kTau * kDt) * running + start);
            else
                phase = np_mod_tau(omega * c_time[kBlock] + start);
        }
    }

    if (live) {
        bk.phase[v] = phase;
        if (target != MT_NONE)
            bk.mod_phase[v] = mod_phase;
        bk.frequency_rw[v] = freq;
        bk.amplitude_rw[v] = amp;
        if (has_env) {
            bk.env_value[v] = es.value;
            bk.env_state[v] = (uint32_t)es.stage | (es.enabled ? 0x100u : 0u);
        }
        if (has_filter) {
            bk.z0[v] = bq.z0;
            bk.z1[v] = bq.z1;
            bk.coef_index_rw[v] = coef_row;
            if (cutoff_lfo)
                bk.lfo_phase[v] = lfo_ph;
        }
        if (trem_stage >= 0) {
            bk.fx_state[(size_t)(2 * trem_stage) * n + v] = trem_phase;
            bk.fx_state[(size_t)(2 * trem_stage + 1) * n + v] = trem_z;
        }
        if (delay_stage >= 0)
            bk.fx_state[(size_t)(2 * delay_stage) * n + v] = (double)delay_pos;
        bk.ev_cursor[v] = ev_i - bk.ev_offset[v];
    }
}

}  // namespace klmi
