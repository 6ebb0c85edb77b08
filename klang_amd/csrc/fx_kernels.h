// fx_kernels.h -- the voice kernel for chains with a short, common effect program behind the
// [Filter]: Tremolo, Delay and element-wise stages (Gain, Transformer, TanhDistortion,
// OctaveDistortion, Subsampler, Bitcrusher: klang/audio/effects.py:58-174, 388-483, 522-561), the program
// known at compile time (gfx950).
//
// voice_mod_kernel (mod_kernels.h) interprets ANY stage program, stage-major over an LDS strip:
// every stage is a loop of its own, so a stage that is one serial recurrence (the biquad, the
// Tremolo's smoothing filter) runs at the latency of its dependent float64 operations with
// nothing to fill the gaps -- one wavefront per SIMD is all a 65,536-voice bank gives the
// scheduler.  Here the program is a template parameter and eight samples of the whole chain
// are two regions of straight-line code, as in voice_kernel:
//   1. the oscillator: eight independent samples, the wave function chosen by one scalar
//      jump per group (closed-form phase, so its speculative remainder is repaired in place);
//   2. envelope, biquad and the Tremolo's Lfo + smoothing filter -- three independent serial
//      recurrences the scheduler interleaves -- with the Delay's eight delayed samples
//      fetched before region 1 (the line is at least one buffer long, so they never depend on
//      this group's writes) and the stages applied in patch order at the end.
// Only the envelope is speculated with a redo (its stage switches depend on rounded values);
// the Tremolo's remainder has its own check and exact form.  Per sample the arithmetic is the
// reference's, operation for operation, and equal to voice_mod_kernel's bit for bit.
#pragma once
#ifndef KLMI_DELAY_CARRY
#define KLMI_DELAY_CARRY 1
#endif
#include "bank_kernels.h"
#include <type_traits>

namespace klmi {

// A program: up to three stage kinds (FxKind), four bits each, the first stage lowest.
constexpr uint32_t fx_program(int a, int b = 0, int c = 0)
{
    return (uint32_t)a | ((uint32_t)b << 4) | ((uint32_t)c << 8);
}
constexpr int fx_program_kind(uint32_t prog, int s) { return (int)((prog >> (4 * s)) & 15u); }
// Slot kind of a program that stands for ANY element-wise stage (Gain, Transformer,
// TanhDistortion, OctaveDistortion, Subsampler, Bitcrusher), its kind chosen at run time by a
// scalar jump; FX_GAIN slots stay plain multiplies (the Tremolo / Delay / Gain programs of the
// reference's patches lose nothing to the generality)
constexpr int FX_ELEM = 15;
constexpr int fx_program_length(uint32_t prog)
{
    return fx_program_kind(prog, 2) ? 3 : fx_program_kind(prog, 1) ? 2 : fx_program_kind(prog, 0) ? 1 : 0;
}
constexpr int fx_program_find(uint32_t prog, int kind)   // stage index of `kind`, -1: none
{
    for (int s = 0; s < 3; ++s)
        if (fx_program_kind(prog, s) == kind)
            return s;
    return -1;
}

struct __attribute__((packed, aligned(8))) DoublePair {
    double a, b;
};

// grid.x = ceil(count / 256) workgroups of 4 waves; partial / out as voice_kernel.
// The chain is bk.fx_kind = [FX_FILTER if FILT] + PROG's stages (the launcher checks).
template <bool ENV, bool FILT, int MODE, uint32_t PROG>
__global__ __launch_bounds__(kGroup) void voice_fx_kernel(BankView bk, int first, int count,
                                                          int block_base, uint32_t clock_base,
                                                          int n_blocks, int span_total,
                                                          float *__restrict__ partial,
                                                          void *__restrict__ out)
{
    using TT = TileTraits<MODE>;
    using T = typename TT::T;
    constexpr int CH = TT::chunk;
    constexpr int NCHUNK = kBlock / CH;
    constexpr int PLANES = TT::planes;
    constexpr bool MIX = (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);
    constexpr int G = 8;
    constexpr int NSTAGE = fx_program_length(PROG);
    constexpr int FIRST_STAGE = FILT ? 1 : 0;   // the Filter, when there is one, is stage 0
    constexpr int TREM = fx_program_find(PROG, FX_TREMOLO);
    constexpr int DELAY = fx_program_find(PROG, FX_DELAY);
    constexpr int RING = fx_program_find(PROG, FX_RINGMOD);
    static_assert(NSTAGE >= 1, "plain chains run in voice_kernel");

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
    const double duty_edge = kTau * bk.duty[v];  // oscillators.py:259
    const int first_wave = __builtin_amdgcn_readfirstlane(wave_id);
    const int uniform_wave = __all(wave_id == first_wave) ? first_wave : -1;
    double phase = bk.phase[v];

    EnvParams ep{};
    EnvState es{};
    if (ENV) {
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
    if (FILT) {
        coef_row = bk.coef_index_rw[v];
        const double *c = bk.coef_table + 5 * (size_t)coef_row;
        bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
        bq.z0 = bk.z0[v];
        bq.z1 = bk.z1[v];
    }
    const bool cutoff_lfo = FILT && bk.lfo_par != nullptr;  // uniform over the launch
    double lfo_ph = cutoff_lfo ? bk.lfo_phase[v] : 0.;

    // Tremolo (effects.py:100-133): rate, depth, duty cycle, one-pole smoothing filter
    double trem_rate = 0., trem_depth = 0., trem_edge = 0., trem_b0 = 0., trem_a1 = 0.;
    double trem_phase = 0., trem_z = 0.;
    if constexpr (TREM >= 0) {
        const double *par = bk.fx_par + (size_t)((FIRST_STAGE + TREM) * kFxSlots) * n + v;
        trem_rate = par[0], trem_depth = par[n], trem_edge = kTau * par[2 * n];
        trem_b0 = par[3 * n], trem_a1 = par[4 * n];
        trem_phase = bk.fx_state[(size_t)(2 * (FIRST_STAGE + TREM)) * n + v];
        trem_z = bk.fx_state[(size_t)(2 * (FIRST_STAGE + TREM) + 1) * n + v];
    }
    const double trem_floor = 1. - trem_depth;
    // RingModulator (effects.py:618-662): an Lfo (its wave function per voice) sweeps the
    // frequency of a sine modulator; frequency, rate, amount, dryWet, Lfo wave
    double ring_freq = 0., ring_rate = 0., ring_amount = 0., ring_dw = 0.;
    double ring_lfo_phase = 0., ring_phase = 0.;
    int ring_wave = 0;
    if constexpr (RING >= 0) {
        const double *par = bk.fx_par + (size_t)((FIRST_STAGE + RING) * kFxSlots) * n + v;
        ring_freq = par[0], ring_rate = par[n], ring_amount = par[2 * n], ring_dw = par[3 * n];
        ring_wave = (int)par[4 * n];
        ring_lfo_phase = bk.fx_state[(size_t)(2 * (FIRST_STAGE + RING)) * n + v];
        ring_phase = bk.fx_state[(size_t)(2 * (FIRST_STAGE + RING) + 1) * n + v];
    }
    const double ring_dry = 1. - ring_dw;
    const int first_ring_wave = __builtin_amdgcn_readfirstlane(ring_wave);
    const int uniform_ring_wave = __all(ring_wave == first_ring_wave) ? first_ring_wave : -1;
    // Delay (effects.py:136-174): lines tiled by absolute time (bank_kernels.h delay_read8);
    // the position is carried only as the state the reference's RingBuffer would report
    int delay_pos = 0, delay_len = 1;
    double delay_fb = 0., delay_dw = 0.;
    const int delay_tiles = bk.fx_ring_tiles;
    if constexpr (DELAY >= 0) {
        const double *par = bk.fx_par + (size_t)((FIRST_STAGE + DELAY) * kFxSlots) * n + v;
        delay_pos = (int)bk.fx_state[(size_t)(2 * (FIRST_STAGE + DELAY)) * n + v];
        delay_len = (int)par[0];
        delay_fb = par[n], delay_dw = par[2 * n];
    }
    const double delay_dry = 1. - delay_dw;
    // Gain stages (effects.py:58-70): one parameter each
    // Element-wise stages (the FX_ELEM slot of a program stands for any of them: Gain,
    // Transformer, TanhDistortion, OctaveDistortion, Subsampler, Bitcrusher -- effects.py:58-70,
    // 388-483, 522-561): their kind is a bank constant, chosen by a scalar jump per group
    double gain_par[NSTAGE], gain_par1[NSTAGE];
    int elem_kind[NSTAGE];
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        const bool elem = fx_program_kind(PROG, s) == FX_GAIN || fx_program_kind(PROG, s) == FX_ELEM;
        gain_par[s] = elem ? bk.fx_par[(size_t)((FIRST_STAGE + s) * kFxSlots) * n + v] : 0.;
        gain_par1[s] = elem ? bk.fx_par[(size_t)((FIRST_STAGE + s) * kFxSlots + 1) * n + v] : 0.;
        elem_kind[s] = elem ? (int)bk.fx_kind[FIRST_STAGE + s] : 0;
    }
    double held = 0.;   // Subsampler: the sample being repeated (at most one per chain)
    double wstash[G] = {};   // Delay, 16-sample tiles: the first half of the tile being written
    // Delay: the lane carries the older group of its two-group window (delay_window8)
    constexpr bool CARRY = DELAY >= 0 && KLMI_DELAY_CARRY;
    constexpr int GPT = kDelayTile / 8;   // time groups per tile
    double dprev[G] = {};
    const int delay_back = delay_len / 8 + 1, delay_shift = 7 - delay_len % 8;
    if constexpr (CARRY) {
        int ga = delay_tile_of_block(clock_base, delay_tiles) * GPT - delay_back;
        ga = ga < 0 ? ga + delay_tiles * GPT : ga;
        delay_load_group8(bk.fx_ring, n, delay_tiles, ga, v, dprev);
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
        const int block_tile = DELAY >= 0 ? delay_tile_of_block(abs_block, delay_tiles) : 0;
        while (next_ev <= abs_block) {  // voices.py:55-64: notes at the block boundary
            const uint32_t kind = bk.ev_kind[ev_i];
            if (kind == 1u) {
                if (ENV)
                    es.gate(ep, true);
                amp = bk.ev_velocity[ev_i];
                freq = bk.ev_frequency[ev_i];
            } else if (kind == 0u) {
                if (ENV)
                    es.gate(ep, false);
            } else if (kind == 2u) {
                if (FILT) {
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

        const double omega = kTau * freq;  // oscillators.py:60  (TAU * frequency) * t
        const double start = phase;
        const double t_end = c_time[kBlock];
        // wave-uniform: every phase argument of this block is >= 0 and < 1e9 (fast remainder)
        const bool fast_osc = uniform_wave >= W_SINE && uniform_wave <= W_PWM &&
                              __all(omega >= 0. && start >= 0. && omega * t_end + start < 1e9);
        const double trem_omega = kTau * trem_rate;   // effects.py:117-133: kTau * rate * t + start
        const double trem_start = trem_phase;
        const bool fast_trem =
            TREM >= 0 && __all(trem_omega >= 0. && trem_start >= 0. && trem_omega * t_end + trem_start < 1e9);
        const double ring_omega = kTau * ring_rate;    // the Lfo: Oscillator.sample, scalar rate
        const double ring_lfo_start = ring_lfo_phase, ring_start = ring_phase;
        double ring_running = 0.;                      // np.cumsum of the modulator's frequencies
        // wave-uniform: the Lfo's and the modulator's phase arguments of this block are >= 0 and
        // < 1e9 (the modulator's frequency ring_freq + amount * [-1, 1] never turns negative)
        const bool fast_ring =
            RING >= 0 && uniform_ring_wave >= W_SINE && uniform_ring_wave <= W_TRIANGLE &&
            __all(ring_omega >= 0. && ring_lfo_start >= 0. && ring_omega * t_end + ring_lfo_start < 1e9 &&
                  ring_freq - fabs(ring_amount) >= 0. && ring_start >= 0. &&
                  (kTau * kDt) * kBlock * (ring_freq + fabs(ring_amount)) + ring_start < 1e9);

        // ---- region 1: eight oscillator samples ------------------------------------------
        auto osc_group = [&](const Time8 &tt, int sample0, double (&osc)[G]) {
            if (uniform_wave == W_ONE) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    osc[j] = 1.;
                return;
            }
            if (uniform_wave == W_EXTERNAL) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    osc[j] = bk.ext_input[(size_t)(block_base * kBlock + sample0 + j) * n + v];
                return;
            }
            double arg[G];
#pragma unroll
            for (int j = 0; j < G; ++j)
                arg[j] = omega * tt.t[j] + start;  // oscillators.py:60
            if (fast_osc) {
                double ph[G];
                uint32_t hi_max = 0;
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const double k = floor(arg[j] * kInvTau);
                    ph[j] = __builtin_fma(-k, kTau, arg[j]);
                    hi_max = max(hi_max, (uint32_t)__double2hiint(ph[j]));
                }
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord) != 0, 0)) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        ph[j] = fmod_tau_fast(arg[j]);
                }
                switch (uniform_wave) {
                case W_SINE:
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        osc[j] = wave_eval<true>(W_SINE, ph[j], duty_edge);
                    break;
                case W_SQUARE:
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        osc[j] = wave_eval<true>(W_SQUARE, ph[j], duty_edge);
                    break;
                case W_SAWTOOTH:
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        osc[j] = wave_eval<true>(W_SAWTOOTH, ph[j], duty_edge);
                    break;
                case W_TRIANGLE:
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        osc[j] = wave_eval<true>(W_TRIANGLE, ph[j], duty_edge);
                    break;
                default:
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        osc[j] = wave_eval<true>(W_PWM, ph[j], duty_edge);
                    break;
                }
            } else {
                // mixed wavefronts, negative or huge phases: the exact form, lane by lane
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (wave_id == W_ONE)
                        osc[j] = 1.;
                    else if (wave_id == W_EXTERNAL)
                        osc[j] = bk.ext_input[(size_t)(block_base * kBlock + sample0 + j) * n + v];
                    else
                        osc[j] = wave_eval<false>(wave_id, np_mod_tau(arg[j]), duty_edge);
                }
            }
        };

        for (int c = 0; c < NCHUNK; ++c) {
            const int s0 = b * kBlock + c * CH;  // sample offset inside this launch
            const Time8 *times = reinterpret_cast<const Time8 *>(c_time) + ((c * CH) >> 3);
            for (int i8 = 0; i8 < CH; i8 += G) {
                const Time8 tt = times[i8 >> 3];
                // the Delay's eight delayed samples, asked for first
                double old[G] = {}, dcur[G] = {};
                int ttile = 0, tsub = 0;   // wave-uniform: time tile of this group, group in the tile
                if constexpr (DELAY >= 0) {
                    ttile = block_tile + (c * CH + i8) / kDelayTile;
                    ttile = ttile >= delay_tiles ? ttile - delay_tiles : ttile;
                    tsub = ((c * CH + i8) % kDelayTile) >> 3;
                    if constexpr (CARRY) {
                        int gb = ttile * GPT + tsub - delay_back + 1;
                        gb = gb < 0 ? gb + delay_tiles * GPT : gb;
                        delay_load_group8(bk.fx_ring, n, delay_tiles, gb, v, dcur);
                    } else {
                        delay_read8(bk.fx_ring, n, delay_tiles, ttile, tsub, delay_len, v, old);
                    }
                }

                double osc[G];
                osc_group(tt, s0 + i8, osc);

                // ---- region 2: the serial recurrences, side by side ------------------------
                double y[G];
                unsigned long long redo = 0;
                double vv = 0.;
                if (ENV) {
                    // v -> fl(base + fl(v*coef)) is monotone in v: if any of the 8 values
                    // crossed the stage's limit, the last one did (see voice_kernel)
                    vv = es.value;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        vv = es.base + vv * es.coef;           // _envelope.c:579-624
                        y[j] = amp * osc[j] * vv;              // voices.py:73 (amplitude*osc)*env
                    }
                    redo = __builtin_amdgcn_ballot_w64(es.reached(vv));
                } else {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        y[j] = osc[j];
                }
                Biquad sp = bq;
                if (FILT) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        y[j] = sp.step(y[j]);
                }
                // the Tremolo's gain for these samples does not depend on the signal:
                // Lfo (PWM wave) -> one-pole lfilter([b0], [1, a1]) -> depth blend
                double tg[G];
                double tz = trem_z;
                bool trem_exact = false;
                if constexpr (TREM >= 0) {
                    uint32_t hi_max = 0;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const double pa = trem_omega * tt.t[j] + trem_start;
                        const double k = floor(pa * kInvTau);
                        const double tp = __builtin_fma(-k, kTau, pa);
                        hi_max = max(hi_max, (uint32_t)__double2hiint(tp));
                        const double pwm = (tp < trem_edge) ? 1. : -1.;
                        const double yt = tz + trem_b0 * pwm;
                        tz = 0. * pwm - trem_a1 * yt;
                        tg[j] = trem_floor + trem_depth * (.5 + .5 * yt);
                    }
                    trem_exact = !fast_trem || __builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord) != 0;
                }

                // the RingModulator's sine does not depend on the signal either: Lfo -> frequency
                // -> running sum -> phase (the array branch of sample_phase, oscillators.py:61-64)
                double rm[G];
                if constexpr (RING >= 0) {
                    const bool first_group = c == 0 && i8 == 0;
                    const double running0 = ring_running;
                    bool exact = !fast_ring;
                    if (!exact) {
                        uint32_t hi_max = 0;
                        double lp[G], f[G];
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const double la = ring_omega * tt.t[j] + ring_lfo_start;
                            const double k = floor(la * kInvTau);
                            lp[j] = __builtin_fma(-k, kTau, la);
                            hi_max = max(hi_max, (uint32_t)__double2hiint(lp[j]));
                        }
                        switch (uniform_ring_wave) {
                        case W_SINE:
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                f[j] = wave_eval<true>(W_SINE, lp[j], kPi);
                            break;
                        case W_SQUARE:
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                f[j] = wave_eval<true>(W_SQUARE, lp[j], kPi);
                            break;
                        case W_SAWTOOTH:
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                f[j] = wave_eval<true>(W_SAWTOOTH, lp[j], kPi);
                            break;
                        default:
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                f[j] = wave_eval<true>(W_TRIANGLE, lp[j], kPi);
                            break;
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const double fj = ring_freq + ring_amount * f[j];
                            const bool first_sample = first_group && j == 0;
                            const double arg = first_sample ? ring_start
                                                            : (kTau * kDt) * ring_running + ring_start;
                            const double k = floor(arg * kInvTau);
                            const double ph = __builtin_fma(-k, kTau, arg);
                            hi_max = max(hi_max, (uint32_t)__double2hiint(ph));
                            ring_running = first_sample ? fj : ring_running + fj;
                            rm[j] = sin_wrapped(ph);
                        }
                        exact = __builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord) != 0;
                    }
                    if (__builtin_expect(exact, 0)) {
                        ring_running = running0;
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const double lp = np_mod_tau(ring_omega * tt.t[j] + ring_lfo_start);
                            const double fj = ring_freq + ring_amount * wave_eval<true>(ring_wave, lp, kPi);
                            const bool first_sample = first_group && j == 0;
                            const double arg = first_sample ? ring_start
                                                            : (kTau * kDt) * ring_running + ring_start;
                            ring_running = first_sample ? fj : ring_running + fj;
                            rm[j] = wave_eval<true>(W_SINE, np_mod_tau(arg), kPi);
                        }
                    }
                }

                if (__builtin_expect(redo != 0, 0)) {
                    // some lane's envelope switches stage in these 8 samples: the exact state
                    // machine, from the uncommitted state
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        double x = amp * osc[j] * es.step(ep);
                        if (FILT)
                            x = bq.step(x);
                        y[j] = x;
                    }
                } else {
                    if (ENV)
                        es.value = vv;
                    if (FILT)
                        bq.z0 = sp.z0, bq.z1 = sp.z1;
                }
                if constexpr (TREM >= 0) {
                    if (__builtin_expect(trem_exact, 0)) {
                        tz = trem_z;
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const double tp = np_mod_tau(trem_omega * tt.t[j] + trem_start);
                            const double pwm = (tp < trem_edge) ? 1. : -1.;
                            const double yt = tz + trem_b0 * pwm;
                            tz = 0. * pwm - trem_a1 * yt;
                            tg[j] = trem_floor + trem_depth * (.5 + .5 * yt);
                        }
                    }
                    trem_z = tz;
                }

                // ---- the stages, in patch order ---------------------------------------------
#pragma unroll
                for (int s = 0; s < NSTAGE; ++s) {
                    const int kind = fx_program_kind(PROG, s);   // folds: PROG and s are constants
                    if (kind == FX_TREMOLO) {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            y[j] = tg[j] * y[j];
                    } else if (kind == FX_RINGMOD) {   // math.py:68-75 blend(x, am * x, dryWet)
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            y[j] = ring_dry * y[j] + ring_dw * (rm[j] * y[j]);
                    } else if (kind == FX_GAIN) {      // effects.py:67-70
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            y[j] = gain_par[s] * y[j];
                    } else if (kind == FX_ELEM) {      // an element-wise stage (wave-uniform kind)
                        const double p0 = gain_par[s];
                        switch (elem_kind[s]) {
                        case FX_TRANSFORM:              // effects.py:558-561
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                y[j] = p0 * y[j] + gain_par1[s];
                            break;
                        case FX_TANH:                   // effects.py:480-483
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                y[j] = tanh(p0 * y[j]);
                            break;
                        case FX_OCTAVE:                 // effects.py:460-462
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                y[j] = 2. * (y[j] * y[j]) - 1.;
                            break;
                        case FX_SUBSAMPLE: {            // effects.py:400-415: np.repeat(x[::skip], skip)
                            const int mask = (int)p0 - 1;
#pragma unroll
                            for (int j = 0; j < G; ++j) {
                                held = ((c * CH + i8 + j) & mask) == 0 ? y[j] : held;
                                y[j] = held;
                            }
                            break;
                        }
                        case FX_BITCRUSH: {             // effects.py:434-447, wavfile.py:14-30
                            const int bits = (int)p0;
#pragma unroll
                            for (int j = 0; j < G; ++j) {
                                const int q = (int)(short)(int)(32767. * y[j]);  // astype(int16)
                                y[j] = (double)(int)(short)((q >> bits) << bits) * (1. / 32768.);
                            }
                            break;
                        }
                        default:                        // Gain, effects.py:67-70
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                y[j] = p0 * y[j];
                            break;
                        }
                    } else if (kind == FX_DELAY) {     // effects.py:166-174: wet = new + feedback * delayed
                        double w[G];
                        if constexpr (CARRY) {
                            delay_window8(dprev, dcur, delay_shift, old);
#pragma unroll
                            for (int j = 0; j < G; ++j)
                                dprev[j] = dcur[j];
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            w[j] = y[j] + delay_fb * old[j];
                        if constexpr (kDelayTile == 16) {
                            // the first half of a tile waits for the second (wave-uniform)
                            if (tsub == 0) {
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    wstash[j] = w[j];
                            } else if (live) {
                                delay_write16(bk.fx_ring, n, delay_tiles, ttile, v, wstash, w);
                            }
                        } else if (live) {
                            delay_write8(bk.fx_ring, n, delay_tiles, ttile, tsub, v, w);
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            y[j] = delay_dry * y[j] + delay_dw * w[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < G; ++j)
                    emit_sample<MODE>(tile, i8 + j, lane, y[j], gl, gr, live, out, s0, count, local);
            }
            flush_chunk<MODE>(tile, lane, wave, b, c, n_blocks, count, span_total, s0, partial, out);
        }
        // next start phases: sample 256 of this block (oscillators.py:66-67)
        if (wave_id != W_EXTERNAL && wave_id != W_ONE)
            phase = np_mod_tau(omega * t_end + start);
        if constexpr (TREM >= 0)
            trem_phase = np_mod_tau(trem_omega * t_end + trem_start);
        if constexpr (RING >= 0) {
            ring_lfo_phase = np_mod_tau(ring_omega * t_end + ring_lfo_start);
            ring_phase = np_mod_tau((kTau * kDt) * ring_running + ring_start);
        }
        if constexpr (DELAY >= 0)
            delay_pos = (delay_pos + kBlock) % delay_len;
    }

    if (live) {
        bk.phase[v] = phase;
        bk.frequency_rw[v] = freq;
        bk.amplitude_rw[v] = amp;
        if (ENV) {
            bk.env_value[v] = es.value;
            bk.env_state[v] = (uint32_t)es.stage | (es.enabled ? 0x100u : 0u);
        }
        if (FILT) {
            bk.z0[v] = bq.z0;
            bk.z1[v] = bq.z1;
            bk.coef_index_rw[v] = coef_row;
            if (cutoff_lfo)
                bk.lfo_phase[v] = lfo_ph;
        }
        if constexpr (TREM >= 0) {
            bk.fx_state[(size_t)(2 * (FIRST_STAGE + TREM)) * n + v] = trem_phase;
            bk.fx_state[(size_t)(2 * (FIRST_STAGE + TREM) + 1) * n + v] = trem_z;
        }
        if constexpr (DELAY >= 0)
            bk.fx_state[(size_t)(2 * (FIRST_STAGE + DELAY)) * n + v] = (double)delay_pos;
        if constexpr (RING >= 0) {
            bk.fx_state[(size_t)(2 * (FIRST_STAGE + RING)) * n + v] = ring_lfo_phase;
            bk.fx_state[(size_t)(2 * (FIRST_STAGE + RING) + 1) * n + v] = ring_phase;
        }
        bk.ev_cursor[v] = ev_i - bk.ev_offset[v];
    }
}

}  // namespace klmi
