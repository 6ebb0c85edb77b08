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
// serial anyway, so the running frequency sum needs no cross-lane scan.  Samples are
// handled in groups of eight (one scalar load of the time points, speculative remainders
// and envelope with one check per group); the arithmetic per sample is the reference's.
// Chains with effect stages run stage-major over a wave-private LDS strip (see HAS_FX).
// Envelope and filter are run-time options so that only the output mode and HAS_FX are
// template parameters.
#pragma once
#include "bank_kernels.h"
#include <type_traits>

namespace klmi {

// G wave-function values with the dispatch OUTSIDE the sample loop: `sel` is the
// wavefront's common wave id (scalar: one jump per group, straight-line code per case)
// or -1, in which case every lane switches on its own `lane_wave`.
// WRAPPED: 0 <= p < TAU (a wrapped phase); otherwise any finite phase (FM).
template <bool WRAPPED, bool BD, int G>
__device__ __forceinline__ void wave_block(int sel, int lane_wave, const double (&p)[G],
                                           double duty_edge, double (&out)[G])
{
    auto eval = [&](int w, double ph) {
        if constexpr (WRAPPED)
            return wave_eval<true>(w, ph, duty_edge);
        else
            return wave_eval_any<BD>(w, ph, duty_edge);
    };
    switch (sel) {
    case W_SINE:
#pragma unroll
        for (int j = 0; j < G; ++j)
            out[j] = eval(W_SINE, p[j]);
        break;
    case W_SQUARE:
#pragma unroll
        for (int j = 0; j < G; ++j)
            out[j] = eval(W_SQUARE, p[j]);
        break;
    case W_SAWTOOTH:
#pragma unroll
        for (int j = 0; j < G; ++j)
            out[j] = eval(W_SAWTOOTH, p[j]);
        break;
    case W_TRIANGLE:
#pragma unroll
        for (int j = 0; j < G; ++j)
            out[j] = eval(W_TRIANGLE, p[j]);
        break;
    case W_PWM:
#pragma unroll
        for (int j = 0; j < G; ++j)
            out[j] = eval(W_PWM, p[j]);
        break;
    default:
#pragma unroll
        for (int j = 0; j < G; ++j)
            out[j] = eval(lane_wave, p[j]);
        break;
    }
}

// G remainders mod TAU.  FAST (wave-uniform: every argument is finite, >= 0 and < 1e9):
// speculate r = fma(-floor(a / TAU), TAU, a) -- exact unless the quotient estimate is off by
// one -- with ONE range check over the group's high words, as voice_kernel does; a hit in any
// lane redoes the group with the repairing form.  Otherwise the exact per-element remainder.
template <bool BD, int G>
__device__ __forceinline__ void mod_tau_group(bool fast, const double (&a)[G], double (&p)[G])
{
    if (BD && fast) {
        uint32_t hi_max = 0;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const double k = floor(a[j] * kInvTau);
            p[j] = __builtin_fma(-k, kTau, a[j]);
            hi_max = max(hi_max, (uint32_t)__double2hiint(p[j]));
        }
        if (__builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord) == 0)
            return;
    }
#pragma unroll
    for (int j = 0; j < G; ++j)
        p[j] = np_mod_tau_sel<BD>(a[j]);
}

// FmOscillator's carrier (oscillators.py:220): wave(q) for G phases q of any sign, bounded
// (|q| < 1e9), one wave function for the whole wavefront.  The branch-free remainder
// np_mod_tau_bounded costs three float64 compares + selects per call (quarter-rate
// instructions: the most expensive part of the FM sample).  Here the remainder of q (sawtooth,
// sine) or q + PI (square, triangle: math.py:9-11 wrap) is speculated as
// r = fma(-floor(a / TAU), TAU, a) -- the same real number x - floor(x/TAU) TAU that NumPy's
// fmod-then-sign-fix yields, rounded once, for either sign -- with ONE range check over the
// group's high words; a quotient estimate that is off by one, or an r that rounds to TAU,
// sends the group to the exact form.  Returns false when `sel` is not one of the four.
template <bool BD, int G>
__device__ __forceinline__ bool fm_wave_group(int sel, const double (&q)[G], double (&x)[G])
{
    if (!BD || sel < W_SINE || sel > W_TRIANGLE)
        return false;
    const bool shifted = sel == W_SQUARE || sel == W_TRIANGLE;
    double r[G];
    uint32_t hi_max = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const double a = shifted ? q[j] + kPi : q[j];
        const double k = floor(a * kInvTau);
        r[j] = __builtin_fma(-k, kTau, a);
        hi_max = max(hi_max, (uint32_t)__double2hiint(r[j]));
    }
    if (__builtin_amdgcn_ballot_w64(hi_max >= kTauHighWord) != 0)
        return false;   // some lane: the exact path for the whole group
    switch (sel) {
    case W_SINE:
#pragma unroll
        for (int j = 0; j < G; ++j)
            x[j] = sin_wrapped(r[j]);
        break;
    case W_SQUARE:
#pragma unroll
        for (int j = 0; j < G; ++j)
            x[j] = (r[j] - kPi >= 0.) ? 1. : -1.;
        break;
    case W_SAWTOOTH:
#pragma unroll
        for (int j = 0; j < G; ++j)
            x[j] = div_pi(r[j]) - 1.;
        break;
    default:
#pragma unroll
        for (int j = 0; j < G; ++j) {
            double t4 = div_tau(4. * (r[j] - kPi));
            t4 = t4 < 0. ? t4 + 4. : t4;
            x[j] = 1. - fabs(t4 - 2.);
        }
        break;
    }
    return true;
}

// two float64 at 8-byte alignment (one 16-byte access)
struct __attribute__((packed, aligned(8))) Double2U {
    double a, b;
};

// G consecutive time points of a block (G = 4 or 8)
template <int G> struct TimeGroup {
    double t[G];
};
template <int G>
__device__ __forceinline__ TimeGroup<G> load_times(const double *times, int i)
{
    TimeGroup<G> r;   // wave-uniform index: one scalar load of G doubles
#pragma unroll
    for (int j = 0; j < G; ++j)
        r.t[j] = times[i + j];
    return r;
}

// HAS_FX: the chain has effect stages (bk.n_fx > 0).  Without them all stage code and
// state is compiled out (232 VGPRs, nothing spilled).  With them the stage program's state
// fills the register file (256 VGPRs + AGPR copies); groups of 4 instead of 8 were measured
// slower (per-group dispatch), so only rarely used parameters are fetched where used.
template <int MODE, bool HAS_FX>
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
    constexpr int G = 8;
    using TimeG = TimeGroup<G>;
    const int n_fx = HAS_FX ? bk.n_fx : 0;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    T *tile = reinterpret_cast<T *>(smem_raw) + (size_t)wave * (PLANES ? PLANES : 1) * CH * TT::stride;
    // HAS_FX: the wave's float64 strip of SUB samples x 64 lanes behind the tiles
    constexpr int SUB = kFxStrip < CH ? kFxStrip : CH;
    double *strip = reinterpret_cast<double *>(smem_raw + tile_bytes<MODE>()) + (size_t)wave * SUB * 64;

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

    // wave functions shared by the whole wavefront (scalar dispatch), or -1
    const int first_wave = __builtin_amdgcn_readfirstlane(wave_id);
    const int uniform_wave = __all(wave_id == first_wave) ? first_wave : -1;
    const int first_mod_wave = __builtin_amdgcn_readfirstlane(mod_wave);
    const int uniform_mod_wave = __all(mod_wave == first_mod_wave) ? first_mod_wave : -1;

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
    int trem_stage = -1, delay_stage = -1, ring_stage = -1;
#pragma unroll
    for (int s = 0; s < 8; ++s) {   // constant indices: the descriptor stays in scalar registers
        if (s >= n_fx)
            break;
        if (bk.fx_kind[s] == FX_TREMOLO)
            trem_stage = s;
        if (bk.fx_kind[s] == FX_DELAY)
            delay_stage = s;
        if (bk.fx_kind[s] == FX_RINGMOD)
            ring_stage = s;
    }
    // RingModulator (effects.py:618-662): its Lfo and its sine modulator
    double ring_freq = 0., ring_rate = 0., ring_amount = 0., ring_dw = 0.;
    double ring_lfo_phase = 0., ring_phase = 0.;
    int ring_wave = 0;
    if (ring_stage >= 0) {
        const double *par = bk.fx_par + (size_t)(ring_stage * kFxSlots) * n + v;
        ring_freq = par[0], ring_rate = par[n], ring_amount = par[2 * n], ring_dw = par[3 * n];
        ring_wave = (int)par[4 * n];
        ring_lfo_phase = bk.fx_state[(size_t)(2 * ring_stage) * n + v];
        ring_phase = bk.fx_state[(size_t)(2 * ring_stage + 1) * n + v];
    }
    double held = 0.;   // Subsampler: the sample being repeated (at most one per chain)
    double trem_phase = 0., trem_z = 0.;
    if (trem_stage >= 0) {
        trem_phase = bk.fx_state[(size_t)(2 * trem_stage) * n + v];
        trem_z = bk.fx_state[(size_t)(2 * trem_stage + 1) * n + v];
    }
    double trem_rate = 0., trem_depth = 0., trem_duty = 0., trem_b0 = 0., trem_a1 = 0.;
    if (trem_stage >= 0) {
        const double *par = bk.fx_par + (size_t)(trem_stage * kFxSlots) * n + v;
        trem_rate = par[0], trem_depth = par[n], trem_duty = par[2 * n];
        trem_b0 = par[3 * n], trem_a1 = par[4 * n];
    }
    int delay_pos = 0, delay_len = 1;
    double delay_fb = 0., delay_dw = 0.;
    // block-major layout only (a line shorter than a buffer): the voice's RingBuffer array
    double *delay_line = bk.fx_ring + (size_t)v * bk.fx_ring_stride;
    if (delay_stage >= 0) {
        const double *par = bk.fx_par + (size_t)(delay_stage * kFxSlots) * n + v;
        delay_pos = (int)bk.fx_state[(size_t)(2 * delay_stage) * n + v];
        delay_len = (int)par[0];
        delay_fb = par[n], delay_dw = par[2 * n];
    }
    // Delay lines shorter than one buffer (bk.fx_block_major, set for the whole bank when any
    // voice has one): RingBuffer.extend's tail and head copies then land inside the window
    // Delay.update reads back (ring_buffer.py:88-109, effects.py:166-174), so the stage's
    // output at sample i can be a value from LATER in the same buffer.  The buffer is then
    // handled in two passes around a block-level Delay: pass 1 runs the voice and the stages
    // in front of the Delay for all 256 samples into a per-wave staging area in global memory
    // ([256][64 lanes] float64, two planes), delay_block() does what the reference does with
    // whole buffers, pass 3 runs the remaining stages and the mix-down.  Lines are the
    // RingBuffer's data array here: `length` slots and a tail of one buffer.
    const bool block_major = HAS_FX && bk.fx_block_major != 0 && delay_stage >= 0;
    double *stage_new = nullptr, *stage_val = nullptr;
    if (block_major) {
        stage_new = bk.fx_stage + (size_t)(blockIdx.x * kWavesPerGroup + wave) * 2 * kBlock * 64;
        stage_val = stage_new + kBlock * 64;
    }
    auto delay_block = [&]() {
        const int start = delay_pos, stop = start + kBlock;
        for (int i = 0; i < kBlock; ++i)     // new + feedback * peek()
            stage_val[i * 64 + lane] = stage_new[i * 64 + lane] + delay_fb * delay_line[start + i];
        if (live) {
            for (int i = 0; i < kBlock; ++i)                 // ring_buffer.py:95-97
                delay_line[start + i] = stage_val[i * 64 + lane];
            if (start < kBlock) {                            // :98-102, the tail
                const int a = start + delay_len, cnt = kBlock - start;
                for (int k = 0; k < cnt; ++k)
                    delay_line[a + k] = stage_val[k * 64 + lane];
            }
            if (stop > delay_len) {                          // :104-108, the head
                const int a = start > delay_len ? start - delay_len : 0;
                const int cnt = stop - delay_len - a;
                for (int k = 0; k < cnt; ++k)
                    delay_line[a + k] = stage_val[(kBlock - cnt + k) * 64 + lane];
            }
        }
        for (int i = 0; i < kBlock; ++i) {   // blend(new, the window after the writes)
            const double nw = stage_new[i * 64 + lane];
            stage_new[i * 64 + lane] = (1. - delay_dw) * nw + delay_dw * delay_line[start + i];
        }
    };
    unsigned long long fx_kinds = 0;   // stage kinds, one byte each (scalar)
#pragma unroll
    for (int s = 0; s < 8; ++s)
        fx_kinds |= (unsigned long long)(s < n_fx ? bk.fx_kind[s] : 0) << (8 * s);

    double gl = 0., gr = 0.;
    if (MIX) {
        gl = live ? bk.gain_l[v] : 0.;
        if (MODE == OUT_MIX_STEREO)
            gr = live ? bk.gain_r[v] : 0.;
    }

    uint32_t ev_i = bk.ev_offset[v] + bk.ev_cursor[v];
    const uint32_t ev_end = bk.ev_offset[v + 1];
    uint32_t next_ev = (live && ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;

    // polyphonic banks: the voice's place in its instrument and the instrument's notes
    const int poly = bk.poly;
    int slot = 0, rr = 0, cur_pitch = 0;
    uint32_t nt_i = 0, nt_end = 0, next_nt = 0xFFFFFFFFu;
    if (poly > 0) {
        const int inst = v / poly;
        slot = v - inst * poly;
        rr = (int)bk.round_robin[v];
        cur_pitch = bk.current_pitch[v];
        nt_i = bk.nt_offset[inst] + bk.nt_cursor[v];
        nt_end = bk.nt_offset[inst + 1];
        next_nt = (live && nt_i < nt_end) ? bk.nt_block[nt_i] : 0xFFFFFFFFu;
    }

    for (int b = 0; b < n_blocks; ++b) {
        const uint32_t abs_block = clock_base + (uint32_t)b;
        // time tile of the block's first sample (long delay lines: bank_kernels.h delay_read8)
        const int block_tile = (HAS_FX && delay_stage >= 0 && bk.fx_ring_tiles > 0)
                                   ? delay_tile_of_block(abs_block, bk.fx_ring_tiles) : 0;
        // Voice.active (voices.py:45-53) is asked before the notes are processed
        const bool was_active = !has_env || es.stage != 0;
        bool had_note = false;
        while (next_ev <= abs_block) {  // voices.py:55-64
            const uint32_t kind = bk.ev_kind[ev_i];
            if (kind == 1u) {
                if (has_env)
                    es.gate(ep, true);
                amp = bk.ev_velocity[ev_i];
                freq = bk.ev_frequency[ev_i];
                had_note = true;
            } else if (kind == 0u) {
                if (has_env)
                    es.gate(ep, false);
                had_note = true;
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
        if (poly > 0) {
            // PolyphonicSynthesizer.process_note (synthesizer.py:200-209): every voice reads
            // its instrument's notes of this buffer and picks out its own.  Note-offs match
            // the pitch the voice's LAST update left (the synthesizer routes all notes before
            // any voice processes its queue); the voice then handles its notes in order.
            const int start_pitch = cur_pitch;
            while (next_nt <= abs_block) {
                const double velocity = bk.nt_velocity[nt_i];
                const int pitch = bk.nt_pitch[nt_i];
                if (velocity > 0.) {             // messages.py:49-52
                    if (rr == slot) {            // next(self.freeVoice)
                        es.gate(ep, true);
                        amp = velocity;
                        freq = bk.nt_frequency[nt_i];
                        cur_pitch = pitch;
                        had_note = true;
                    }
                    rr = rr + 1 == poly ? 0 : rr + 1;
                } else if (start_pitch == pitch) {
                    es.gate(ep, false);
                    cur_pitch = 0;
                    had_note = true;
                }
                ++nt_i;
                next_nt = (nt_i < nt_end) ? bk.nt_block[nt_i] : 0xFFFFFFFFu;
            }
        }
        // synthesizer.py:177-179, 214-217: an idle voice is not updated.  Its envelope is OFF
        // (samples exactly 0, state unchanged); what has to stay put is the oscillator.
        const bool run = !bk.gated || had_note || was_active;

        if (cutoff_lfo)
            lfo_cutoff_step(bk, v, lfo_ph, coef_row, bq);

        const double omega = kTau * freq;          // (TAU * frequency) * t
        const double start = phase;
        const double mod_omega = kTau * mod_freq;
        const double mod_start = mod_phase;
        double running = 0.;                        // np.cumsum(frequency) within this block
        const double trem_start = trem_phase;
        const double ring_lfo_start = ring_lfo_phase, ring_start = ring_phase;
        double ring_running = 0.;                   // np.cumsum of the modulator's frequencies
        // Wave-uniform: every remainder argument of this block is finite and below 1e9 in
        // magnitude, so the branch-free remainder applies (lane-divergent early-outs and
        // the library fallback cost far more than the operations they save).
        const double t_end = c_time[kBlock];
        const bool bounded_block = __all(
            fabs(omega) * t_end + fabs(start) < 1e9 &&
            fabs(mod_omega) * t_end + fabs(mod_start) < 1e9 && fabs(mod_scale) < 1e8 &&
            (kTau * kDt) * kBlock * (fabs(mod_scale) + fabs(mod_offset)) + fabs(start) < 1e9 &&
            fabs(kTau * trem_rate) * t_end + fabs(trem_start) < 1e9 &&
            fabs(kTau * ring_rate) * t_end + fabs(ring_lfo_start) < 1e9 &&
            (kTau * kDt) * kBlock * (fabs(ring_freq) + fabs(ring_amount)) + fabs(ring_start) < 1e9);
        // ... and which of the closed-form phases are non-negative throughout (fast remainder)
        const bool pos_osc = bounded_block && __all(omega >= 0. && start >= 0.);
        const bool pos_mod = bounded_block && __all(mod_omega >= 0. && mod_start >= 0.);
        const bool pos_trem = bounded_block && __all(trem_rate >= 0. && trem_start >= 0.);
        const bool pos_ring = bounded_block && __all(ring_rate >= 0. && ring_lfo_start >= 0.);

        // Eight oscillator samples (with the modulator's): per sample the reference's
        // operations in the reference's order, but staged over the group so that every
        // dispatch (target, wave functions) happens once per group, not once per sample.
        auto osc_group = [&](auto bounded, const TimeG &tt, bool block_start, int sample0,
                             double (&x)[G]) {
            constexpr bool BD = decltype(bounded)::value;
            double m[G], p[G], arg[G];
            if (target != MT_NONE) {   // Oscillator.sample / Lfo.sample with shape 256
#pragma unroll
                for (int j = 0; j < G; ++j)
                    arg[j] = mod_omega * tt.t[j] + mod_start;
                mod_tau_group<BD, G>(pos_mod, arg, p);
                wave_block<true, BD, G>(uniform_mod_wave, mod_wave, p, kPi, m);
                if (target != MT_FM) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        m[j] = mod_scale * m[j] + mod_offset;
                }
            }
            if (uniform_wave == W_ONE) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    x[j] = 1.;
                return;
            }
            if (uniform_wave == W_EXTERNAL) {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    x[j] = bk.ext_input[(size_t)(block_base * kBlock + sample0 + j) * n + v];
                return;
            }
            if (target == MT_FREQUENCY) {   // oscillators.py:61-64, sequential running sum
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const bool first_sample = block_start && j == 0;
                    p[j] = first_sample ? np_mod_tau_sel<BD>(start)
                                        : np_mod_tau_sel<BD>((kTau * kDt) * running + start);
                    running = first_sample ? m[j] : running + m[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < G; ++j)
                    arg[j] = omega * tt.t[j] + start;
                mod_tau_group<BD, G>(pos_osc, arg, p);
            }
            if (target == MT_FM) {          // oscillators.py:220: any phase
#pragma unroll
                for (int j = 0; j < G; ++j)
                    p[j] = p[j] + mod_scale * m[j];
                if (!fm_wave_group<BD, G>(uniform_wave, p, x))
                    wave_block<false, BD, G>(uniform_wave, wave_id, p, duty_edge, x);
            } else if (target == MT_DUTY) { // oscillators.py:259-260
#pragma unroll
                for (int j = 0; j < G; ++j)
                    x[j] = (p[j] < kTau * m[j]) ? 1. : -1.;
            } else {
                wave_block<true, BD, G>(uniform_wave, wave_id, p, duty_edge, x);
            }
            if (uniform_wave < 0) {         // mixed wavefront: constant and external lanes
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    if (wave_id == W_ONE)
                        x[j] = 1.;
                    else if (wave_id == W_EXTERNAL)
                        x[j] = bk.ext_input[(size_t)(block_base * kBlock + sample0 + j) * n + v];
                }
            }
        };

        // pass 0: the whole chain, chunk by chunk.  Block-major Delay: pass 1 (voice and the
        // stages in front of the Delay -> staging), delay_block(), pass 3 (staging -> the
        // remaining stages -> mix-down)
        const int n_pass = block_major ? 2 : 1;
        for (int pi = 0; pi < n_pass; ++pi) {
        const int pass = block_major ? (pi == 0 ? 1 : 3) : 0;
        const int s_begin = pass == 3 ? delay_stage + 1 : 0;
        const int s_end = pass == 1 ? delay_stage : n_fx;
        for (int c = 0; c < NCHUNK; ++c) {
            const int s0 = b * kBlock + c * CH;
            const double *times = c_time + c * CH;
            // G samples per group: one scalar load of the time points, stage parameters from
            // registers, and the G delayed samples of a Delay stage fetched together (the line
            // is at least one buffer long, so they never depend on this block's writes).
            // Arithmetic per sample is exactly the per-sample reference order.
            // oscillator, modulator and envelope for samples [i8, i8 + G) of the chunk
            auto voice_group = [&](int i8, double (&x)[G]) {
                const TimeG tt = load_times<G>(times, i8);
                // -- oscillator and its modulator ---------------------------------------
                if (bounded_block)
                    osc_group(std::true_type{}, tt, c == 0 && i8 == 0, s0 + i8, x);
                else
                    osc_group(std::false_type{}, tt, c == 0 && i8 == 0, s0 + i8, x);
                // -- envelope: speculate "no stage switch in these G" (monotone recurrence,
                // see voice_kernel); on a hit in any lane the exact state machine runs
                if (has_env) {
                    double e[G];
                    double vv = es.value;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        vv = es.base + vv * es.coef;
                        e[j] = vv;
                    }
                    if (__builtin_amdgcn_ballot_w64(es.reached(vv)) != 0) {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            e[j] = es.step(ep);
                    } else {
                        es.value = vv;
                    }
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        x[j] = amp * x[j] * e[j];
                }
            };
            if constexpr (!HAS_FX) {
                // G samples per group, everything in registers
                for (int i8 = 0; i8 < CH; i8 += G) {
                    double x[G];
                    voice_group(i8, x);
                    if (has_filter) {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            x[j] = bq.step(x[j]);
                    }
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        emit_sample<MODE>(tile, i8 + j, lane, x[j], gl, gr, live, out, s0, count, local);
                }
            } else {
                // Stage-major over sub-chunks of SUB samples that live in the wave's LDS strip
                // (float64 [SUB][64 lanes], lane-contiguous: conflict-free 8-byte accesses):
                // the voice fills the strip, every stage of the wave-uniform program runs its own
                // tight loop over it, the last pass hands the samples to the mix-down tile.  A
                // stage's state is live in its own loop only, so the stage program no longer
                // fights over the register file (the group-major form spent most of its
                // instructions on register copies); the price is two LDS accesses per sample
                // and stage.  The stage kinds were unpacked into scalar registers; a stage's
                // first parameter comes from the (cache-resident) parameter rows.
                auto load_x = [&](int i, double (&x)[G]) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        x[j] = strip[(i + j) * 64 + lane];
                };
                auto store_x = [&](int i, const double (&x)[G]) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        strip[(i + j) * 64 + lane] = x[j];
                };
                for (int half = 0; half < CH; half += SUB) {
                    if (pass != 3) {
                        for (int i8 = half; i8 < half + SUB; i8 += G) {
                            double x[G];
                            voice_group(i8, x);
                            store_x(i8 - half, x);
                        }
                    } else {
                        for (int i = 0; i < SUB; ++i)
                            strip[i * 64 + lane] = stage_new[(c * CH + half + i) * 64 + lane];
                    }
#pragma nounroll
                    for (int s = s_begin; s < s_end; ++s) {
                        const double p0 = bk.fx_par[(size_t)(s * kFxSlots) * n + v];
                        switch ((int)((fx_kinds >> (8 * s)) & 0xFFu)) {
                        case FX_FILTER: {
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    x[j] = bq.step(x[j]);
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_GAIN: {       // effects.py:67-70
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    x[j] = p0 * x[j];
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_TRANSFORM: {  // effects.py:558-561
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
                                const double p1 = bk.fx_par[(size_t)(s * kFxSlots + 1) * n + v];
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    x[j] = p0 * x[j] + p1;
                        
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_TANH: {       // effects.py:480-483
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    x[j] = tanh(p0 * x[j]);
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_OCTAVE: {     // effects.py:460-462
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    x[j] = 2. * (x[j] * x[j]) - 1.;
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_TREMOLO: {  // effects.py:117-133
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
                                const TimeG tt = load_times<G>(times, i8);
                                double pa[G], tp[G];
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    pa[j] = kTau * trem_rate * tt.t[j] + trem_start;
                                if (bounded_block)
                                    mod_tau_group<true, G>(pos_trem, pa, tp);
                                else
                                    mod_tau_group<false, G>(false, pa, tp);
#pragma unroll
                                for (int j = 0; j < G; ++j) {
                                    const double p = tp[j];
                                    const double pwm = (p < kTau * trem_duty) ? 1. : -1.;
                                    const double y = trem_z + trem_b0 * pwm;   // lfilter([b0], [1, a1])
                                    trem_z = 0. * pwm - trem_a1 * y;
                                    x[j] = ((1. - trem_depth) + trem_depth * (.5 + .5 * y)) * x[j];
                                }
                        
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_DELAY: {    // effects.py:166-174 (the blend reads the window after
                                            // the write: wet = new + feedback * delayed)
                            double wstash[G] = {};   // 16-sample tiles: first half of the tile
                            static_assert(SUB % 16 == 0, "whole tiles per strip");
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
                                // lines tiled by absolute time (bank_kernels.h delay_read8);
                                // length >= a buffer, so the 8 reads never see this block's writes
                                int ttile = block_tile + (c * CH + i8) / kDelayTile;
                                ttile = ttile >= bk.fx_ring_tiles ? ttile - bk.fx_ring_tiles : ttile;
                                const int tsub = ((c * CH + i8) % kDelayTile) >> 3;
                                double old[G], w[G];
                                delay_read8(bk.fx_ring, n, bk.fx_ring_tiles, ttile, tsub, delay_len, v, old);
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    w[j] = x[j] + delay_fb * old[j];
                                if constexpr (kDelayTile == 16) {
                                    if (tsub == 0) {
#pragma unroll
                                        for (int j = 0; j < G; ++j)
                                            wstash[j] = w[j];
                                    } else if (live) {
                                        delay_write16(bk.fx_ring, n, bk.fx_ring_tiles, ttile, v, wstash, w);
                                    }
                                } else if (live) {
                                    delay_write8(bk.fx_ring, n, bk.fx_ring_tiles, ttile, tsub, v, w);
                                }
#pragma unroll
                                for (int j = 0; j < G; ++j)
                                    x[j] = (1. - delay_dw) * x[j] + delay_dw * w[j];
                        
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_SUBSAMPLE: {  // effects.py:400-415: np.repeat(x[::skip], skip)
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
                                const int mask = (int)p0 - 1;
#pragma unroll
                                for (int j = 0; j < G; ++j) {
                                    held = ((c * CH + i8 + j) & mask) == 0 ? x[j] : held;
                                    x[j] = held;
                                }
                        
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_BITCRUSH: {   // effects.py:434-447, wavfile.py:14-30
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
                                const int bits = (int)p0;
#pragma unroll
                                for (int j = 0; j < G; ++j) {
                                    const int q = (int)(short)(int)(32767. * x[j]);  // astype(int16)
                                    x[j] = (double)(int)(short)((q >> bits) << bits) * (1. / 32768.);
                                }
                        
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        case FX_RINGMOD: {    // effects.py:650-662
                            for (int i8 = half; i8 < half + SUB; i8 += G) {
                                double x[G];
                                load_x(i8 - half, x);
                                const TimeG tt = load_times<G>(times, i8);
                                double f[G], ph[G], am[G];
#pragma unroll
                                for (int j = 0; j < G; ++j)     // the Lfo: Oscillator.sample, scalar rate
                                    am[j] = kTau * ring_rate * tt.t[j] + ring_lfo_start;
                                if (bounded_block)
                                    mod_tau_group<true, G>(pos_ring, am, ph);
                                else
                                    mod_tau_group<false, G>(false, am, ph);
                                wave_block<true, false, G>(-1, ring_wave, ph, kPi, f);
#pragma unroll
                                for (int j = 0; j < G; ++j) {   // modulator: array branch of sample_phase
                                    const double fj = ring_freq + ring_amount * f[j];
                                    const bool first_sample = c == 0 && i8 == 0 && j == 0;
                                    const double arg = first_sample ? ring_start
                                                                    : (kTau * kDt) * ring_running + ring_start;
                                    ph[j] = bounded_block ? np_mod_tau_bounded(arg) : np_mod_tau(arg);
                                    ring_running = first_sample ? fj : ring_running + fj;
                                }
                                wave_block<true, false, G>(W_SINE, W_SINE, ph, kPi, am);
#pragma unroll
                                for (int j = 0; j < G; ++j)     // math.py:68-75 blend(x, am * x, dryWet)
                                    x[j] = (1. - ring_dw) * x[j] + ring_dw * (am[j] * x[j]);
                        
                                store_x(i8 - half, x);
                            }
                            break;
                        }
                        default:
                            break;
                        }
                    }
                    if (pass == 1) {
                        for (int i = 0; i < SUB; ++i)
                            stage_new[(c * CH + half + i) * 64 + lane] = strip[i * 64 + lane];
                        continue;
                    }
                    for (int i8 = half; i8 < half + SUB; i8 += G) {
                        double x[G];
                        load_x(i8 - half, x);
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            emit_sample<MODE>(tile, i8 + j, lane, x[j], gl, gr, live, out, s0, count, local);
                    }
                }
            }
            if (pass != 1)
                flush_chunk<MODE>(tile, lane, wave, b, c, n_blocks, count, span_total, s0, partial, out);
        }
        if (pass == 1)
            delay_block();
        }
        // next start phases: sample 256 of this block
        if (trem_stage >= 0)
            trem_phase = np_mod_tau(kTau * trem_rate * c_time[kBlock] + trem_start);
        if (delay_stage >= 0)
            delay_pos = (delay_pos + kBlock) % delay_len;
        if (ring_stage >= 0) {
            ring_lfo_phase = np_mod_tau(kTau * ring_rate * c_time[kBlock] + ring_lfo_start);
            ring_phase = np_mod_tau((kTau * kDt) * ring_running + ring_start);
        }
        if (target != MT_NONE)
            mod_phase = np_mod_tau(mod_omega * c_time[kBlock] + mod_start);
        if (wave_id != W_ONE && wave_id != W_EXTERNAL) {
            if (target == MT_FREQUENCY)
                phase = np_mod_tau((kTau * kDt) * running + start);
            else
                phase = np_mod_tau(omega * c_time[kBlock] + start);
        }
        if (!run) {
            phase = start;
            mod_phase = mod_start;
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
        if (ring_stage >= 0) {
            bk.fx_state[(size_t)(2 * ring_stage) * n + v] = ring_lfo_phase;
            bk.fx_state[(size_t)(2 * ring_stage + 1) * n + v] = ring_phase;
        }
        bk.ev_cursor[v] = ev_i - bk.ev_offset[v];
        if (poly > 0) {
            bk.nt_cursor[v] = nt_i - bk.nt_offset[v / poly];
            bk.round_robin[v] = (uint32_t)rr;
            bk.current_pitch[v] = cur_pitch;
        }
    }
}

}  // namespace klmi
