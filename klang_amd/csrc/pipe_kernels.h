// pipe_kernels.h -- the voice kernel for banks that do not fill the chip: oscillator,
// serial chain and output run as three pipeline stages on different waves of a workgroup
// (gfx950).
//
// voice_kernel (bank_kernels.h) maps one voice to one lane and walks time serially: 65,536
// voices are one wavefront per SIMD, but a 4,096-voice bank leaves 15 of 16 SIMDs idle and
// the native 20-voice haunting_envelopes patch is a single wavefront.  Only two parts of a
// voice are serial in time: the envelope recurrence (_envelope.c:579-624: its stage
// switches depend on the rounded values) and the biquad (lfilter's z-state).  The
// oscillator is not: inside a buffer its phase is the closed form (TAU*f)*t[n] + start
// (oscillators.py:57-67), and notes land on buffer boundaries (voices.py:55-64).  Neither
// is the output side (gain, rounding, the sum over voices).
//
// A workgroup of 4 waves owns V voices (a power of two, 1..64) and moves through time in
// rounds of 32 samples.  In step t
//   waves 1-2 (128 lanes, 128/V lanes per voice) evaluate amplitude*oscillator for round t,
//             time-parallel, into xbuf[t & 1];
//   wave 0    (lane = voice) runs envelope and biquad over round t-1: xbuf -> ybuf, eight
//             samples at a time as one straight-line region with voice_kernel's
//             monotone-envelope speculation -- the only serial work left, and it waits
//             for nothing but LDS;
//   wave 3    (lane = sample) applies the gains to round t-2 and sums the voices (mix
//             modes: one partial-bus row per workgroup) or writes per-voice rows;
// one workgroup barrier per step.  Each stage keeps the state it needs itself: the
// oscillator lanes follow frequency / amplitude / phase through the events of their
// voice, the chain lane follows gates and cutoffs.  The arithmetic per sample is
// voice_kernel's, operation for operation.
#pragma once
#include "bank_kernels.h"
#include <type_traits>

namespace klmi {

constexpr int kPipeRound = 32;                       // samples per pipeline step
constexpr int kPipeRounds = kBlock / kPipeRound;     // steps per buffer
constexpr int kPipeMaxVoices = 64;                   // per workgroup
// A voice's round in LDS: 32 doubles + 2 of padding.  The chain lane reads and writes its
// row with 16-byte accesses; a row pitch of 68 dwords puts the rows of 16 consecutive lanes
// on disjoint banks.
constexpr int kPipePitch = kPipeRound + 2;

__host__ __device__ constexpr size_t pipe_lds_bytes(int voices)
{
    // xbuf[2][V][pitch] + ybuf[2][V][pitch] + gains[2][V]
    return sizeof(double) * ((size_t)4 * kPipePitch * voices + 2 * (size_t)voices);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the
// wave's global stores to be acknowledged (a release fence over every address space): the
// output stage's stores to the partial bus would put a memory round trip into every
// pipeline step, although nothing in this kernel reads them.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// grid.x workgroups of 4 waves: the first n_big own V = 2^voices_log2 voices each, the others
// half as many (pipe_groups in pipe_tu.hip: a grid that is a whole number of workgroups per
// compute unit -- 20,480 voices on 256 CUs are 512 x 32 + 256 x 16, three workgroups and 80
// voices on every CU instead of 640 x 32 with three on half of them).
// partial: [gridDim.x][n_blocks*planes*256] float32 (mix modes)
// out:     [count][span_total] per-voice rows (OUT_VOICES_F32 / OUT_VOICES_F64)
template <bool ENV, bool FILT, int MODE>
__global__ __launch_bounds__(kGroup) void voice_pipe_kernel(BankView bk, int first, int count,
                                                            int block_base, uint32_t clock_base,
                                                            int n_blocks, int span_total,
                                                            int voices_log2, int n_big,
                                                            float *__restrict__ partial,
                                                            void *__restrict__ out)
{
    constexpr bool MIX = (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);
    static_assert(MIX || MODE == OUT_VOICES_F32 || MODE == OUT_VOICES_F64, "unsupported output mode");
    constexpr int PLANES = MODE == OUT_MIX_STEREO ? 2 : 1;
    using T = typename TileTraits<MODE>::T;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    const bool big = (int)blockIdx.x < n_big;   // uniform over the workgroup
    const int group_log2 = big ? voices_log2 : voices_log2 - 1;
    const int V = 1 << group_log2;
    double *xbuf = reinterpret_cast<double *>(smem_raw);           // [2][V][pitch]
    double *ybuf = xbuf + 2 * kPipePitch * V;                      // [2][V][pitch]
    double *gains = ybuf + 2 * kPipePitch * V;                     // [2][V]: left (or mono), right
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // first voice of the workgroup, relative to `first`
    const int base_voice = big ? (int)blockIdx.x << voices_log2
                               : (n_big << voices_log2) + (((int)blockIdx.x - n_big) << group_log2);
    const int rounds = n_blocks * kPipeRounds;

    // ---- set-up per role ------------------------------------------------------------------
    // oscillator lanes: q = 0..127, Lv lanes per voice, Ks samples per lane and round
    const int q = (int)threadIdx.x - 64;
    const int Lv = V >= 4 ? 128 >> group_log2 : 32;
    const int Ks = kPipeRound / Lv;
    const int o_voice = q >= 0 ? q / Lv : 0, o_sub = q >= 0 ? q % Lv : 0;
    const bool o_role = wave == 1 || wave == 2;
    const bool o_active = o_role && o_voice < V;
    const bool o_live = o_active && base_voice + o_voice < count;
    // chain lanes: lane = voice
    const bool c_role = wave == 0;
    const bool c_live = c_role && lane < V && base_voice + lane < count;
    const int my_voice = o_role ? o_voice : lane;
    const int v = first + ((o_live || c_live) ? base_voice + my_voice : 0);

    double freq = 0., amp = 0., phase = 0., duty_edge = 0.;
    int wave_id = W_ONE;
    if (o_role) {
        freq = bk.frequency_rw[v];
        amp = bk.amplitude_rw[v];
        wave_id = bk.wave[v];
        duty_edge = kTau * bk.duty[v];  // oscillators.py:259
        phase = bk.phase[v];
    }
    EnvParams ep{};
    EnvState es{};
    Biquad bq{};
    int coef_row = 0;
    const bool cutoff_lfo = FILT && bk.lfo_par != nullptr;  // uniform over the launch
    double lfo_ph = 0.;
    if (c_role) {
        if (ENV) {
            ep.aCoef = bk.env_coef[0 * (size_t)bk.n + v];
            ep.aBase = bk.env_coef[1 * (size_t)bk.n + v];
            ep.dCoef = bk.env_coef[2 * (size_t)bk.n + v];
            ep.dBase = bk.env_coef[3 * (size_t)bk.n + v];
            ep.rCoef = bk.env_coef[4 * (size_t)bk.n + v];
            ep.rBase = bk.env_coef[5 * (size_t)bk.n + v];
            ep.sustain = bk.env_coef[6 * (size_t)bk.n + v];
            const uint32_t fl = bk.env_flags[v];
            ep.loop = fl & 1u;
            ep.retrigger = fl & 2u;
            const uint32_t st = bk.env_state[v];
            es.stage = (int)(st & 7u);
            es.enabled = (st >> 8) & 1u;
            es.value = bk.env_value[v];
            es.select(ep);
        }
        if (FILT) {
            coef_row = bk.coef_index_rw[v];
            const double *c = bk.coef_table + 5 * (size_t)coef_row;
            bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
            bq.z0 = bk.z0[v];
            bq.z1 = bk.z1[v];
            if (cutoff_lfo)
                lfo_ph = bk.lfo_phase[v];
        }
        if (MIX && lane < V) {
            gains[lane] = c_live ? bk.gain_l[v] : 0.;
            gains[V + lane] = (c_live && MODE == OUT_MIX_STEREO) ? bk.gain_r[v] : 0.;
        }
    }
    // both the oscillator lanes and the chain lane walk their voice's events
    uint32_t ev_i = 0, ev_end = 0, next_ev = 0xFFFFFFFFu;
    if (o_live || c_live) {
        ev_i = bk.ev_offset[v] + bk.ev_cursor[v];
        ev_end = bk.ev_offset[v + 1];
        next_ev = ev_i < ev_end ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
    }
    double omega = 0., start = 0.;
    bool fast_block = false;
    // the wave function shared by every active lane of the oscillator wave, or -1
    int uniform_wave = -1;
    if (o_role) {
        const int first_wave = __builtin_amdgcn_readfirstlane(wave_id);
        uniform_wave = __all(!o_active || wave_id == first_wave) ? first_wave : -1;
    }
    __syncthreads();   // the gains

    for (int t = 0; t < rounds + 2; ++t) {
        if (o_role && t < rounds) {
            // ---- oscillator stage: round t -> xbuf[t & 1] ------------------------------------
            const int b = t / kPipeRounds, rr = t % kPipeRounds;
            if (rr == 0) {
                const uint32_t abs_block = clock_base + (uint32_t)b;
                while (next_ev <= abs_block) {   // voices.py:55-64, the oscillator's share
                    const uint32_t kind = bk.ev_kind[ev_i];
                    if (kind == 1u) {
                        amp = bk.ev_velocity[ev_i];
                        freq = bk.ev_frequency[ev_i];
                    } else if (kind == 3u) {
                        freq = bk.ev_frequency[ev_i];
                    }
                    ++ev_i;
                    next_ev = (ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
                }
                omega = kTau * freq;  // oscillators.py:60  (TAU * frequency) * t
                start = phase;
                // wave-uniform: every phase argument of this block is >= 0 and < 1e9
                fast_block = __all(omega >= 0. && start >= 0. &&
                                   omega * (kDt * (double)kBlock) + start < 1e9);
            }
            if (o_active) {
                double *dst = xbuf + ((size_t)(t & 1) * V + o_voice) * kPipePitch;
                // WK: the wave function when the whole wavefront agrees on it (compile-time
                // dispatch, straight-line code), or -1 (per-lane switch).  FAST: every phase
                // argument of the block is >= 0 and < 1e9 (no sign handling in the remainder).
                auto produce = [&](auto wk, auto fast) {
                    constexpr int WK = decltype(wk)::value;
                    constexpr bool FAST = decltype(fast)::value;
                    const int wsel = WK >= 0 ? WK : wave_id;
                    for (int k = 0; k < Ks; ++k) {
                        const int s = o_sub * Ks + k;            // sample inside the round
                        const int n = rr * kPipeRound + s;       // sample inside the buffer
                        double osc;
                        if (wsel == W_EXTERNAL) {
                            osc = o_live ? bk.ext_input[(size_t)((block_base + b) * kBlock + n) * bk.n + v] : 0.;
                        } else if (wsel == W_ONE) {
                            osc = 1.;
                        } else {
                            // helpers.py:30  t = dt * arange: the product the host's table holds
                            const double arg = omega * (kDt * (double)n) + start;  // oscillators.py:60
                            if constexpr (FAST)
                                osc = wave_eval<true>(wsel, fmod_tau_fast(arg), duty_edge);
                            else
                                osc = wave_eval<false>(wsel, np_mod_tau(arg), duty_edge);
                        }
                        const double xs = ENV ? amp * osc : osc;     // voices.py:73 (amplitude*osc)*env
                        dst[s] = o_live ? xs : 0.;
                    }
                };
                using std::integral_constant;
                if (fast_block) {
                    switch (uniform_wave) {
                    case W_SINE: produce(integral_constant<int, W_SINE>{}, std::true_type{}); break;
                    case W_SQUARE: produce(integral_constant<int, W_SQUARE>{}, std::true_type{}); break;
                    case W_SAWTOOTH: produce(integral_constant<int, W_SAWTOOTH>{}, std::true_type{}); break;
                    case W_TRIANGLE: produce(integral_constant<int, W_TRIANGLE>{}, std::true_type{}); break;
                    case W_PWM: produce(integral_constant<int, W_PWM>{}, std::true_type{}); break;
                    case W_EXTERNAL: produce(integral_constant<int, W_EXTERNAL>{}, std::true_type{}); break;
                    default: produce(integral_constant<int, -1>{}, std::true_type{}); break;
                    }
                } else {
                    produce(integral_constant<int, -1>{}, std::false_type{});
                }
            }
            // oscillators.py:66-67: next start phase is sample 256 of this block
            if (rr == kPipeRounds - 1 && wave_id != W_EXTERNAL && wave_id != W_ONE)
                phase = np_mod_tau(omega * (kDt * (double)kBlock) + start);
        }
        if (c_role && t >= 1 && t <= rounds) {
            // ---- chain stage: round t-1, xbuf -> ybuf ----------------------------------------
            const int r = t - 1;
            const int b = r / kPipeRounds, rr = r % kPipeRounds;
            if (rr == 0) {
                const uint32_t abs_block = clock_base + (uint32_t)b;
                while (next_ev <= abs_block) {   // voices.py:55-64, the chain's share
                    const uint32_t kind = bk.ev_kind[ev_i];
                    if (kind == 1u) {
                        if (ENV)
                            es.gate(ep, true);
                    } else if (kind == 0u) {
                        if (ENV)
                            es.gate(ep, false);
                    } else if (kind == 2u) {
                        if (FILT) {
                            coef_row = bk.ev_index[ev_i];
                            const double *c = bk.coef_table + 5 * (size_t)coef_row;
                            bq.b0 = c[0], bq.b1 = c[1], bq.b2 = c[2], bq.a1 = c[3], bq.a2 = c[4];
                        }
                    }
                    ++ev_i;
                    next_ev = (ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
                }
                if (cutoff_lfo)
                    lfo_cutoff_step(bk, v, lfo_ph, coef_row, bq);
            }
            const int col = lane < V ? lane : 0;
            const double2 *src = reinterpret_cast<const double2 *>(
                xbuf + ((size_t)(r & 1) * V + col) * kPipePitch);
            double2 *dst = reinterpret_cast<double2 *>(ybuf + ((size_t)(r & 1) * V + col) * kPipePitch);
            double x[kPipeRound];
#pragma unroll
            for (int s = 0; s < kPipeRound; s += 2) {
                const double2 p = src[s >> 1];
                x[s] = p.x, x[s + 1] = p.y;
            }
#pragma unroll
            for (int g8 = 0; g8 < kPipeRound; g8 += 8) {
                constexpr int G = 8;
                // speculation as in voice_kernel: the envelope's 8 values move one way, so if
                // any crossed the stage's limit the last did; the biquad runs on a copy
                unsigned long long redo = 0;
                double y[G];
                double vv = 0.;
                if (ENV) {
                    vv = es.value;
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        vv = es.base + vv * es.coef;           // _envelope.c:579-624
                        y[j] = x[g8 + j] * vv;
                    }
                    redo |= __builtin_amdgcn_ballot_w64(es.reached(vv));
                } else {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        y[j] = x[g8 + j];
                }
                Biquad sp = bq;
                if (FILT) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        y[j] = sp.step(y[j]);
                }
                if (__builtin_expect(redo != 0, 0)) {
                    // exact path, from the uncommitted state
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        double z = x[g8 + j];
                        if (ENV)
                            z = z * es.step(ep);
                        if (FILT)
                            z = bq.step(z);
                        y[j] = z;
                    }
                } else {
                    if (ENV)
                        es.value = vv;
                    if (FILT)
                        bq.z0 = sp.z0, bq.z1 = sp.z1;
                }
                if (lane < V) {
#pragma unroll
                    for (int j = 0; j < G; j += 2)
                        dst[(g8 + j) >> 1] = make_double2(y[j], y[j + 1]);
                }
            }
        }
        if (wave == 3 && t >= 2) {
            // ---- output stage: round t-2 ------------------------------------------------------
            const int r = t - 2;
            const int b = r / kPipeRounds, rr = r % kPipeRounds;
            const double *src = ybuf + (size_t)(r & 1) * V * kPipePitch;
            const int s = lane & (kPipeRound - 1);      // sample inside the round
            const int half = lane >> 5;                 // the two halves of the wave share the voices
            const int n = rr * kPipeRound + s;          // sample inside the buffer
            if constexpr (MIX) {
                double acc_l = 0., acc_r = 0.;
                for (int u = half; u < V; u += 2) {
                    const double y = src[u * kPipePitch + s];
                    acc_l += gains[u] * y;                       // mixer.py:43-44
                    if (MODE == OUT_MIX_STEREO)
                        acc_r += gains[V + u] * y;               // mixer.py:99
                }
                acc_l += __shfl_xor(acc_l, 32, 64);
                if (MODE == OUT_MIX_STEREO)
                    acc_r += __shfl_xor(acc_r, 32, 64);
                if (half == 0) {
                    float *row = partial + (size_t)blockIdx.x * n_blocks * PLANES * kBlock +
                                 (size_t)b * PLANES * kBlock + n;
                    row[0] = (float)acc_l;
                    if (MODE == OUT_MIX_STEREO)
                        row[kBlock] = (float)acc_r;
                }
            } else {
                T *o = reinterpret_cast<T *>(out);
                for (int u = half; u < V; u += 2) {
                    if (base_voice + u < count)
                        o[(size_t)(base_voice + u) * span_total + (size_t)b * kBlock + n] = (T)src[u * kPipePitch + s];
                }
            }
        }
        lds_barrier();
    }

    // ---- store state ----------------------------------------------------------------------
    if (o_live && o_sub == 0) {
        bk.phase[v] = phase;
        bk.frequency_rw[v] = freq;
        bk.amplitude_rw[v] = amp;
    }
    if (c_live) {
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
        bk.ev_cursor[v] = ev_i - bk.ev_offset[v];
    }
}

}  // namespace klmi
