// wide_kernels.h -- the voice kernel for banks that do not fill the chip: the time axis of
// a voice is split over a group of lanes (gfx950).
//
// voice_kernel (bank_kernels.h) maps one voice to one lane and walks time serially: 65,536
// voices are one wavefront per SIMD, but a 4,096-voice bank leaves 15 of 16 SIMDs idle and
// the native 20-voice haunting_envelopes patch is a single wavefront.  Only two parts of a
// voice are serial in time: the envelope recurrence (_envelope.c:579-624: its stage
// switches depend on the rounded values) and the biquad (lfilter's z-state).  The
// oscillator is not: inside a buffer its phase is the closed form (TAU*f)*t[n] + start
// (oscillators.py:57-67), and notes land on buffer boundaries (voices.py:55-64).
//
// Mapping: a voice owns L = 2..32 adjacent lanes (a power of two), a wavefront 64/L voices.
// Per round of R = max(L, 8) samples:
//   1. every lane evaluates the oscillator for ITS R/L samples (time-parallel) and leaves
//      amplitude*osc in a wave-private LDS strip;
//   2. every lane of the group reads the strip back (broadcast reads) and runs the SAME
//      serial chain over the R samples -- the state of a voice is replicated in its L
//      lanes, so nothing has to be handed from lane to lane, and the replicas cost nothing
//      (the lanes would idle otherwise).  Eight samples at a time as one straight-line
//      region with voice_kernel's monotone-envelope speculation;
//   3. lane 0 of the group leaves the chain's outputs in the strip, every lane picks up its
//      own samples, applies the gains and emits them (mix tile / per-voice rows).
// The arithmetic per sample is voice_kernel's, operation for operation: results are
// bit-identical whatever L is.  Events are read by all lanes of a group alike.
#pragma once
#include "bank_kernels.h"

namespace klmi {

constexpr int kWideStripDoubles = 256;   // per wave: 64/L voices x R samples <= 32 x 8
constexpr int kWideMaxLanesLog2 = 5;     // L <= 32: a round never exceeds a stereo chunk

template <int MODE> __host__ __device__ constexpr size_t wide_lds_bytes()
{
    constexpr bool MIX = (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);
    return sizeof(double) * kWavesPerGroup * kWideStripDoubles + (MIX ? tile_bytes<MODE>() : 0);
}

// grid.x = ceil(count / (4 * (64 >> lanes_log2))) workgroups of 4 independent waves.
// partial: [gridDim.x * 4 waves][n_blocks*planes*256] float32 (mix modes)
// out:     [count][span_total] per-voice rows (OUT_VOICES_F32 / OUT_VOICES_F64)
template <bool ENV, bool FILT, int MODE>
__global__ __launch_bounds__(kGroup) void voice_wide_kernel(BankView bk, int first, int count,
                                                            int block_base, uint32_t clock_base,
                                                            int n_blocks, int span_total,
                                                            int lanes_log2,
                                                            float *__restrict__ partial,
                                                            void *__restrict__ out)
{
    using TT = TileTraits<MODE>;
    using T = typename TT::T;
    constexpr bool MIX = (MODE == OUT_MIX_MONO || MODE == OUT_MIX_STEREO);
    static_assert(MIX || MODE == OUT_VOICES_F32 || MODE == OUT_VOICES_F64, "unsupported output mode");
    constexpr int CH = MIX ? TT::chunk : 64;   // samples between two flushes of the mix tile
    constexpr int NCHUNK = kBlock / CH;
    constexpr int PLANES = MIX ? TT::planes : 1;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *strip = reinterpret_cast<double *>(smem_raw) + (size_t)wave * kWideStripDoubles;
    T *tile = reinterpret_cast<T *>(smem_raw + sizeof(double) * kWavesPerGroup * kWideStripDoubles) +
              (size_t)wave * PLANES * CH * TT::stride;

    const int L = 1 << lanes_log2;                  // lanes per voice
    const int R = L < 8 ? 8 : L;                    // samples per round
    const int K = R >> lanes_log2;                  // samples per lane and round: 4, 2 or 1
    const int vpw = 64 >> lanes_log2;               // voices per wave
    const int vi = lane >> lanes_log2;              // the lane's voice within the wave
    const int sub = lane & (L - 1);                 // the lane's place in its group
    const int local = (blockIdx.x * kWavesPerGroup + wave) * vpw + vi;
    const bool live = local < count;
    const int v = first + (live ? local : 0);
    double *mine = strip + vi * R + sub * K;        // this lane's K places in the strip
    const double *group = strip + vi * R;           // the group's R places

    // ---- load parameters and state (replicated in the L lanes of the voice) ------------
    double freq = bk.frequency_rw[v];
    double amp = bk.amplitude_rw[v];
    const int wave_id = bk.wave[v];
    const double duty_edge = kTau * bk.duty[v];  // oscillators.py:259
    double phase = bk.phase[v];

    EnvParams ep;
    EnvState es;
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

    Biquad bq;
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

    double gl = 0., gr = 0.;
    if (MIX) {
        gl = live ? bk.gain_l[v] : 0.;
        if (MODE == OUT_MIX_STEREO)
            gr = live ? bk.gain_r[v] : 0.;
        // columns of voices this wave does not have stay zero for the whole launch
        for (int i = lane; i < PLANES * CH * TT::stride; i += 64)
            tile[i] = T(0);
    }

    uint32_t ev_i = bk.ev_offset[v] + bk.ev_cursor[v];
    const uint32_t ev_end = bk.ev_offset[v + 1];
    uint32_t next_ev = (live && ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;

    for (int b = 0; b < n_blocks; ++b) {
        const uint32_t abs_block = clock_base + (uint32_t)b;
        // voices.py:55-64: notes are consumed at the block boundary
        while (next_ev <= abs_block) {
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
        // wave-uniform: every phase argument of this block is >= 0 and < 1e9
        const bool fast_block =
            __all(omega >= 0. && start >= 0. && omega * (kDt * (double)kBlock) + start < 1e9);
        const bool has_phase = wave_id != W_EXTERNAL && wave_id != W_ONE;

        for (int c = 0; c < NCHUNK; ++c) {
            const int s0 = b * kBlock + c * CH;   // sample offset of the chunk inside this launch
            for (int i0 = 0; i0 < CH; i0 += R) {
                const int r0 = c * CH + i0;       // first sample of the round inside the buffer
                // ---- 1. the lane's own samples: oscillator, times amplitude ----------------
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < K) {
                        const int n = r0 + sub * K + k;      // sample index inside the buffer
                        double osc;
                        if (wave_id == W_EXTERNAL) {
                            osc = bk.ext_input[(size_t)(block_base * kBlock + b * kBlock + n) * bk.n + v];
                        } else if (wave_id == W_ONE) {
                            osc = 1.;
                        } else {
                            // helpers.py:30  t = dt * arange: the same product the host's table holds
                            const double arg = omega * (kDt * (double)n) + start;  // oscillators.py:60
                            osc = fast_block ? wave_eval<true>(wave_id, fmod_tau_fast(arg), duty_edge)
                                             : wave_eval<false>(wave_id, np_mod_tau(arg), duty_edge);
                        }
                        mine[k] = ENV ? amp * osc : osc;     // voices.py:73 (amplitude*osc)*env
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- 2. the serial chain over the round, 8 samples at a time ----------------
                for (int g8 = 0; g8 < R; g8 += 8) {
                    constexpr int G = 8;
                    double y[G];
                    {
                        const double2 *src = reinterpret_cast<const double2 *>(group + g8);
#pragma unroll
                        for (int j = 0; j < G; j += 2) {
                            const double2 p = src[j >> 1];
                            y[j] = p.x, y[j + 1] = p.y;
                        }
                    }
                    double x8[G];
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        x8[j] = y[j];
                    // speculation as in voice_kernel: the envelope's 8 values move one way, so
                    // if any crossed the stage's limit the last did; the biquad runs on a copy
                    unsigned long long redo = 0;
                    double vv = 0.;
                    if (ENV) {
                        vv = es.value;
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            vv = es.base + vv * es.coef;           // _envelope.c:579-624
                            y[j] = x8[j] * vv;
                        }
                        redo |= __builtin_amdgcn_ballot_w64(es.reached(vv));
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
                            double x = x8[j];
                            if (ENV)
                                x = x * es.step(ep);
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
                    // ---- 3a. one lane of the group leaves the outputs in the strip ---------
                    if (sub == 0) {
                        double2 *dst = reinterpret_cast<double2 *>(strip + vi * R + g8);
#pragma unroll
                        for (int j = 0; j < G; j += 2)
                            dst[j >> 1] = make_double2(y[j], y[j + 1]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- 3b. every lane emits its own samples -----------------------------------
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < K) {
                        const double x = mine[k];
                        const int i = i0 + sub * K + k;   // sample inside the chunk
                        if constexpr (MODE == OUT_MIX_MONO) {
                            tile[i * TT::stride + vi] = (float)(gl * x);
                        } else if constexpr (MODE == OUT_MIX_STEREO) {
                            tile[i * TT::stride + vi] = (float)(gl * x);
                            tile[(TT::chunk + i) * TT::stride + vi] = (float)(gr * x);
                        } else {
                            if (live)
                                reinterpret_cast<T *>(out)[(size_t)local * span_total + s0 + i] = (T)x;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (MIX)
                flush_chunk<MODE>(tile, lane, wave, b, c, n_blocks, count, span_total, s0, partial, out);
        }
        // oscillators.py:66-67: next start phase is sample 256 of this block
        if (has_phase)
            phase = np_mod_tau(omega * (kDt * (double)kBlock) + start);
    }

    // ---- store state (one lane per voice) -------------------------------------------------
    if (live && sub == 0) {
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
        bk.ev_cursor[v] = ev_i - bk.ev_offset[v];
    }
}

}  // namespace klmi
