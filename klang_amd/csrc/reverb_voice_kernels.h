// reverb_voice_kernels.h -- Voice -> Reverb -> Mixer in ONE kernel (gfx950): the channel's dry
// signal is made inside the wavefront that runs its combs, so nothing but delay-line slots
// (and one partial row per workgroup) moves through HBM.
//
// reverb_kernels.h reads the dry signal from a float64 bus that the voice kernel writes first:
// 8 B written + 8 B read per channel-sample on top of the 80 B of comb traffic (10 combs x
// read + write x 4 B), and one 4-byte partial row per WAVE on top of that.  Here a wavefront
// plays two roles per buffer (= one pass of 256 samples, reverb_kernels.h):
//
//   role A, lane = one of the wave's VPW voices: the voice's events of this buffer
//     (voices.py:55-64), then the envelope's 256 samples -- the one part of a voice that is
//     serial in time (_envelope.c:579-624: its stage switches depend on the rounded values) --
//     eight at a time with the fast kernel's monotone speculation (bank_kernels.h), into a
//     wave-private LDS row; the oscillator's closed-form parameters of the buffer (TAU*f,
//     start phase, amplitude, wave function) next to it; the phase steps to sample 256
//     (oscillators.py:66-67).
//   role B, lane = 4 consecutive samples (as in reverb_kernel): for each voice the oscillator
//     at its 4 time points (closed form: oscillators.py:60, waves.py:22-39; t[n] = DT * n is
//     formed on the spot, the same IEEE product helpers.py:30 forms), (amplitude * osc) * env
//     (voices.py:73), the ten combs, dry/wet, gains.
//
// The two roles meet in LDS only; a wave's LDS operations execute in order, so no barrier is
// needed between them.  The four waves of a workgroup then fold their four partial sums
// through LDS (one workgroup barrier per buffer, double-buffered) and store ONE partial row
// per workgroup: 1 B per channel-sample instead of 4.
// The arithmetic per sample is the two-kernel path's, operation for operation: float64
// streams, ring contents and voice state are bit-identical to it.
#pragma once
#include "bank_kernels.h"
#include "reverb_kernels.h"

namespace klmi {

constexpr int kFusedVoicesPerWaveMax = 4;

template <int VPW> struct FusedWaveLds {
    double env[VPW][kBlock];      // role A -> role B: the envelope of this buffer
    double omega[VPW], start[VPW], amp[VPW], duty_edge[VPW];
    int wave_id[VPW], fast[VPW];
    // role A's voice state between buffers (from the bank's arrays at the start of the launch,
    // back to them at its end): registers would be held -- or spilled -- across role B, and
    // re-reading the arrays every buffer would put the serial envelope behind a wait for the
    // wave's outstanding ring stores (vector memory counts in order on gfx9)
    double freq[VPW], phase[VPW], env_value[VPW], env_coef[7][VPW];
    uint32_t env_state[VPW], env_flags[VPW], ev_i[VPW], ev_end[VPW], next_ev[VPW];
};

template <int VPW, int PLANES> struct FusedGroupLds {
    FusedWaveLds<VPW> wave[kReverbWaves];
    float sum[2][kReverbWaves][PLANES][kBlock];   // per-wave partial sums, double-buffered
};

// grid = ceil(n / (4 * VPW)), block = 256.  `n_blocks` buffers from absolute block
// `clock_base` (events) / block `block_base` of this call (external input).
// partial[group * row_stride + (block * PLANES + plane) * 256 + i]: the sum over the
// workgroup's voices (waves in ascending order, float32 per wave, summed in float64).
template <typename RingT, bool STEREO, bool ENV, int VPW>
__global__ __launch_bounds__(kReverbThreads, 2) void reverb_voice_kernel(
    ReverbState r, const BankView *bkp, int block_base, uint32_t clock_base, int n_blocks,
    float *__restrict__ partial, size_t row_stride)
{
    // The bank's view (some forty pointers) stays in memory and is read where it is used: as a
    // by-value kernel argument its pointers were all kept in scalar registers next to the
    // per-comb constants, a hundred of them spilled, and role B paid for it in every buffer.
    constexpr int NE = 10;
    constexpr int PLANES = STEREO ? 2 : 1;
    __shared__ __align__(16) FusedGroupLds<VPW, PLANES> lds;
    const int lane = threadIdx.x & 63;
    const int wig = threadIdx.x >> 6;   // wave in group
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kReverbWaves + wig);
    const int v0 = min(wave * VPW, r.n);
    const int v1 = min(v0 + VPW, r.n);
    const int nv = v1 - v0;
    FusedWaveLds<VPW> &w = lds.wave[wig];
    const double echo_count = (double)NE;
    const double dry = 1. - r.dry_wet, wet_gain = r.dry_wet;
    RingT *__restrict__ rings = static_cast<RingT *>(r.rings);
    const int base = lane * 4;

    // role A: lane j < nv is voice v0 + j; its state lives in the wave's LDS record
    const bool mine = lane < nv;
    const int v = mine ? v0 + lane : 0;
    if (mine) {
        const BankView &bk = *bkp;
        w.freq[lane] = bk.frequency_rw[v];
        w.amp[lane] = bk.amplitude_rw[v];
        w.phase[lane] = bk.phase[v];
        w.wave_id[lane] = bk.wave[v];
        w.duty_edge[lane] = kTau * bk.duty[v];  // oscillators.py:259
        if (ENV) {
#pragma unroll
            for (int c = 0; c < 7; ++c)
                w.env_coef[c][lane] = bk.env_coef[c * (size_t)bk.n + v];
            w.env_flags[lane] = bk.env_flags[v];
            w.env_state[lane] = bk.env_state[v];
            w.env_value[lane] = bk.env_value[v];
        }
        const uint32_t ev_first = bk.ev_offset[v];
        const uint32_t ev_i = ev_first + bk.ev_cursor[v], ev_end = bk.ev_offset[v + 1];
        w.ev_i[lane] = ev_i;
        w.ev_end[lane] = ev_end;
        w.next_ev[lane] = ev_i < ev_end ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
    }

    int pos[NE];   // wave-uniform: ring positions at the start of the current buffer
#pragma unroll
    for (int k = 0; k < NE; ++k)
        pos[k] = r.position[k];

    for (int b = 0; b < n_blocks; ++b) {
        // ---- ring positions of this buffer; the first voice's ring reads go out NOW, so that
        // role A's serial arithmetic runs under their latency
        int slot[NE], again[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            int s = pos[k] + base;
            s = s >= r.length[k] ? s - r.length[k] : s;
            slot[k] = s + r.offset[k];
            again[k] = s + 3 >= r.length[k] ? -r.length[k] : s < kReverbMirror ? r.length[k] : 0;
        }
        Quad<RingT> old[NE];
        if (nv > 0) {
            const RingT *ring = rings + (size_t)v0 * (size_t)r.stride;
#pragma unroll
            for (int k = 0; k < NE; ++k)
                old[k] = load_quad(ring + slot[k]);
        }
        asm volatile("" ::: "memory");
        // ================= role A =================
        {
            const int l = mine ? lane : 0;
            double freq = w.freq[l];
            double amp = w.amp[l];
            const int wave_id = w.wave_id[l];
            double phase = w.phase[l];
            EnvParams ep;
            EnvState es;
            if (ENV) {
                ep.aCoef = w.env_coef[0][l], ep.aBase = w.env_coef[1][l];
                ep.dCoef = w.env_coef[2][l], ep.dBase = w.env_coef[3][l];
                ep.rCoef = w.env_coef[4][l], ep.rBase = w.env_coef[5][l];
                ep.sustain = w.env_coef[6][l];
                const uint32_t fl = w.env_flags[l];
                ep.loop = fl & 1u;
                ep.retrigger = fl & 2u;
                const uint32_t st = w.env_state[l];
                es.stage = (int)(st & 7u);
                es.enabled = (st >> 8) & 1u;
                es.value = w.env_value[l];
                es.select(ep);
            }
            uint32_t next_ev = mine ? w.next_ev[l] : 0xFFFFFFFFu;
            const uint32_t abs_block = clock_base + (uint32_t)b;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(next_ev <= abs_block) != 0, 0)) {
                const BankView &bk = *bkp;
                uint32_t ev_i = w.ev_i[l];
                const uint32_t ev_end = w.ev_end[l];
                while (next_ev <= abs_block) {  // voices.py:55-64 (bank_kernels.h event loop)
                    const uint32_t kind = bk.ev_kind[ev_i];
                    if (kind == 1u) {
                        if (ENV)
                            es.gate(ep, true);
                        amp = bk.ev_velocity[ev_i];
                        freq = bk.ev_frequency[ev_i];
                    } else if (kind == 0u) {
                        if (ENV)
                            es.gate(ep, false);
                    } else if (kind != 2u) {    // (2: cutoff change -- no filter in this chain)
                        freq = bk.ev_frequency[ev_i];
                    }
                    ++ev_i;
                    next_ev = (ev_i < ev_end) ? bk.ev_block[ev_i] : 0xFFFFFFFFu;
                }
                if (mine) {
                    w.ev_i[l] = ev_i, w.next_ev[l] = next_ev;
                    w.freq[l] = freq, w.amp[l] = amp;
                }
            }
            const double omega = kTau * freq;  // oscillators.py:60  (TAU * frequency) * t
            const double start = phase;
            const double t_end = kDt * (double)kBlock;
            if (mine) {
                w.omega[l] = omega, w.start[l] = start;
                w.fast[l] = (omega >= 0. && start >= 0. && omega * t_end + start < 1e9) ? 1 : 0;
            }
            if (ENV) {
                for (int i8 = 0; i8 < kBlock; i8 += 8) {
                    // the fast kernel's speculation: the recurrence is monotone, so if any of
                    // the eight values crossed the stage's limit the last one did; a hit (any
                    // lane) redoes the eight with the exact state machine from the uncommitted
                    // state
                    double e8[8];
                    double vv = es.value;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        vv = es.base + vv * es.coef;       // _envelope.c:579-624
                        e8[j] = vv;
                    }
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(es.reached(vv)) != 0, 0)) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            e8[j] = es.step(ep);
                    } else {
                        es.value = vv;
                    }
                    if (mine) {
                        double2 *dst = reinterpret_cast<double2 *>(&w.env[l][i8]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            dst[j] = make_double2(e8[2 * j], e8[2 * j + 1]);
                    }
                }
            }
            // oscillators.py:66-67: next start phase is sample 256 of this block
            if (wave_id != W_EXTERNAL && wave_id != W_ONE)
                phase = np_mod_tau(omega * t_end + start);
            if (mine) {
                w.phase[l] = phase;
                if (ENV) {
                    w.env_value[l] = es.value;
                    w.env_state[l] = (uint32_t)es.stage | (es.enabled ? 0x100u : 0u);
                }
            }
        }
        // LDS operations of one wave execute in order; keep the compiler from reordering
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ================= role B =================
        double acc_l[4] = {0., 0., 0., 0.}, acc_r[4] = {0., 0., 0., 0.};
#pragma unroll 1
        for (int j = 0; j < nv; ++j) {
            const int vb = v0 + j;
            RingT *ring = rings + (size_t)vb * (size_t)r.stride;
            const double gl = bkp->gain_l[vb];
            const double gr = STEREO ? bkp->gain_r[vb] : 0.;
            const int wid = __builtin_amdgcn_readfirstlane(w.wave_id[j]);
            double x[4];
            if (wid == W_EXTERNAL) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    x[i] = bkp->ext_input[(size_t)((block_base + b) * kBlock + base + i) * r.n + vb];
            } else if (wid == W_ONE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    x[i] = 1.;
            } else {
                const double om = w.omega[j], st = w.start[j], de = w.duty_edge[j];
                const bool fast = __builtin_amdgcn_readfirstlane(w.fast[j]) != 0;
                double ph[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double arg = om * (kDt * (double)(base + i)) + st;   // oscillators.py:60
                    ph[i] = fast ? fmod_tau_fast(arg) : np_mod_tau(arg);
                }
                switch (wid) {      // wave-uniform
                case W_SINE:
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        x[i] = wave_eval<false>(W_SINE, ph[i], de);
                    break;
                case W_SQUARE:
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        x[i] = wave_eval<false>(W_SQUARE, ph[i], de);
                    break;
                case W_SAWTOOTH:
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        x[i] = wave_eval<false>(W_SAWTOOTH, ph[i], de);
                    break;
                case W_TRIANGLE:
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        x[i] = wave_eval<false>(W_TRIANGLE, ph[i], de);
                    break;
                default:
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        x[i] = wave_eval<false>(W_PWM, ph[i], de);
                    break;
                }
            }
            if (ENV) {
                const double a = w.amp[j];
                const double2 e01 = *reinterpret_cast<const double2 *>(&w.env[j][base]);
                const double2 e23 = *reinterpret_cast<const double2 *>(&w.env[j][base + 2]);
                x[0] = a * x[0] * e01.x, x[1] = a * x[1] * e01.y;    // voices.py:73
                x[2] = a * x[2] * e23.x, x[3] = a * x[3] * e23.y;
            }
            // all ring loads of the voice first, then compute and store (the first voice's
            // went out before role A)
            if (j > 0) {
#pragma unroll
                for (int k = 0; k < NE; ++k)
                    old[k] = load_quad(ring + slot[k]);
            }
            double sum[4];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                Quad<RingT> y;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double yi = x[i] + r.alpha[k] * (double)old[k].e[i];  // _filters.c:275
                    y.e[i] = (RingT)yi;                                          // _filters.c:276
                    sum[i] = (k == 0) ? yi : sum[i] + yi;                        // effects.py:610-612
                }
                store_quad(ring + slot[k], y);
                if (again[k] != 0)
                    store_quad(ring + slot[k] + again[k], y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double wet = sum[i] / echo_count;                     // effects.py:614
                const double out = dry * x[i] + wet_gain * wet;             // math.py:75
                acc_l[i] += gl * out;                                       // mixer.py:43-44
                if (STEREO)
                    acc_r[i] += gr * out;                                   // mixer.py:99
            }
        }
        {
        // ---- the workgroup's four partial sums -> one row (fixed order) -------------------
            {
                float(&mysum)[PLANES][kBlock] = lds.sum[b & 1][wig];
                Quad<float> ql, qr;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    ql.e[i] = (float)acc_l[i], qr.e[i] = (float)acc_r[i];
                store_quad(&mysum[0][base], ql);
                if (STEREO)
                    store_quad(&mysum[PLANES - 1][base], qr);
            }
            __syncthreads();    // one per buffer: buffer b+1 writes the other half of lds.sum
            if (wig == (b & 3)) {   // the folding wave rotates
                float *o = partial + (size_t)blockIdx.x * row_stride + (size_t)b * (PLANES * kBlock) + base;
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    double s[4] = {0., 0., 0., 0.};
#pragma unroll
                    for (int q = 0; q < kReverbWaves; ++q) {
                        const Quad<float> p = load_quad(&lds.sum[b & 1][q][pl][base]);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            s[i] += (double)p.e[i];
                    }
                    Quad<float> o4;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        o4.e[i] = (float)s[i];
                    // written once, read once by bus_slice_kernel much later: a streaming store,
                    // so that the row does not take ring lines' places in the caches
                    typedef float Float4 __attribute__((ext_vector_type(4)));
                    Float4 v4 = {o4.e[0], o4.e[1], o4.e[2], o4.e[3]};
                    __builtin_nontemporal_store(v4, reinterpret_cast<Float4 *>(o + pl * kBlock));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            pos[k] += kReverbPass;
            pos[k] = pos[k] >= r.length[k] ? pos[k] - r.length[k] : pos[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- role A's state back to the bank ------------------------------------------------
    if (mine) {
        const BankView &bk = *bkp;
        bk.phase[v] = w.phase[lane];
        bk.frequency_rw[v] = w.freq[lane];
        bk.amplitude_rw[v] = w.amp[lane];
        if (ENV) {
            bk.env_value[v] = w.env_value[lane];
            bk.env_state[v] = w.env_state[lane];
        }
        bk.ev_cursor[v] = w.ev_i[lane] - bk.ev_offset[v];
    }
}

}  // namespace klmi
