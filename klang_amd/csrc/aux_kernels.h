// aux_kernels.h -- batched forms of the reference's two CPython extension
// types (klang/audio/_envelope.c, klang/audio/_filters.c) for gfx950.
#pragma once
#include "dsp_device.h"

namespace klmi {

// _envelope.c:636-663 sample(n) for `count` independent envelopes; lane = envelope.
// coef: [7][count]; out: [n][count] (time-major, coalesced); state updated in place.
__global__ void env_sample_kernel(int count, long n, const double *__restrict__ coef,
                                  const uint32_t *__restrict__ flags, double *__restrict__ value,
                                  uint32_t *__restrict__ state, double *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count)
        return;
    EnvParams ep;
    ep.aCoef = coef[0 * (size_t)count + e];
    ep.aBase = coef[1 * (size_t)count + e];
    ep.dCoef = coef[2 * (size_t)count + e];
    ep.dBase = coef[3 * (size_t)count + e];
    ep.rCoef = coef[4 * (size_t)count + e];
    ep.rBase = coef[5 * (size_t)count + e];
    ep.sustain = coef[6 * (size_t)count + e];
    ep.loop = flags[e] & 1u;
    ep.retrigger = flags[e] & 2u;
    EnvState es;
    es.stage = (int)(state[e] & 7u);
    es.enabled = (state[e] >> 8) & 1u;
    es.value = value[e];
    es.select(ep);
    for (long i = 0; i < n; ++i)
        out[(size_t)i * count + e] = es.step(ep);
    value[e] = es.value;
    state[e] = (uint32_t)es.stage | (es.enabled ? 0x100u : 0u);
}

// _envelope.c:549-567 gate() on envelopes [first, first+count).
__global__ void env_gate_kernel(int total, int first, int count, int on,
                                const uint32_t *__restrict__ flags, uint32_t *__restrict__ state)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const int e = first + i;
    (void)total;
    int stage = (int)(state[e] & 7u);
    bool enabled;
    const bool retrigger = flags[e] & 2u;
    if (on) {
        enabled = true;
        if (retrigger || stage == S_OFF || stage == S_RELEASE)
            stage = S_ATTACK;
    } else {
        enabled = false;
        if (stage == S_ATTACK || stage == S_DECAY || stage == S_SUSTAIN)
            stage = S_RELEASE;
    }
    state[e] = (uint32_t)stage | (enabled ? 0x100u : 0u);
}

// _filters.c:231-306 for `count` filters of equal ring length; lane = filter.
// ring: [length][count], x / y: [n][count] (time-major).  Serial in time per
// lane: correct for any length >= 1 (a lane's loads and stores to one address
// stay in program order).
template <int KIND>
__global__ void ring_filter_kernel(int count, int length, int position, long n,
                                   const double *__restrict__ alpha, double *ring,
                                   const double *__restrict__ x, double *__restrict__ y)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= count)
        return;
    const double a = alpha[f];
    int pos = position;
    for (long i = 0; i < n; ++i) {
        const double old = ring[(size_t)pos * count + f];
        const double xi = x[(size_t)i * count + f];
        double yi, store;
        if (KIND == 0) {         // forward comb
            yi = xi + a * old;
            store = xi;
        } else if (KIND == 1) {  // backward comb
            yi = xi + a * old;
            store = yi;
        } else {                 // echo
            yi = old;
            store = a * yi + xi;
        }
        y[(size_t)i * count + f] = yi;
        ring[(size_t)pos * count + f] = store;
        pos = pos + 1 == length ? 0 : pos + 1;
    }
}

// klang.audio.synthesizer.Kick x count (synthesizer.py:22-55, 253-283, waves.py:56-61): one
// workgroup of 256 threads = one kick, thread = sample position i of the buffer.  The kick's
// currentPhase is a VECTOR of 256 start phases after its first buffer (sample_wave returns
// wrap(phase[-1] + TAU * frequency * DT) with the frequency array), so every thread carries
// its own phase and needs the last thread's phase of the buffer: one LDS broadcast per buffer.
// par: [4][count] frequency, decay, intensity, pitchDecay; state: time [count], phase
// [count][256]; trig: [count][n_blocks] 1 where a note-on with pitch > 0 arrives;
// out: [count][n_blocks*256] float64.
__global__ __launch_bounds__(kBlock) void kick_kernel(int count, int n_blocks,
                                                      const double *__restrict__ par,
                                                      double *__restrict__ time_state,
                                                      double *__restrict__ phase_state,
                                                      const unsigned char *__restrict__ trig,
                                                      const double *__restrict__ t_table,
                                                      double *__restrict__ out)
{
    __shared__ double last_phase;
    const int k = blockIdx.x, i = threadIdx.x;
    const double frequency = par[k], decay = par[count + k];
    const double intensity = par[2 * (size_t)count + k], pitch_decay = par[3 * (size_t)count + k];
    const double cd = -kPi / decay, cp = -kPi / pitch_decay;   // -PI / decay * t
    const double t = t_table[i];                                // helpers.py:30 T
    double time = time_state[k];
    double phase0 = phase_state[(size_t)k * kBlock + i];
    const double shape_d = exp(cd * t), shape_p = exp(cp * t);  // np.exp(-PI / decay * T)
    for (int b = 0; b < n_blocks; ++b) {
        if (trig[(size_t)k * n_blocks + b]) {                   // synthesizer.py:275-278
            time = 0.;
            phase0 = 0.;
        }
        const double env = exp(cd * time) * shape_d;            // synthesizer.py:34-35
        const double pitch = frequency * (1. + intensity * (exp(cp * time) * shape_p));
        const double w = kTau * pitch;
        const double phase = w * t + phase0;                    // waves.py:60
        out[((size_t)k * n_blocks + b) * kBlock + i] = env * sin(phase);
        if (i == kBlock - 1)
            last_phase = phase;
        __syncthreads();
        const double x = last_phase + w * kDt + kPi;            // waves.py:61, math.py:9-11 wrap
        phase0 = np_mod_tau(x) - kPi;
        time = time + kDt * kBlock;                             // synthesizer.py:36
        __syncthreads();
    }
    phase_state[(size_t)k * kBlock + i] = phase0;
    if (i == 0)
        time_state[k] = time;
}

// klang.audio.oscillators.Chirper x count (oscillators.py:280-308) over chirp_phase
// (oscillators.py:18-38): scipy.signal's _chirp_phase (vertex_zero=True) until tEnd, the end
// frequency held after it.  One workgroup of 256 threads = one chirper, thread =
// sample position; every thread owns one element of the chirper's time vector, which gains
// DT * BUFFER_SIZE per buffer (a float64 running sum per element, as `self.t += ...`).
// par: [3][count] startFrequency, duration, endFrequency; kind: [2][count] method, wave;
// t_state: [count][256]; out: [count][n_blocks*256] float64.
enum ChirpMethod : int { CHIRP_LINEAR = 0, CHIRP_QUADRATIC, CHIRP_LOGARITHMIC, CHIRP_HYPERBOLIC };

__global__ __launch_bounds__(kBlock) void chirp_kernel(int count, int n_blocks,
                                                       const double *__restrict__ par,
                                                       const int32_t *__restrict__ kind,
                                                       double *__restrict__ t_state,
                                                       double *__restrict__ out)
{
    const int k = blockIdx.x, i = threadIdx.x;
    const double f0 = par[k], t1 = par[count + k], f1 = par[2 * (size_t)count + k];
    const int method = kind[k], wave = kind[count + k];
    const double w1 = kTau * f1;
    __shared__ double t_sh[kBlock];
    __shared__ double offset_sh;
    double t = t_state[(size_t)k * kBlock + i];
    for (int b = 0; b < n_blocks; ++b) {
        t_sh[i] = t;
        __syncthreads();
        const double t_first = t_sh[0], t_last = t_sh[kBlock - 1];
        double p;
        if (t1 <= t_first || f0 == f1) {
            p = w1 * t;   // oscillators.py:23-25: held frequency, no offset
        } else {
            if (method == CHIRP_QUADRATIC) {
                const double beta = (f1 - f0) / (t1 * t1);
                p = kTau * (f0 * t + beta * pow(t, 3.) / 3.);
            } else if (method == CHIRP_LOGARITHMIC) {
                const double beta = t1 / log(f1 / f0);
                p = kTau * beta * f0 * (pow(f1 / f0, t / t1) - 1.0);
            } else if (method == CHIRP_HYPERBOLIC) {
                const double sing = -f1 * t1 / (f0 - f1);
                p = kTau * (-sing * f0) * log(fabs(1 - t / sing));
            } else {
                const double beta = (f1 - f0) / t1;
                p = kTau * (f0 * t + 0.5 * beta * t * t);
            }
            if (!(t_last < t1)) {
                // oscillators.py:33-37: this buffer straddles tEnd (uniform over the
                // workgroup); the first sample at or after it hands its phase on
                const bool held = t >= t1;
                if (held && !(t_sh[i - 1] >= t1))   // i > 0 here: t_first < t1
                    offset_sh = p - w1 * t;
                __syncthreads();
                if (held)
                    p = w1 * t + offset_sh;
            }
        }
        out[((size_t)k * n_blocks + b) * kBlock + i] = wave_eval_any(wave, p, kPi);
        t = t + kDt * kBlock;   // oscillators.py:308
        __syncthreads();        // t_sh and offset_sh are rewritten by the next buffer
    }
    t_state[(size_t)k * kBlock + i] = t;
}

}  // namespace klmi
