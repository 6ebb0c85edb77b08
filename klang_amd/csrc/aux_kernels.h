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

}  // namespace klmi
