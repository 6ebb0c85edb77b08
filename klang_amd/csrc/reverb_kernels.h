// reverb_kernels.h -- klang's Reverb (effects.py:564-615): n prime-length
// feedback comb filters in parallel (BackwardCombFilter, _filters.c:257-280),
// mean, dry/wet blend -- for a bank of independent channels on gfx950.
//
// Data layout: ring slot-major, voice fastest: rings[(offset_k + slot) * n + v],
// so a wave's 64 voices read/write one 512-byte row per (comb, sample).
// Parallelism: every comb delay is >= min_length samples, so up to min_length
// consecutive samples depend only on ring contents written by EARLIER launches.
// One launch covers such a slab with voices x samples threads; the host walks
// the slabs in order (stream order is the only synchronisation needed).
// This kernel is HBM-bound: per voice-sample it moves n_echos * (8 + 8) bytes
// of ring state plus 8 bytes of dry input.
#pragma once
#include "dsp_device.h"

namespace klmi {

constexpr int kMaxEchos = 16;
constexpr int kReverbSamplesPerWave = 4;   // default; 2 and 8 are instantiated for tuning

struct ReverbState {
    int n;         // voices
    int n_echos;
    int min_length;
    double dry_wet;
    int length[kMaxEchos];
    int position[kMaxEchos];      // ring write/read position at slab sample 0
    long long offset[kMaxEchos];  // first slot of comb k in `rings` (in slots)
    double alpha[kMaxEchos];
    double *rings;
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// grid = (ceil(n/64), ceil(ns / kReverbSamplesPerWave)), block = 64.
// x: [ns][n] dry input; partial[group * partial_stride + s] = sum over the
// wave's voices of gain * Reverb(x) at slab sample s.
template <int SPW>
__global__ __launch_bounds__(64) void reverb_kernel(ReverbState r, const double *__restrict__ x,
                                                    int ns, const double *__restrict__ gain,
                                                    float *__restrict__ partial,
                                                    size_t partial_stride)
{
    const int v = blockIdx.x * 64 + threadIdx.x;
    const bool live = v < r.n;
    const int s_begin = blockIdx.y * SPW;
    const size_t n = (size_t)r.n;
    const double g = live ? gain[v] : 0.;
    const double echo_count = (double)r.n_echos;

    double xin[SPW];
    double old[SPW][kMaxEchos];

    // Row of comb k at slab sample s (wave-uniform: scalar registers).  s < ns <=
    // min_length <= length and position < length, so one conditional subtract wraps.
    auto ring_row = [&](int k, int s) -> size_t {
        int slot = r.position[k] + s;
        slot = slot >= r.length[k] ? slot - r.length[k] : slot;
        return ((size_t)r.offset[k] + (size_t)slot) * n;
    };

    // issue every load of this wave first (they are independent of its stores)
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int s = s_begin + j;
        const bool on = live && s < ns;
        xin[j] = on ? x[(size_t)s * n + v] : 0.;
#pragma unroll
        for (int k = 0; k < kMaxEchos; ++k) {
            if (k < r.n_echos)
                old[j][k] = on ? r.rings[ring_row(k, s) + v] : 0.;
        }
    }

#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        const int s = s_begin + j;
        if (s >= ns)
            break;
        double acc = 0.;
#pragma unroll
        for (int k = 0; k < kMaxEchos; ++k) {
            if (k < r.n_echos) {
                const double y = xin[j] + r.alpha[k] * old[j][k];  // _filters.c:275
                if (live)
                    r.rings[ring_row(k, s) + v] = y;               // _filters.c:276
                acc = (k == 0) ? y : acc + y;                      // effects.py:610-612
            }
        }
        const double wet = acc / echo_count;                         // effects.py:614
        const double out = (1. - r.dry_wet) * xin[j] + r.dry_wet * wet;  // math.py:75
        const double mixed = wave_sum_f64(live ? g * out : 0.);
        if (threadIdx.x == 0)
            partial[(size_t)blockIdx.x * partial_stride + s] = (float)mixed;
    }
}

// bus[s] = (sum_g partial[g * stride + s]) / divisor
__global__ void bus_reduce_strided_kernel(const float *__restrict__ partial, int n_groups,
                                          size_t stride, size_t bus_len, double divisor,
                                          float *__restrict__ bus)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bus_len)
        return;
    double acc = 0.;
    for (int g = 0; g < n_groups; ++g)
        acc += (double)partial[(size_t)g * stride + s];
    bus[s] = (float)(acc / divisor);
}

}  // namespace klmi
