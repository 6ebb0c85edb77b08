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
// This kernel is HBM-bound: per voice-sample it moves n_echos * 2 * sizeof(RingT)
// bytes of ring state plus 8 bytes of dry input.  RingT is the STORAGE type of the
// delay lines: double as in _filters.c:33, or float (the arithmetic stays float64; one
// rounding per ring write, see DESIGN.md for the measured effect).
// STEREO: the voice's output goes to two buses with gain*pan_l / gain*pan_r
// (mixer.py:95-104), two partial planes per wave.
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
    void *rings;                  // double or float, see the kernel's RingT
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m, 64);
    return v;
}

// grid = (ceil(n/64), ceil(ns / SPW)), block = 64.
// x: [ns][n] dry input; partial[(group * planes + plane) * partial_stride + s] = sum over
// the wave's voices of gain_plane * Reverb(x) at slab sample s (planes = STEREO ? 2 : 1).
template <int SPW, typename RingT, bool STEREO>
__global__ __launch_bounds__(64) void reverb_kernel(ReverbState r, const double *__restrict__ x,
                                                    int ns, const double *__restrict__ gain_l,
                                                    const double *__restrict__ gain_r,
                                                    float *__restrict__ partial,
                                                    size_t partial_stride)
{
    const int v = blockIdx.x * 64 + threadIdx.x;
    const bool live = v < r.n;
    const int s_begin = blockIdx.y * SPW;
    const size_t n = (size_t)r.n;
    const double g = live ? gain_l[v] : 0.;
    const double g2 = (STEREO && live) ? gain_r[v] : 0.;
    const double echo_count = (double)r.n_echos;
    RingT *__restrict__ rings = static_cast<RingT *>(r.rings);

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
                old[j][k] = on ? (double)rings[ring_row(k, s) + v] : 0.;
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
                    rings[ring_row(k, s) + v] = (RingT)y;          // _filters.c:276
                acc = (k == 0) ? y : acc + y;                      // effects.py:610-612
            }
        }
        const double wet = acc / echo_count;                         // effects.py:614
        const double out = (1. - r.dry_wet) * xin[j] + r.dry_wet * wet;  // math.py:75
        constexpr int PLANES = STEREO ? 2 : 1;
        const double mixed = wave_sum_f64(live ? g * out : 0.);
        if (threadIdx.x == 0)
            partial[(size_t)blockIdx.x * PLANES * partial_stride + s] = (float)mixed;
        if constexpr (STEREO) {
            const double mixed_r = wave_sum_f64(live ? g2 * out : 0.);
            if (threadIdx.x == 0)
                partial[((size_t)blockIdx.x * PLANES + 1) * partial_stride + s] = (float)mixed_r;
        }
    }
}

// bus[block][plane][i] = (sum_g partial[(g * planes + plane) * stride + block*256 + i]) / divisor
// (bus_len = samples of the span, a multiple of 256)
__global__ void bus_reduce_strided_kernel(const float *__restrict__ partial, int n_groups,
                                          int planes, size_t stride, size_t bus_len,
                                          double divisor, float *__restrict__ bus)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bus_len)
        return;
    const size_t block = s / kBlock, i = s % kBlock;
    for (int pl = 0; pl < planes; ++pl) {
        double acc = 0.;
        for (int g = 0; g < n_groups; ++g)
            acc += (double)partial[((size_t)g * planes + pl) * stride + s];
        bus[(block * planes + pl) * kBlock + i] = (float)(acc / divisor);
    }
}

}  // namespace klmi
