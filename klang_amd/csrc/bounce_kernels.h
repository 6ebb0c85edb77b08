// bounce_kernels.h -- the output side of an offline bounce on gfx950:
// Dac.collect_samples / pack_signals (klang/audio/klanggeber.py:81-97, 164-176), the
// fade-out (klanggeber.py:205-209, 243-249) and write_wave's peak normalisation to int16
// (klang/audio/wavfile.py:54-63, klang/math.py:14-21).  HBM-bound streaming kernels:
// every access is a coalesced 4- to 16-byte-per-lane row; the peak is a maximum, so its
// reduction order does not matter and no atomics are needed.
#pragma once
#include "bank_kernels.h"

namespace klmi {

// One thread = one frame (time point) of the piece.  bus: [n_blocks][CH][256] float32,
// frames: [n_blocks*256][CH] float32, play_time: [n_blocks] float64 (the host's running sum
// of INTERVAL, klanggeber.py:248), group_peak: [gridDim.x] max |sample| of this group.
// blockDim.x == 256 == one buffer, so blockIdx.x is the buffer.
template <int CH>
__global__ __launch_bounds__(kBlock) void bounce_pack_kernel(const float *__restrict__ bus,
                                                             const double *__restrict__ play_time,
                                                             double fade_start, double duration,
                                                             float *__restrict__ frames,
                                                             float *__restrict__ group_peak)
{
    __shared__ float wave_peak[kBlock / 64];
    const int b = blockIdx.x, i = threadIdx.x;
    const double t0 = play_time[b];
    const bool fading = t0 >= fade_start;        // klanggeber.py:243, per buffer
    double env = 1.;
    if (fading) {                                // np.interp(t + T, [start, duration], [1, 0])
        const double slope = (0. - 1.) / (duration - fade_start);
        const double x = t0 + c_time[i];
        if (x > duration || x == duration)
            env = 0.;
        else if (x < fade_start)
            env = 1.;
        else
            env = slope * (x - fade_start) + 1.;
    }
    float v[CH];
    float peak = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float s = bus[((size_t)b * CH + c) * kBlock + i];
        if (fading)
            s = (float)((double)s * env);        // float32 array *= float64 array
        v[c] = s;
        peak = fmaxf(peak, fabsf(s));
    }
    const size_t f = (size_t)b * kBlock + i;
    if constexpr (CH == 2)
        reinterpret_cast<float2 *>(frames)[f] = make_float2(v[0], v[1]);
    else
        frames[f] = v[0];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        peak = fmaxf(peak, __shfl_xor(peak, off));
    if ((i & 63) == 0)
        wave_peak[i >> 6] = peak;
    __syncthreads();
    if (i == 0)
        group_peak[b] = fmaxf(fmaxf(wave_peak[0], wave_peak[1]), fmaxf(wave_peak[2], wave_peak[3]));
}

// peak[0] = max of group_peak[0..n): one workgroup
__global__ __launch_bounds__(256) void bounce_peak_kernel(const float *__restrict__ group_peak, int n,
                                                          float *__restrict__ peak)
{
    __shared__ float part[4];
    float m = 0.f;
    for (int k = threadIdx.x; k < n; k += 256)
        m = fmaxf(m, group_peak[k]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        peak[0] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// wavfile.py:59: (audio / peak * 32767).astype(int16) in float32; 4 samples per thread
// (16-byte load, 8-byte store).  total is a multiple of 4 (256 frames per buffer).
__global__ __launch_bounds__(256) void bounce_pcm_kernel(const float *__restrict__ frames, size_t total4,
                                                         const float *__restrict__ peak,
                                                         int16_t *__restrict__ pcm)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total4)
        return;
    const float p = peak[0];
    const float4 x = reinterpret_cast<const float4 *>(frames)[k];
    short4 o;
    o.x = (short)(int)(x.x / p * 32767.f);
    o.y = (short)(int)(x.y / p * 32767.f);
    o.z = (short)(int)(x.z / p * 32767.f);
    o.w = (short)(int)(x.w / p * 32767.f);
    reinterpret_cast<short4 *>(pcm)[k] = o;
}

}  // namespace klmi
