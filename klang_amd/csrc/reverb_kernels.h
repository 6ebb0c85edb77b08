// reverb_kernels.h -- klang's Reverb (effects.py:564-615): n prime-length
// feedback comb filters in parallel (BackwardCombFilter, _filters.c:257-280),
// mean, dry/wet blend -- for a bank of independent channels on gfx950.
//
// This kernel is HBM-bound: per channel-sample it moves n_echos * 2 * sizeof(RingT)
// bytes of delay-line state plus 8 bytes of dry input.  RingT is the STORAGE type
// of the delay lines: float (the arithmetic stays float64, every ring write is
// rounded once -- measured effect in DESIGN.md) or double as in _filters.c:33.
//
// Data layout: voice-major.  rings[v * stride + offset_k + slot]: a voice's combs are
// one contiguous run, so 4 consecutive samples of one comb are 4 consecutive slots (16
// bytes of float32) and a wavefront moves 1 KiB per memory instruction.  Every ring is
// followed by 3 MIRROR slots that repeat its first 3, so that the 4 slots starting
// anywhere in the ring can be read with one access, and preceded by 3 spare slots.  A
// lane whose 4 slots run over the ring's end, or touch its first 3, stores its 4 values
// a second time, one ring length lower or higher (about two lanes per 256 samples): the
// elements that have a second place land on it, the others in spare slots.  The dry
// input is x[v][sample] (float64), as the voice kernel's per-voice output mode writes it.
//
// Parallelism: every comb delay is >= min_length samples, so up to min_length
// consecutive samples (a "pass", 256 here) depend only on ring contents written before
// the pass.  A WAVEFRONT owns a few voices for the whole launch: lane l handles samples
// 4l .. 4l+3 of the pass for each of its voices in turn, then moves on to the next pass.
// What a pass reads was written BY THE SAME WAVE in an earlier pass -- at least five passes
// earlier with the default 30 ms pre-delay, in the immediately preceding pass (by another lane)
// when the shortest comb is only 257..511 samples long.  Vector memory operations of one
// wavefront reach the cache in program order, which is all the ordering needed
// (tests/test_gpu_reverb_scale.py pins the 6 ms regime bit for bit against the oracle); voices
// never interact, so there is no barrier and no launch boundary along time: one launch covers
// any number of buffers, and the memory latency of one wave is hidden by the other waves of
// the compute unit.
// Reverbs whose shortest comb is below 256 samples (preDelay < 6 ms) or that have another
// number of combs than the default 10 take reverb_short_kernel: one sample per lane,
// passes of at most 64 samples.
#pragma once
#include "dsp_device.h"

namespace klmi {

constexpr int kMaxEchos = 16;
constexpr int kReverbThreads = 256;                 // 4 independent waves per workgroup
constexpr int kReverbWaves = kReverbThreads / 64;
constexpr int kReverbPass = 256;                    // samples of one pass: 64 lanes x 4
constexpr int kReverbMirror = 3;                    // spare slots before, mirror slots behind every ring
constexpr int kReverbVoicesPerWaveMax = 16;

struct ReverbState {
    int n;         // voices
    int n_echos;
    int min_length;
    double dry_wet;
    int length[kMaxEchos];
    int position[kMaxEchos];      // ring write/read position at sample 0 of the launch
    int offset[kMaxEchos];        // first slot of comb k inside a voice's run
    long long stride;             // slots per voice (sum of the lengths)
    double alpha[kMaxEchos];
    void *rings;                  // float or double, see the kernel's RingT
};

// 4 consecutive elements at an address that is only element-aligned (a ring position is
// any slot): 16-byte accesses, which global memory takes at 4-byte alignment on gfx950.
template <typename T> struct Quad {
    T e[4];
};
template <typename T> __device__ __forceinline__ Quad<T> load_quad(const T *p)
{
    Quad<T> q;
    __builtin_memcpy(&q, p, sizeof q);
    return q;
}
template <typename T> __device__ __forceinline__ void store_quad(T *p, const Quad<T> &q)
{
    __builtin_memcpy(p, &q, sizeof q);
}

// grid = ceil(n / (4 * voices_per_wave)), block = 256 (4 independent waves).
// x: [n][x_stride] dry input (sample 0 of the launch at column 0); `total` samples, a
// multiple of 256, in passes of 256 (needs min_length >= 256).
// partial[row * row_stride + (block * planes + plane) * 256 + i], row = global wave index:
// sum over the wave's voices of gain_plane * Reverb(x) at sample block*256 + i of the launch
// (planes = STEREO ? 2 : 1; the layout bus_slice_kernel folds).  Rows of waves without
// voices are written as zeros.
// NE: the comb count (compile time: the per-comb values live in registers).
template <typename RingT, bool STEREO, int NE>
__global__ __launch_bounds__(kReverbThreads, 3) void reverb_kernel(
    ReverbState r, const double *__restrict__ x, size_t x_stride, int total,
    int voices_per_wave, const double *__restrict__ gain_l, const double *__restrict__ gain_r,
    float *__restrict__ partial, size_t row_stride)
{
    constexpr int KMAX = NE;
    constexpr int PLANES = STEREO ? 2 : 1;
    constexpr int n_echos = NE;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kReverbWaves + (threadIdx.x >> 6));
    const int v0 = min(wave * voices_per_wave, r.n);
    const int v1 = min(v0 + voices_per_wave, r.n);
    const double echo_count = (double)n_echos;
    const double dry = 1. - r.dry_wet, wet_gain = r.dry_wet;
    RingT *__restrict__ rings = static_cast<RingT *>(r.rings);
    float *__restrict__ row = partial + (size_t)wave * row_stride;
    const int base = lane * 4;

    int pos[KMAX];   // wave-uniform: positions at the start of the current pass
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        pos[k] = k < n_echos ? r.position[k] : 0;

    for (int s0 = 0; s0 < total; s0 += kReverbPass) {
        // slot of this lane's first sample in every comb (the same for all voices);
        // 256 <= min_length <= length and pos < length: one conditional subtract wraps
        int slot[KMAX];     // offset of the lane's first slot inside the voice's run
        int again[KMAX];    // 0, or the distance of the second store (+- the ring length)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < n_echos) {
                int s = pos[k] + base;
                s = s >= r.length[k] ? s - r.length[k] : s;
                slot[k] = s + r.offset[k];
                again[k] = s + 3 >= r.length[k] ? -r.length[k] : s < kReverbMirror ? r.length[k] : 0;
            }
        }
        double acc_l[4] = {0., 0., 0., 0.}, acc_r[4] = {0., 0., 0., 0.};
#pragma unroll 1
        for (int v = v0; v < v1; ++v) {
            RingT *ring = rings + (size_t)v * (size_t)r.stride;
            const double gl = gain_l[v];
            const double gr = STEREO ? gain_r[v] : 0.;
            // all loads of the voice first, then compute and store
            const Quad<double> xin = load_quad(x + (size_t)v * x_stride + (size_t)(s0 + base));
            Quad<RingT> old[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < n_echos)
                    old[k] = load_quad(ring + slot[k]);
            double sum[4];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < n_echos) {
                    Quad<RingT> y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double yj = xin.e[j] + r.alpha[k] * (double)old[k].e[j];  // _filters.c:275
                        y.e[j] = (RingT)yj;                                              // _filters.c:276
                        sum[j] = (k == 0) ? yj : sum[j] + yj;                            // effects.py:610-612
                    }
                    store_quad(ring + slot[k], y);
                    if (again[k] != 0)
                        store_quad(ring + slot[k] + again[k], y);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double wet = sum[j] / echo_count;                     // effects.py:614
                const double out = dry * xin.e[j] + wet_gain * wet;         // math.py:75
                acc_l[j] += gl * out;                                       // mixer.py:43-44
                if (STEREO)
                    acc_r[j] += gr * out;                                   // mixer.py:99
            }
        }
        // a pass is one buffer: block s0 / 256, samples base .. base + 3
        float *o = row + (size_t)(s0 / kBlock) * (PLANES * kBlock) + base;
        Quad<float> ql, qr;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ql.e[j] = (float)acc_l[j], qr.e[j] = (float)acc_r[j];
        store_quad(o, ql);
        if (STEREO)
            store_quad(o + kBlock, qr);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < n_echos) {
                pos[k] += kReverbPass;
                pos[k] = pos[k] >= r.length[k] ? pos[k] - r.length[k] : pos[k];
            }
        }
    }
}

// The same for reverbs whose shortest comb is below 256 samples: one sample per lane,
// passes of `pass` <= min(min_length, 64) samples, nothing unrolled (not a throughput path;
// the mirror slots are not used and not maintained).
template <typename RingT, bool STEREO>
__global__ __launch_bounds__(kReverbThreads) void reverb_short_kernel(
    ReverbState r, const double *__restrict__ x, size_t x_stride, int total, int pass,
    int voices_per_wave, const double *__restrict__ gain_l, const double *__restrict__ gain_r,
    float *__restrict__ partial, size_t row_stride)
{
    constexpr int PLANES = STEREO ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * kReverbWaves + (threadIdx.x >> 6));
    const int v0 = min(wave * voices_per_wave, r.n);
    const int v1 = min(v0 + voices_per_wave, r.n);
    const double echo_count = (double)r.n_echos;
    RingT *__restrict__ rings = static_cast<RingT *>(r.rings);
    float *__restrict__ row = partial + (size_t)wave * row_stride;
    int pos[kMaxEchos];
    for (int k = 0; k < kMaxEchos; ++k)
        pos[k] = k < r.n_echos ? r.position[k] : 0;
    for (int s0 = 0; s0 < total; s0 += pass) {
        const int ns = min(pass, total - s0);
        if (lane < ns) {
            const int s = s0 + lane;
            double acc_l = 0., acc_r = 0.;
            for (int v = v0; v < v1; ++v) {
                RingT *ring = rings + (size_t)v * (size_t)r.stride;
                const double xin = x[(size_t)v * x_stride + (size_t)s];
                double sum = 0.;
                for (int k = 0; k < r.n_echos; ++k) {
                    int e = pos[k] + lane;
                    e = e >= r.length[k] ? e - r.length[k] : e;
                    RingT *slot = ring + r.offset[k] + e;
                    const double y = xin + r.alpha[k] * (double)*slot;   // _filters.c:275
                    *slot = (RingT)y;                                    // _filters.c:276
                    sum = (k == 0) ? y : sum + y;                        // effects.py:610-612
                }
                const double out = (1. - r.dry_wet) * xin + r.dry_wet * (sum / echo_count);
                acc_l += gain_l[v] * out;
                if (STEREO)
                    acc_r += gain_r[v] * out;
            }
            float *o = row + (size_t)(s / kBlock) * (PLANES * kBlock) + (s % kBlock);
            o[0] = (float)acc_l;
            if (STEREO)
                o[kBlock] = (float)acc_r;
        }
        for (int k = 0; k < kMaxEchos; ++k) {
            if (k < r.n_echos) {
                pos[k] += ns;
                pos[k] = pos[k] >= r.length[k] ? pos[k] - r.length[k] : pos[k];
            }
        }
    }
}

}  // namespace klmi
