// bus_kernels.h -- deterministic fold of the per-wave partial buses (gfx950).
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

namespace klmi {

// Deterministic bus reduction in two steps (no float atomics anywhere):
//   stage[slice][s] = sum of the partial rows of `slice` (ascending, float64)
//   bus[s]          = (sum of the slices, ascending, float64) / divisor -> float32
// The slice count depends on the row count only, so the association is fixed for
// a given bank size.  One thread owns 4 consecutive samples (16-byte loads).
constexpr int kBusSlices = 32;

__host__ __device__ inline int bus_slices(int n_rows)
{
    return n_rows < kBusSlices ? n_rows : kBusSlices;
}

// grid = (ceil(bus_len / 1024), slices), block = 256; bus_len is a multiple of 256
__global__ __launch_bounds__(256) void bus_slice_kernel(const float *__restrict__ partial,
                                                        int n_rows, size_t bus_len,
                                                        double *__restrict__ stage)
{
    const size_t s = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (s >= bus_len)
        return;
    const int slices = gridDim.y;
    const int r0 = (int)((long long)n_rows * blockIdx.y / slices);
    const int r1 = (int)((long long)n_rows * (blockIdx.y + 1) / slices);
    double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
        typedef float Float4 __attribute__((ext_vector_type(4)));
        const Float4 p = __builtin_nontemporal_load(reinterpret_cast<const Float4 *>(partial + (size_t)r * bus_len + s));
        a0 += (double)p.x, a1 += (double)p.y, a2 += (double)p.z, a3 += (double)p.w;
    }
    double *o = stage + (size_t)blockIdx.y * bus_len + s;
    o[0] = a0, o[1] = a1, o[2] = a2, o[3] = a3;
}

__global__ void bus_finish_kernel(const double *__restrict__ stage, int slices, size_t bus_len,
                                  double divisor, float *__restrict__ bus)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bus_len)
        return;
    double acc = 0.;
    for (int k = 0; k < slices; ++k)
        acc += stage[(size_t)k * bus_len + s];
    bus[s] = (float)(acc / divisor);
}

// A mix bus [blocks][channels][256] float32 as the external input [span][n] float64 of
// another bank (klmi_bank_set_input_from_bank): voice v listens to channel v % channels.
__global__ void bus_to_input_kernel(const float *__restrict__ mix, int channels, int n,
                                    size_t span, double *__restrict__ ext)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= span)
        return;
    const size_t block = s / 256, i = s % 256;
    for (int v = 0; v < n; ++v)
        ext[s * n + v] = (double)mix[(block * channels + (size_t)(v % channels)) * 256 + i];
}

// Streaming probes for klmi_ctx_stream_rate: what this GPU's memory system gives a plain
// hand-written kernel, as the yardstick for the HBM-bound kernels.  One 16-byte access per
// lane, one workgroup per 4 KiB, workgroups dispatched in address order: the shape that
// measured fastest on MI355X (tools/ubench/stream_probe.hip: grid-stride loops and several
// accesses per lane in distant windows run 15-30 % slower).
//   kind 0 copy   dst[i] = src[i]            (read + write, two buffers)
//   kind 1 update buf[i] = buf[i] * a        (read-modify-write in place: the reverb's mix)
//   kind 2 fill   dst[i] = a                 (write only)
//   kind 3 sum    keeps src[i] live          (read only)
template <int KIND>
__global__ __launch_bounds__(256) void stream_probe_kernel(const float4 *__restrict__ src,
                                                           float4 *__restrict__ dst, size_t n4,
                                                           float a)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4)
        return;
    if constexpr (KIND == 0) {
        dst[i] = src[i];
    } else if constexpr (KIND == 1) {
        const float4 v = dst[i];
        dst[i] = make_float4(v.x * a, v.y * a, v.z * a, v.w * a);
    } else if constexpr (KIND == 2) {
        dst[i] = make_float4(a, a, a, a);
    } else {
        const float4 v = src[i];
        if (v.x + v.y + v.z + v.w == 12345.678f)   // keeps the load alive, never true on zeros
            dst[i] = v;
    }
}

}  // namespace klmi
