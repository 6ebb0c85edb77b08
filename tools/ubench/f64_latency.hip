// Micro-benchmark: float64 VALU dependent-issue latency on gfx950: C independent
// chains of v_fma_f64 per lane, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off f64_latency.hip -o bin/f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 8192;

template <int C>
__global__ void kern(double *out, double a, double b)
{
    double x[C];
#pragma unroll
    for (int i = 0; i < C; ++i)
        x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < C; ++i)
                x[i] = __builtin_fma(x[i], b, a);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < C; ++i)
        s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int C>
void run(double *d, int waves_per_simd)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<C><<<256 * waves_per_simd, 256>>>(d, 1.0000001, 0.9999999);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    kern<C><<<256 * waves_per_simd, 256>>>(d, 1.0000001, 0.9999999);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)ITERS * 8 * C * waves_per_simd;
    printf("chains %2d waves/SIMD %d: %.3f ms  %.2f cycles(2.4GHz) per fma per SIMD\n", C,
           waves_per_simd, ms, (ms - 0.006) * 1e-3 * 2.4e9 / instr);
}

int main()
{
    double *d;
    (void)hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    for (int w : {1, 2}) {
        run<1>(d, w);
        run<2>(d, w);
        run<3>(d, w);
        run<4>(d, w);
        run<8>(d, w);
    }
    return 0;
}
