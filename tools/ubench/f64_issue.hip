// Micro-benchmark: float64 VALU issue rate on gfx950, by instruction and by
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off f64_issue.hip -o f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;  // independent chains per lane

template <int OP>
__global__ void kern(double *out, double a, double b)
{
    double x[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i)
        x[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) x[i] = x[i] * b;                       // v_mul_f64
            if (OP == 1) x[i] = x[i] + b;                       // v_add_f64
            if (OP == 2) x[i] = __builtin_fma(x[i], b, a);      // v_fma_f64
            if (OP == 3) x[i] = floor(x[i]) + 0.;               // v_floor_f64 (+ nothing)
            if (OP == 4) x[i] = (double)(float)x[i];            // cvt f64->f32->f64
            if (OP == 5) x[i] = x[i] < b ? a : x[i];            // v_cmp + 2 cndmask
            if (OP == 6) x[i] = fmin(x[i], b);                  // v_min_f64
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i)
        s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int waves_per_simd, double *d)
{
    // 256 CUs x 4 SIMDs; blocks of 256 threads = 4 waves = one per SIMD
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<OP><<<blocks, 256>>>(d, 1.0000001, 0.9999999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<OP><<<blocks, 256>>>(d, 1.0000001, 0.9999999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)ITERS * UNROLL;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    printf("%-12s waves/SIMD %d: %.3f ms, %.2f cycles(2.4GHz)/wave-instr/SIMD\n", name,
           waves_per_simd, ms, cyc);
}

int main()
{
    double *d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
    for (int w : {1, 2, 4}) {
        run<0>("mul_f64", w, d);
        run<1>("add_f64", w, d);
        run<2>("fma_f64", w, d);
        run<3>("floor+add", w, d);
        run<4>("cvt x2", w, d);
        run<5>("cmp+2cnd", w, d);
        run<6>("min_f64", w, d);
    }
    return 0;
}
