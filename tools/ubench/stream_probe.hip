// stream_probe.hip -- what does a plain streaming kernel reach on this GPU?  Variants of
// copy / in-place update / fill over 1 GiB buffers: launch shape (one float4 per thread vs
// grid-stride), accesses in flight per lane, non-temporal hints.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_probe.hip -o tools/ubench/bin/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

template <int KIND, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void probe(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const size_t j = i + k * stride;
            if (KIND != 2 && j < n4) {
                const float4 *p = (KIND == 1 ? dst : src) + j;
                if (NT) {
                    v[k].x = __builtin_nontemporal_load(&p->x); v[k].y = __builtin_nontemporal_load(&p->y);
                    v[k].z = __builtin_nontemporal_load(&p->z); v[k].w = __builtin_nontemporal_load(&p->w);
                } else v[k] = *p;
            }
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const size_t j = i + k * stride;
            if (j < n4) {
                float4 o = KIND == 2 ? make_float4(1.f, 2.f, 3.f, 4.f) : v[k];
                if (KIND == 1) o.x *= 1.0001f, o.y *= 1.0001f, o.z *= 1.0001f, o.w *= 1.0001f;
                if (NT) {
                    __builtin_nontemporal_store(o.x, &dst[j].x); __builtin_nontemporal_store(o.y, &dst[j].y);
                    __builtin_nontemporal_store(o.z, &dst[j].z); __builtin_nontemporal_store(o.w, &dst[j].w);
                } else dst[j] = o;
            }
        }
    }
}

// UNROLL accesses per thread inside ONE compact window: a workgroup covers 256*UNROLL
// consecutive float4
template <int KIND, int UNROLL>
__global__ __launch_bounds__(256) void probe_block(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    float4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
        if (KIND != 2 && base + k * 256 < n4)
            v[k] = (KIND == 1 ? dst : src)[base + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
        if (base + k * 256 < n4) {
            float4 o = KIND == 2 ? make_float4(1.f, 2.f, 3.f, 4.f) : v[k];
            if (KIND == 1) o.x *= 1.0001f, o.y *= 1.0001f, o.z *= 1.0001f, o.w *= 1.0001f;
            dst[base + k * 256] = o;
        }
}

// K read windows at distant offsets feeding one write window (the shape of a comb bank whose
// delay lines are interleaved in time order): dst[i] = sum_k src[(i + k * n4 / K) % n4]
template <int K>
__global__ __launch_bounds__(256) void probe_windows(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = make_float4(0, 0, 0, 0);
    // each window reads only 1/K of the buffer: chunks of 256 float4 (4 KiB), every K-th one
    const size_t chunk = i / 256, lane = i % 256;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t c = (chunk / K) * K + k;           // window k: chunk c of this group of K
        size_t j = (c * 256 + lane + (size_t)k * 1234567 * 256) % n4;
        const float4 v = src[j];
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    dst[i] = acc;
}

// 16-byte accesses at an address that is only 4-byte aligned (ring slots): update in place
struct Q4 { float e[4]; };
__global__ __launch_bounds__(256) void probe_misaligned(float *buf, size_t n4, int shift)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= n4) return;
    float *p = buf + i * 4 + shift;
    Q4 q;
    __builtin_memcpy(&q, p, 16);
    q.e[0] *= 1.0001f; q.e[1] *= 1.0001f; q.e[2] *= 1.0001f; q.e[3] *= 1.0001f;
    __builtin_memcpy(p, &q, 16);
}

// pieces of 4 KiB (1,024 floats) scattered with a 55,584-byte pitch, 10 pieces per "voice"
// at odd offsets: the shape of voice-major comb lines; each workgroup updates one piece
__global__ __launch_bounds__(256) void probe_pieces(float *buf, size_t n_floats, int shift)
{
    const size_t piece = blockIdx.x;
    const size_t voice = piece / 10, k = piece % 10;
    const size_t base = (voice * 13896 + k * 1389 + (voice * 37 + k * 101) % 300) % (n_floats - 2048);
    float *p = buf + base + threadIdx.x * 4 + shift;
    Q4 q;
    __builtin_memcpy(&q, p, 16);
    q.e[0] *= 1.0001f; q.e[1] *= 1.0001f; q.e[2] *= 1.0001f; q.e[3] *= 1.0001f;
    __builtin_memcpy(p, &q, 16);
}

static void run_misc(float4 *a, size_t n4)
{
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int shift = 0; shift < 4; ++shift) {
        const unsigned grid = (unsigned)(n4 / 256);
        hipLaunchKernelGGL(probe_misaligned, dim3(grid), dim3(256), 0, 0, (float *)a, n4, shift);
        CK(hipEventRecord(t0, 0));
        for (int r = 0; r < 10; ++r)
            hipLaunchKernelGGL(probe_misaligned, dim3(grid), dim3(256), 0, 0, (float *)a, n4, shift);
        CK(hipEventRecord(t1, 0));
        CK(hipEventSynchronize(t1));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("update 1 quad/thread, address + %d floats        %7.0f GB/s\n", shift, (double)n4 * 32 * 10 / (ms * 1e-3) / 1e9);
    }
    for (int shift = 0; shift < 2; ++shift) {
        const unsigned grid = 4096 * 10 * 4;   // pieces
        hipLaunchKernelGGL(probe_pieces, dim3(grid), dim3(256), 0, 0, (float *)a, n4 * 4, shift);
        CK(hipEventRecord(t0, 0));
        for (int r = 0; r < 10; ++r)
            hipLaunchKernelGGL(probe_pieces, dim3(grid), dim3(256), 0, 0, (float *)a, n4 * 4, shift);
        CK(hipEventRecord(t1, 0));
        CK(hipEventSynchronize(t1));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("update scattered 4 KiB pieces, + %d floats        %7.0f GB/s\n", shift, (double)grid * 8192 * 10 / (ms * 1e-3) / 1e9);
    }
}

template <int KIND, int UNROLL>
static void run_block(const char *name, float4 *a, float4 *b, size_t n4)
{
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const unsigned grid = (unsigned)((n4 + 256 * UNROLL - 1) / (256 * UNROLL));
    hipLaunchKernelGGL((probe_block<KIND, UNROLL>), dim3(grid), dim3(256), 0, 0, b, a, n4);
    CK(hipEventRecord(t0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((probe_block<KIND, UNROLL>), dim3(grid), dim3(256), 0, 0, b, a, n4);
    CK(hipEventRecord(t1, 0));
    CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    const double bytes = (double)n4 * 16 * (KIND <= 1 ? 2 : 1) * reps;
    printf("%-44s grid %8u  %7.0f GB/s\n", name, grid, bytes / (ms * 1e-3) / 1e9);
}

template <int KIND, int UNROLL, bool NT>
static void run(const char *name, float4 *a, float4 *b, size_t n4, unsigned grid)
{
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    if (grid == 0) grid = (unsigned)((n4 + 256 * UNROLL - 1) / (256 * UNROLL));
    hipLaunchKernelGGL((probe<KIND, UNROLL, NT>), dim3(grid), dim3(256), 0, 0, b, a, n4);
    CK(hipEventRecord(t0, 0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((probe<KIND, UNROLL, NT>), dim3(grid), dim3(256), 0, 0, b, a, n4);
    CK(hipEventRecord(t1, 0));
    CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    const double bytes = (double)n4 * 16 * (KIND <= 1 ? 2 : 1) * reps;
    printf("%-44s grid %8u  %7.0f GB/s\n", name, grid, bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t n4 = ((size_t)1 << 30) / 16;
    float4 *a, *b;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
    int cus = 256;
    run<0, 1, false>("copy   1 float4/thread", a, b, n4, 0);
    run<0, 4, false>("copy   4/thread, full grid", a, b, n4, 0);
    run<0, 4, false>("copy   grid-stride x4, 32 WG/CU", a, b, n4, cus * 32);
    run<0, 4, false>("copy   grid-stride x4, 8 WG/CU", a, b, n4, cus * 8);
    run<0, 8, false>("copy   grid-stride x8, 8 WG/CU", a, b, n4, cus * 8);
    run<0, 4, true>("copy   nt 4/thread, full grid", a, b, n4, 0);
    run<0, 8, true>("copy   nt grid-stride x8, 8 WG/CU", a, b, n4, cus * 8);
    run<1, 1, false>("update 1 float4/thread", a, b, n4, 0);
    run<1, 4, false>("update 4/thread, full grid", a, b, n4, 0);
    run<1, 8, false>("update grid-stride x8, 8 WG/CU", a, b, n4, cus * 8);
    run<1, 4, true>("update nt 4/thread, full grid", a, b, n4, 0);
    run<2, 1, false>("fill   1 float4/thread", a, b, n4, 0);
    run<2, 4, false>("fill   4/thread, full grid", a, b, n4, 0);
    run<2, 4, true>("fill   nt 4/thread, full grid", a, b, n4, 0);
    run<2, 8, false>("fill   grid-stride x8, 8 WG/CU", a, b, n4, cus * 8);
    run_misc(a, n4);
    run_block<0, 2>("copy   2/thread, one window", a, b, n4);
    run_block<0, 4>("copy   4/thread, one window", a, b, n4);
    run_block<0, 8>("copy   8/thread, one window", a, b, n4);
    run_block<1, 4>("update 4/thread, one window", a, b, n4);
    run_block<1, 8>("update 8/thread, one window", a, b, n4);
    run_block<2, 4>("fill   4/thread, one window", a, b, n4);
    return 0;
}
